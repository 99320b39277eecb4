"""world_size-2 gloo tests (CPU) of the N>1 logic: view sharding, flat gradient all-reduce, image all-gather."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gaussctrl_amd.dist import allgather_view_images, allreduce_gradients, shard_views
        n = 7
        mine = shard_views(n, world, rank)
        # gradients: each rank holds grad = rank+1 on differently shaped tensors
        ps = [torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(5, 15, 3)), torch.nn.Parameter(torch.zeros(5, 1))]
        for p in ps:
            p.grad = torch.full_like(p, float(rank + 1))
        ps.append(torch.nn.Parameter(torch.zeros(2)))           # no grad: skipped
        allreduce_gradients(ps, world)
        ok_grad = all(torch.allclose(p.grad, torch.full_like(p, sum(range(1, world + 1)) / world)) for p in ps[:3])
        local = {v: torch.full((4, 6, 3), float(v)) for v in mine}
        allv = allgather_view_images(local, n, world, rank, (4, 6, 3), "cpu")
        ok_img = sorted(allv) == list(range(n)) and all(float(allv[v].mean()) == float(v) for v in range(n))
        # reference K / V^T bank: owner rank 1 -> everyone, one flat message per DDIM step, K keeps the Q|K row stride
        from gaussctrl_amd.dist import broadcast_ref_bank
        from gaussctrl_amd.sd.unet import RefBank
        bank = RefBank()
        g = torch.Generator().manual_seed(5)
        want = {}
        for st in range(3):
            for li, (L_, C_) in enumerate([(16, 8), (4, 24)]):
                qk = torch.randn(8, L_, 2 * C_, generator=g).to(torch.bfloat16)
                vt = torch.randn(8, C_, L_, generator=g).to(torch.bfloat16)
                want[(st, ("unet", f"layer{li}"))] = (qk[..., C_:].clone(), vt.clone())
                if rank == 1:
                    bank.store[(st, ("unet", f"layer{li}"))] = (qk[..., C_:], vt)
        if rank == 1:
            bank.mode = "use"
        broadcast_ref_bank(bank, 1, world, rank, "cpu")
        ok_bank = bank.mode == "use" and set(bank.store) == set(want)
        for key, (k, vt) in bank.store.items():
            ok_bank = ok_bank and torch.equal(k, want[key][0]) and torch.equal(vt, want[key][1]) and k.stride(1) == 2 * k.shape[2]
        # pipelined variant: the owner advances the reference trajectory one DDIM step at a time and posts each step's K / V^T as an
        # async broadcast; a stand-in pipe records deterministic K / V^T per step (the real one needs the GPU)
        from gaussctrl_amd.dist import broadcast_ref_bank_pipelined

        class FakePipe:
            def begin_ref_bank(self, *a):
                b = RefBank(); b.mode = "record"
                return {"bank": b, "i": 0}

            def advance_ref_bank(self, tr, n):
                st = tr["i"]
                for li, (L_, C_) in enumerate([(16, 8), (4, 24)]):
                    qk, vt = want2[(st, ("unet", f"layer{li}"))]
                    tr["bank"].store[(st, ("unet", f"layer{li}"))] = (qk[..., C_:], vt)
                tr["i"] += 1
                if tr["i"] == 3:
                    tr["bank"].mode = "use"
                    return tr["bank"]
                return None
        g2 = torch.Generator().manual_seed(9)
        want2 = {}
        for st in range(3):
            for li, (L_, C_) in enumerate([(16, 8), (4, 24)]):
                want2[(st, ("unet", f"layer{li}"))] = (torch.randn(8, L_, 2 * C_, generator=g2).to(torch.bfloat16),
                                                       torch.randn(8, C_, L_, generator=g2).to(torch.bfloat16))
        bank2 = broadcast_ref_bank_pipelined(FakePipe(), None, None, None, None, 0, world, rank, "cpu", 3)
        ok_pipe = bank2.mode == "use" and set(bank2.store) == set(want2)
        for key, (k, vt) in bank2.store.items():
            C_ = vt.shape[1]
            ok_pipe = ok_pipe and torch.equal(k, want2[key][0][..., C_:]) and torch.equal(vt, want2[key][1]) and k.stride(1) == 2 * C_
        # the stream form bench.py / a scene-streaming caller uses: steps posted in uneven groups between other collectives (a flat
        # gradient all-reduce on the default group), rotating owner (rank 1), layout passed on from the previous stream
        from gaussctrl_amd.dist import FlatGrads, RefBankStream
        first = RefBankStream(FakePipe(), 0, world, rank, "cpu", 3).begin(None, None, None, None)
        first.advance(None); first.finish()
        st = RefBankStream(FakePipe(), 1, world, rank, "cpu", 3, layers=first.layers).begin(None, None, None, None)
        fg = FlatGrads({"a": torch.zeros(5, 3), "b": torch.zeros(4, 15, 3), "c": torch.zeros(7)})
        assert fg.views["b"].data_ptr() == fg.flat[15:].data_ptr() and fg.flat.numel() == 15 + 180 + 7
        done = st.advance(1)
        fg.views["b"].fill_(float(rank + 1)); fg.views["c"].fill_(2.0)
        fg.reduce_async(world)
        assert not done
        st.drain(1)
        done = st.advance(2)
        fg.wait()
        ok_fg = bool((fg.views["b"] == 3.0).all()) and bool((fg.views["c"] == 4.0).all()) and bool((fg.views["a"] == 0).all())
        bank3 = st.finish()
        ok_stream = done and ok_fg and bank3.mode == "use" and set(bank3.store) == set(want2)
        for key, (k, vt) in bank3.store.items():
            C_ = vt.shape[1]
            ok_stream = ok_stream and torch.equal(k, want2[key][0][..., C_:]) and torch.equal(vt, want2[key][1])
        ret[rank] = (mine, ok_grad, ok_img and ok_bank and ok_pipe and ok_stream)
    finally:
        dist.destroy_process_group()


def test_view_sharding_partitions():
    from gaussctrl_amd.dist import shard_views
    for n in (1, 7, 40, 80):
        for w in (1, 2, 4, 8):
            parts = [shard_views(n, w, r) for r in range(w)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_balanced_sharding_accounts_for_the_reference_trajectory():
    from gaussctrl_amd.dist import shard_views_balanced, split_chunks
    for n, w in ((40, 8), (40, 4), (40, 2), (80, 8), (7, 4), (3, 8)):
        for owner in range(w):
            parts = [shard_views_balanced(n, w, r, owner) for r in range(w)]
            assert sorted(sum(parts, [])) == list(range(n))                       # a partition of the views
            others = [len(p) for r, p in enumerate(parts) if r != owner]
            assert max(others) - min(others) <= 1
            assert len(parts[owner]) <= min(others)                              # the owner edits fewer: it also runs the references
            if n >= 4 * w:
                assert abs((len(parts[owner]) + 4) - sum(others) / len(others)) <= 1.0
    assert [len(shard_views_balanced(40, 8, r, 3)) for r in range(8)] == [6, 6, 6, 2, 5, 5, 5, 5]
    assert shard_views_balanced(40, 8, 2, -1) == [2, 10, 18, 26, 34]              # no owner: plain v % N
    assert shard_views_balanced(40, 1, 0, 0) == list(range(40))
    assert [len(c) for c in split_chunks(list(range(7)), 4, 3)] == [2, 2, 2, 1]
    assert [len(c) for c in split_chunks(list(range(2)), 2, 3)] == [1, 1]
    assert split_chunks([], 2, 3) == [[], []]


def test_gloo_world2_collectives():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, 29000 + os.getpid() % 2000, ret), nprocs=world, join=True)
    assert ret[0][0] == [0, 2, 4, 6] and ret[1][0] == [1, 3, 5]
    assert all(ret[r][1] and ret[r][2] for r in range(world))


def _shard_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gaussctrl_amd.dist import RefShard
        sh = RefShard(world, rank)
        g = torch.Generator().manual_seed(9)
        L, C, Lp = 24, 16, 24
        qk = torch.randn(8, L, 2 * C, generator=g).to(torch.bfloat16)          # the full reference batch, sample s = half * 4 + frame
        vt = torch.randn(8, C, Lp, generator=g).to(torch.bfloat16)
        eps = torch.randn(8, 4, 4, 8, generator=g)
        mine = sh.samples
        ok = mine == [s for s in range(8) if s % world == rank] and sh.frames == sorted({s % 4 for s in mine})
        ok = ok and sh.half_base == 4 * (mine[0] // 4) and len(sh.halves) == (1 if world == 8 else 2)
        kr, vr = sh.gather_kv(qk[mine][..., C:], vt[mine])                      # K is a column slice of the Q | K buffer, as in the network
        ok = ok and torch.equal(kr, qk[..., C:]) and torch.equal(vr, vt) and kr.stride(1) == 2 * C and kr.shape == (8, L, C)
        if len(sh.halves) == 1:
            pair = sh.gather_eps_pairs(eps[mine])
            f = sh.frames
            ok = ok and torch.equal(pair, torch.cat([eps[f], eps[[4 + x for x in f]]]))
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_reference_trajectory_shards_allgather_layout(world):
    """dist.RefShard (north_star's per-layer all-gather of the reference K / V^T): sample -> rank assignment, the gathered bank in
    sample order with the Q | K row stride, and the CFG partner exchange when a rank holds a single half (world 8)."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_shard_worker, args=(world, 29650 + world + os.getpid() % 200, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world)), dict(ret)
