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


# ------------------------------------------------------------------------------------------------ train modes
class _FusedFake(torch.autograd.Function):
    """stand-in of the fused render + backward with the RenderAux.grad_into contract: loss_view = sum_k w_view * (p_k ** 2).sum(); when the
    model carries `grad_into` the backward WRITES the leaf gradients into those buffers and hands autograd None (as gsplat_ops._RenderView)."""

    @staticmethod
    def forward(ctx, w, into, aux, *ps):
        ctx.save_for_backward(*ps)
        ctx.w, ctx.into, ctx.aux = w, into, aux
        return sum((p ** 2).sum() for p in ps) * w

    @staticmethod
    def backward(ctx, g):
        grads = [2 * p * ctx.w * g for p in ctx.saved_tensors]
        ctx.aux.xys_grad = grads[0]
        if ctx.into is not None:
            for k, gk in zip(("means", "scales", "quats", "opacities", "features_dc", "features_rest"), grads):
                ctx.into[k].copy_(gk)
            return (None,) * (3 + len(grads))
        return (None, None, None) + tuple(grads)


def _train_mode_worker(rank, world, port, mode, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import random
        from gaussctrl_amd.gc_datamanager import _NextTrainMixin
        from gaussctrl_amd.gc_pipeline import GaussCtrlPipeline

        class Model(torch.nn.Module):
            def __init__(self):
                super().__init__()
                g = torch.Generator().manual_seed(0)                     # replicated parameters
                shapes = dict(means=(5, 3), scales=(5, 3), quats=(5, 4), opacities=(5, 1), features_dc=(5, 3), features_rest=(5, 15, 3))
                for k, s in shapes.items():
                    setattr(self, k, torch.nn.Parameter(torch.randn(*s, generator=g)))
                self.grad_into = None
                self._aux = None

            def forward(self, view):
                into = None if getattr(self, "ignore_grad_into", False) else self.grad_into      # (a crop box makes the real model ignore it)
                self._aux = type("Aux", (), {"xys_grad": None, "grad_into": into})()
                ps = [getattr(self, k) for k in GaussCtrlPipeline._GRAD_KEYS]
                self.backgrounds.append(getattr(self, "background_override", None))
                return {"loss": _FusedFake.apply(float(view + 1), into, self._aux, *ps)}

            num_points = property(lambda self: self.means.shape[0])
            device = property(lambda self: self.means.device)
            backgrounds = []

            def get_metrics_dict(self, out, batch): return {}
            def get_loss_dict(self, out, batch, metrics=None): return {"main_loss": out["loss"]}

        class DM(_NextTrainMixin):
            def __init__(self):
                self.train_data = list(range(6)); self._init_sampling(6); self.seen = []

            def next_train(self, step):
                i = self._pop_view(); self.seen.append(i)
                return i, {}

        random.seed(13789)                   # the SAME global seed on every rank, as GaussCtrlPipeline.__init__ / the datamanager leave it
        pipe = object.__new__(GaussCtrlPipeline)
        torch.nn.Module.__init__(pipe)
        pipe.world_size, pipe.local_rank, pipe._spread_calls = world, rank, 0
        mode, no_into = (mode[:-len("+cropbox")], True) if mode.endswith("+cropbox") else (mode, False)
        pipe.config = type("C", (), {"train_mode": mode})()
        pipe.datamanager, pipe._model = DM(), Model()
        pipe._model.ignore_grad_into = no_into
        opts = {"all": torch.optim.SGD(pipe._model.parameters(), lr=0.0)}
        out = []
        for step in range(4):
            loss, _, _ = pipe.train_iteration(opts, step)
            out.append((float(loss), pipe.datamanager.seen[-1], {k: getattr(pipe._model, k).grad.clone() for k in GaussCtrlPipeline._GRAD_KEYS},
                        pipe._model.means.grad.data_ptr() == pipe._fg.views["means"].data_ptr() if mode == "throughput" else None))
        ret[rank] = (out, {k: getattr(pipe._model, k).detach().clone() for k in GaussCtrlPipeline._GRAD_KEYS},
                     [None if b is None else b.clone() for b in pipe._model.backgrounds])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["parity", "throughput", "throughput+cropbox"])
def test_train_modes_world2(mode):
    """GaussCtrlPipeline.train_iteration at world_size 2 over gloo with a stand-in model that honours the grad_into contract; every rank's global
    `random` is seeded identically, as in the product:
    parity     -- both ranks train on the view rank 0 drew against the background rank 0 drew (control-path broadcast), gradients equal the
                  single-view gradient, no reduction;
    throughput -- the ranks take DISTINCT views of one per-epoch permutation (no collective), the loss carries 1 / N, the flat buffer is
                  all-reduced in place: every rank ends with the MEAN of the two views' gradients, and the optimizers read views of that one
                  buffer (no autograd .grad tensors, no gather copy);
    +cropbox   -- the model does not honour grad_into (a crop box during training): autograd's gradients are moved into the flat buffer first,
                  the reduction never sees stale buffer contents."""
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_train_mode_worker, args=(2, 29800 + os.getpid() % 150 + ["parity", "throughput", "throughput+cropbox"].index(mode), mode, ret), nprocs=2, join=True)
    (o0, p0, b0), (o1, p1, b1) = ret[0], ret[1]
    seen = []
    for (l0, v0, g0, f0), (l1, v1, g1, f1) in zip(o0, o1):
        for k in g0:
            assert torch.equal(g0[k], g1[k]), (mode, k)                  # replicas stay in lock step in both modes
        if mode == "parity":
            assert v0 == v1 and l0 == l1
            assert torch.allclose(g0["means"], 2 * p0["means"] * (v0 + 1))
        else:
            assert v0 != v1, "two ranks rendered the same view in one step: the all-reduce would average copies of one gradient"
            assert (f0 and f1) if mode == "throughput" else True
            assert torch.allclose(g0["means"], 2 * p0["means"] * ((v0 + 1) + (v1 + 1)) / 2)
            seen += [v0, v1]
    if mode == "parity":
        assert all(a is not None and torch.equal(a, b) for a, b in zip(b0, b1)), "parity replicas must render against the same background"
        assert len({tuple(a.tolist()) for a in b0}) == len(b0), "a fresh background per step"
    else:
        assert sorted(seen[:6]) == list(range(6)), "one epoch of the view schedule covers every view once across the ranks"


# ---------------------------------------------------------------------------------------------------------------- sharded optimizer (8e, collective 2)
_HYPER = {"means": (1.6e-2, 1e-15), "scales": (5e-3, 1e-15), "quats": (1e-3, 1e-15), "opacities": (5e-2, 1e-15), "features_dc": (2.5e-3, 1e-15),
          "features_rest": (1.25e-4, 1e-15)}
_SHAPES = dict(means=(7, 3), scales=(7, 3), quats=(7, 4), opacities=(7, 1), features_dc=(7, 3), features_rest=(7, 15, 3))      # 413 elements: not a multiple of 2 x 4


def _ref_adam(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step):
    """torch.optim.Adam (no weight decay / amsgrad) on one slice, in place -- the CPU stand-in of gc_adam_step for the gloo tests only."""
    exp_avg.mul_(beta1).add_(grad, alpha=1 - beta1)
    exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
    param.addcdiv_(exp_avg, (exp_avg_sq.sqrt() / (bc2 ** 0.5)).add_(eps), value=-lr / bc1)


def _init_params():
    g = torch.Generator().manual_seed(0)
    return {k: torch.nn.Parameter(torch.randn(*s, generator=g)) for k, s in _SHAPES.items()}


def _rank_grad(k, rank, step):
    g = torch.Generator().manual_seed(1000 * step + 10 * rank + sorted(_SHAPES).index(k))
    return torch.randn(*_SHAPES[k], generator=g)


def _sharded_adam_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gaussctrl_amd.dist import FlatGrads, FlatParams, ShardedAdam
        params = _init_params()
        fp = FlatParams(params, world)
        fg = FlatGrads(params, pad_to=fp.flat.numel())
        sa = ShardedAdam(fp, fg, world, rank, adam=_ref_adam)
        assert fp.matches(params) and all(p.data_ptr() == fp.flat.data_ptr() + 4 * fp.spans[k][0] for k, p in params.items())
        for step in range(3):
            for k in params:
                fg.views[k].copy_(_rank_grad(k, rank, step))
            sa.step(_HYPER)
        ret[rank] = ({k: p.detach().clone() for k, p in params.items()}, sa.exp_avg.numel(), fp.shard, fp.flat.numel())
    finally:
        dist.destroy_process_group()


def test_sharded_adam_world2_equals_replicated_adam():
    """dist.ShardedAdam (reduce-scatter of the flat gradient buffer -> Adam on each rank's slice of the flat parameter buffer -> all-gather) at
    world 2 over gloo: after 3 steps both ranks hold the parameters a single process gets from Adam on the SUMMED gradients with the per-group
    lr / eps, bit for bit; a rank keeps moments for its slice only; parameter tensors whose elements straddle the slice boundary are handled."""
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_sharded_adam_worker, args=(2, 29500 + os.getpid() % 150, ret), nprocs=2, join=True)
    (p0, nstate, shard, padded), (p1, _, _, _) = ret[0], ret[1]
    assert nstate == shard and padded == 2 * shard and shard % 4 == 0 and shard < sum(int(torch.tensor(s).prod()) for s in _SHAPES.values())
    ref = _init_params()
    st = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in ref.items()}
    with torch.no_grad():
        for step in range(3):
            for k, v in ref.items():
                g = _rank_grad(k, 0, step) + _rank_grad(k, 1, step)
                _ref_adam(v.data, g, st[k][0], st[k][1], _HYPER[k][0], 0.9, 0.999, _HYPER[k][1], step + 1)
    for k in ref:
        assert torch.equal(p0[k], p1[k]), k
        assert torch.equal(p0[k], ref[k].detach()), k


def _sharded_mode_worker(rank, world, port, cull, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import random
        import types
        import gaussctrl_amd.dist as gdist
        from gaussctrl_amd.gc_datamanager import _NextTrainMixin
        from gaussctrl_amd.gc_pipeline import GaussCtrlPipeline
        gdist._hip_adam = _ref_adam                     # (CPU test only: the product's slice update is the HIP kernel and raises on CPU tensors)

        class Model(torch.nn.Module):
            def __init__(self):
                super().__init__()
                for k, v in _init_params().items():
                    setattr(self, k, v)
                self.grad_into = None
                self._aux = None

            def forward(self, view):
                self._aux = type("Aux", (), {"xys_grad": None, "grad_into": self.grad_into})()
                ps = [getattr(self, k) for k in GaussCtrlPipeline._GRAD_KEYS]
                return {"loss": _FusedFake.apply(float(view + 1), self.grad_into, self._aux, *ps)}

            num_points = property(lambda self: self.means.shape[0])
            device = property(lambda self: self.means.device)

            def get_metrics_dict(self, out, batch): return {}
            def get_loss_dict(self, out, batch, metrics=None): return {"main_loss": out["loss"]}

        class DM(_NextTrainMixin):
            def __init__(self):
                self.train_data = list(range(6)); self._init_sampling(6); self.seen = []

            def next_train(self, step):
                i = self._pop_view(); self.seen.append(i)
                return i, {}

        random.seed(13789)                  # the same global seed on every rank, as in the product
        pipe = object.__new__(GaussCtrlPipeline)
        torch.nn.Module.__init__(pipe)
        pipe.world_size, pipe.local_rank, pipe._spread_calls = world, rank, 0
        pipe.config = type("C", (), {"train_mode": "sharded"})()
        pipe.datamanager, pipe._model = DM(), Model()
        stepped = []
        opts = {g: types.SimpleNamespace(param_groups=[{"lr": _HYPER[key][0], "eps": _HYPER[key][1]}], zero_grad=lambda set_to_none=True: None,
                                         step=lambda: (_ for _ in ()).throw(AssertionError("train_mode 'sharded' must not call the per-group optimizers")))
                for g, key in GaussCtrlPipeline._GROUP_OF.items()}
        opts["camera_opt"] = types.SimpleNamespace(param_groups=[{"lr": 1e-3, "eps": 1e-15}], zero_grad=lambda set_to_none=True: None,
                                                   step=lambda: stepped.append(1))      # a group outside the six leaf tensors keeps its optimizer
        extra = {}
        for step in range(5 if cull else 3):
            pipe.train_iteration(opts, step)
            if cull and step == 1:          # what gc_trainer.CullCallback / SplatfactoModel.cull_gaussians do: one row mask over every leaf tensor
                keep = torch.tensor(_KEEP)
                m = pipe._model
                with torch.no_grad():
                    for k in GaussCtrlPipeline._GRAD_KEYS:
                        getattr(m, k).data = getattr(m, k).data[keep].contiguous()
                m._cull_keep = keep
            if cull and step == 3:          # checkpoint round trip (a collective: every rank calls it), restored into a FRESH ShardedAdam
                sd = pipe.sharded_adam_state()
                extra = {"steps": sd["steps"], "n": int(sd["exp_avg"].numel())}
                pipe._sa = None
                pipe.load_sharded_adam_state(sd)
        assert len(stepped) == (5 if cull else 3)
        ret[rank] = ({k: getattr(pipe._model, k).detach().clone() for k in GaussCtrlPipeline._GRAD_KEYS}, list(pipe.datamanager.seen), extra)
    finally:
        dist.destroy_process_group()


_KEEP = [True, False, True, True, False, True, True]          # the cull of the mid-run test: rows 1 and 4 of the 7 Gaussians go


@pytest.mark.parametrize("cull", [False, True])
def test_train_mode_sharded_world2(cull):
    """GaussCtrlPipeline.train_iteration with train_mode "sharded" at world 2 over gloo: each rank renders its own (distinct) view into the flat
    gradient buffer (loss carries 1 / N), dist.ShardedAdam reduce-scatters it, updates its slice of the flat PARAMETER buffer with the groups' lr / eps
    and all-gathers: both ranks end with the parameters of Adam on the mean gradient of the two views; the per-group optimizers of the six leaf
    tensors are never stepped, a group outside them is.
    cull=True: after step 1 every leaf tensor loses rows 1 and 4 (CullCallback's mask).  The replicated path prunes Adam's moments with the
    same mask and keeps counting steps (gc_trainer.py CullCallback); the sharded state must do the same (ShardedAdam.adopt) -- compared against
    exactly that replicated computation.  After step 3 the state goes through sharded_adam_state() / load_sharded_adam_state() into a fresh
    ShardedAdam: step 4 must not notice."""
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_sharded_mode_worker, args=(2, 29650 + os.getpid() % 150 + int(cull), cull, ret), nprocs=2, join=True)
    (p0, seen0, ex0), (p1, seen1, _) = ret[0], ret[1]
    ref = _init_params()
    st = {k: [torch.zeros_like(v), torch.zeros_like(v)] for k, v in ref.items()}
    with torch.no_grad():
        for step in range(5 if cull else 3):
            grads = {k: (2 * v.data * float(seen0[step] + 1)) / 2 + (2 * v.data * float(seen1[step] + 1)) / 2 for k, v in ref.items()}
            for k, v in ref.items():
                _ref_adam(v.data, grads[k], st[k][0], st[k][1], _HYPER[k][0], 0.9, 0.999, _HYPER[k][1], step + 1)
            if cull and step == 1:
                keep = torch.tensor(_KEEP)
                for k, v in ref.items():
                    v.data = v.data[keep].contiguous()
                    st[k] = [st[k][0][keep].contiguous(), st[k][1][keep].contiguous()]
    assert all(a != b for a, b in zip(seen0, seen1)), "the ranks of a step must render distinct views"
    if cull:
        assert ex0 == {"steps": 4, "n": sum(int(v.numel()) for v in ref.values())}
    for k in ref:
        assert torch.equal(p0[k], p1[k]), k
        assert p0[k].shape == ref[k].shape
        assert torch.allclose(p0[k], ref[k].detach(), rtol=1e-6, atol=1e-7), k
