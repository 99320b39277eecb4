"""Multi-rank readiness on ONE GPU: two ranks, both on cuda:0, gloo backend, through the real product path
GaussCtrlPipeline.render_reverse -> edit_images -> train_iteration (views sharded v % 2, reference K / V^T replicated OR computed by an
owner rank and broadcast step by step, edited images all-gathered, Gaussian gradients all-reduced), compared with the single-rank run
of the same scene.  (8-GPU RCCL runs are the driver's; this exercises every line of the N > 1 logic on the HIP kernels.)"""
import os

import numpy as np
import pytest
from _margins import within
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
V, H, W, N = 6, 128, 128, 20000


def _build(world, rank, owner, gather=False):
    from gaussctrl_amd import synthetic as syn
    from gaussctrl_amd.gc_model import GaussCtrlModel, GaussCtrlModelConfig
    from gaussctrl_amd.gc_pipeline import GaussCtrlPipeline, GaussCtrlPipelineConfig, SimpleDataManager
    from gaussctrl_amd.ns_compat import Cameras
    dev = "cuda:0"
    P = syn.make_gaussians(N, seed=0, scale_mean=0.03)
    cams = Cameras(syn.make_cameras(V, seed=1), 140.0, 140.0, 64.0, 64.0, W, H)
    model = GaussCtrlModel(GaussCtrlModelConfig(background_color="black"), params=P, device=dev)
    cfg = GaussCtrlPipelineConfig(edit_prompt="a polar bear", reverse_prompt="a bear", chunk_size=2, num_inference_steps=3, dtype="f16",
                                  synthetic_weights=True, ref_bank_owner=owner, ref_bank_allgather=gather,
                                  inflight_chunks=int(os.environ.get("GC_TEST_INFLIGHT", "2")))
    pipe = GaussCtrlPipeline(cfg, dev, world_size=world, local_rank=rank, datamanager=SimpleDataManager(cams, seed=3), model=model)
    return pipe, model


def _h(t):
    import hashlib
    return hashlib.md5(t.detach().float().cpu().contiguous().numpy().tobytes()).hexdigest()[:8]


TRACE = {}


def _run(pipe, model):
    from gaussctrl_amd.gc_config import build_optimizers
    pipe.render_reverse()
    td = pipe.datamanager.train_data
    tr = {"z0": {i: _h(t["z_0_image"]) for i, t in enumerate(td) if "z_0_image" in t},
          "depth": {i: _h(t["depth_image"]) for i, t in enumerate(td) if "depth_image" in t}}
    def hash_bank(bank):                      # called inside edit_images: the pipeline does not keep the bank alive afterwards
        if bank is None:
            return
        import hashlib
        hh = hashlib.md5()
        for key in sorted(bank.store, key=str):
            k, vt = bank.store[key]
            hh.update(k.detach().float().cpu().contiguous().numpy().tobytes()); hh.update(vt.detach().float().cpu().contiguous().numpy().tobytes())
        tr["bank"] = hh.hexdigest()[:8]
        tr["bank_step0"] = {str(key[1]): _h(bank.store[key][0]) for key in sorted(bank.store, key=str) if key[0] == 0}
    pipe.bank_hook = hash_bank
    pipe.edit_images()
    pipe.bank_hook = None
    assert not hasattr(pipe, "_last_bank")
    tr["z0_refs"] = {i: _h(td[i]["z_0_image"]) for i in pipe.ref_indices if "z_0_image" in td[i]}
    tr["depth_refs"] = {i: _h(td[i]["depth_image"]) for i in pipe.ref_indices if "depth_image" in td[i]}
    tr["rgb_refs"] = {i: _h(td[i]["unedited_image"]) for i in pipe.ref_indices if "unedited_image" in td[i]}
    tr["img"] = {i: _h(t["image"]) for i, t in enumerate(td)}
    TRACE.clear(); TRACE.update(tr)
    imgs = torch.stack([t["image"] for t in pipe.datamanager.train_data]).cpu()
    opts = build_optimizers(model)
    import random
    random.seed(11)                                 # same view order on every rank
    losses = [float(pipe.train_iteration(opts, 30000 + s)[0]) for s in range(3)]
    return imgs, losses, model.means.detach().cpu()


def _worker(rank, world, port, owner, ret, gather=False):
    torch.set_num_threads(4)                       # several ranks share the host cores
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gaussctrl_amd.sd import ops as sdops
    sdops.configure(sdops.options_from_env())          # the spawned rank takes GC_BATCH_INVARIANT from the parent test
    try:
        pipe, model = _build(world, rank, owner, gather)
        imgs, losses, means = _run(pipe, model)
        ret[rank] = (imgs.numpy(), losses, means.numpy(), dict(TRACE))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("owner,invariant", [(-1, False), (0, False), (0, True), ("allgather", False), ("allgather", True)])
def test_two_ranks_one_gpu_match_single_rank(owner, invariant, monkeypatch):
    """invariant: batch-invariant kernel planning (sd.ops.BATCH_INVARIANT; the spawned ranks read GC_BATCH_INVARIANT) -- the edited images
    of the 2-rank run are then BIT-identical to the single-rank run, although the chunks hold other views.
    owner "allgather": the reference trajectory itself is sharded by sample (rank r runs both CFG halves of frames {r, r + 2}) and every
    cross-view attention layer all-gathers K / V^T (dist.RefShard) -- bit-identical to the single-rank bank in invariant mode too."""
    gather = owner == "allgather"
    owner = -1 if gather else owner
    from gaussctrl_amd.sd import ops as sdops
    monkeypatch.setenv("GC_BATCH_INVARIANT", "1" if invariant else "0")
    monkeypatch.setattr(sdops, "BATCH_INVARIANT", invariant)
    # poison the caching allocator first: a fresh process hands out zero pages, a long-running one recycled garbage -- a kernel that
    # reads memory nobody wrote would agree with the (fresh) spawned ranks only by luck of the zeros
    junk = [torch.full((n,), float("nan"), device="cuda:0") for n in (1 << 26, 1 << 24, 1 << 22, 1 << 20, 1 << 18)]
    junk += [torch.full((1 << 16,), float("nan"), device="cuda:0") for _ in range(64)]
    del junk
    pipe, model = _build(1, 0, -1)
    ref_imgs, ref_losses, ref_means = _run(pipe, model)
    ref_trace = dict(TRACE)
    del pipe, model
    torch.cuda.empty_cache()
    import hashlib

    def two_ranks(attempt):
        mgr = mp.Manager()
        ret = mgr.dict()
        port = 29500 + (os.getpid() + 37 * attempt) % 400 + (7 if owner >= 0 else 0) + (13 if gather else 0)
        mp.spawn(_worker, args=(2, port, owner, ret, gather), nprocs=2, join=True)
        print(f"CHK attempt {attempt} owner={owner} gather={gather} invariant={invariant} parent {hashlib.md5(ref_imgs.numpy().tobytes()).hexdigest()[:10]} " +
              " ".join(f"rank{r} {hashlib.md5(ret[r][0].tobytes()).hexdigest()[:10]}" for r in range(2)))
        for r in range(2):          # where a run deviates from the single-rank one, stage by stage ('=' same hash, 'X' different)
            t = ret[r][3]
            eq = lambda name: " ".join(f"{i}:{'=' if t[name][i] == ref_trace[name].get(i) else 'X'}" for i in sorted(t[name]))
            print(f"TRACE rank{r}: z0 {eq('z0')} | depth {eq('depth')} | z0_refs {eq('z0_refs')} | depth_refs {eq('depth_refs')} | rgb_refs {eq('rgb_refs')}"
                  f" | bank {'=' if t.get('bank') == ref_trace.get('bank') else 'X'} | img {eq('img')}")
        return ret

    # (Round 4: this test used to deviate in about one run of six -- ONE view's projected Gaussians differed on one rank.  Root cause: packed
    # fp32 instructions (v_pk_mul / add / fma_f32) return wrong results in lanes 48..63 when a wavefront of ANOTHER process issues MFMAs on
    # the same SIMD, which only this two-processes-on-one-GPU configuration produces.  The library is built without them now
    # (csrc/Makefile, tests/test_abi.py, profiles/r04_packed_fp32_fault.txt); the stage-by-stage trace above stays as the diagnostic.)
    ret = two_ranks(0)
    for r in range(2):
        imgs, losses, means = ret[r][:3]
        # every rank ends with ALL edited views (all-gather); f16 kernels with float atomics in the GroupNorm statistics: not bit-equal
        assert imgs.shape == tuple(ref_imgs.shape)
        # not bit-equal: a rank's chunks hold different views than the single-rank chunks, and the GEMM tile / split-K choice
        # depends on the batch size (different fp32 accumulation orders, amplified by the random-weight network); measured on MI355X
        # (3 DDIM steps, f16): mean |diff| 1.7e-3, max 3e-2 on images in [0, 1]
        d = np.abs(imgs - ref_imgs.numpy())
        if invariant:
            assert np.array_equal(imgs, ref_imgs.numpy()), (d.mean(), d.max(), "per-view max |diff|:", [float(d[v].max()) for v in range(d.shape[0])])
        # (allgather: the BANK itself is computed with another batch composition -- 4 samples per rank instead of 8 -- so every view also
        # sees the reference frames' accumulation-order noise through the cross-view terms: twice the bars; the invariant variant of the
        # same run is bit-identical, which is the actual check of the sharded trajectory)
        within("2-rank vs 1-rank edited images: mean |diff|", d.mean(), 1e-2 if gather else 5e-3, strict=True)
        within("2-rank vs 1-rank edited images: max |diff|", d.max(), 0.2 if gather else 0.1, strict=True)
        # same views, averaged gradients of identical renders = the single-rank gradients: same losses / parameters up to atomics noise
        assert np.allclose(losses, ref_losses, rtol=2e-2, atol=1e-3), (losses, ref_losses)
        assert np.abs(means - ref_means.numpy()).max() < 1e-3
    assert np.array_equal(ret[0][0], ret[1][0])              # both ranks hold the same gathered images


# ------------------------------------------------------------------------------------------------ RefShard at world 4 / 8, end to end
def _shard_worker(rank, world, port, ret):
    torch.set_num_threads(4)                       # up to 8 ranks share the host cores
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gaussctrl_amd.dist import RefShard
        from gaussctrl_amd.sd import ops as sdops
        sdops.configure(batch_invariant=True)
        pipe, inputs = _tiny_pipe()
        tr = pipe.begin_ref_bank_sharded(*inputs, RefShard(world, rank), steps=3)
        bank = pipe.advance_ref_bank(tr, None)
        assert bank is not None and bank.mode == "use"
        # the bank a rank ends with is COMPLETE (8 samples per layer and step) and serves an edit chunk
        g = torch.Generator().manual_seed(9)
        lat = torch.randn(2, 4, 8, 8, generator=g).to("cuda:0"); disp = torch.rand(2, 3, 64, 64, generator=g).to("cuda:0")
        out = pipe.edit_chunk_cached(lat, disp, inputs[2], inputs[3], bank, steps=3)
        ret[rank] = ({str(k): (_h(v[0]), _h(v[1]), tuple(v[0].shape)) for k, v in bank.store.items()}, out.cpu().numpy())
    finally:
        dist.destroy_process_group()


def _tiny_pipe():
    """narrow SD-topology UNet / ControlNet (oracle TINY shapes, seeded weights) in f16: the networks of __graft_entry__.smoke()"""
    from gaussctrl_amd.sd.pipeline import DenoisePipeline
    from gaussctrl_amd.sd.weights import prepare
    from oracle import sd15_torch as sd
    cfg = sd.TINY
    uw, cw = sd.make_unet_weights(cfg, 1), sd.make_controlnet_weights(cfg, 2)
    pcfg = dict(block_out_channels=cfg["block_out_channels"], layers_per_block=2, heads=cfg["heads"], cross_dim=cfg["cross_dim"],
                groups=cfg["groups"], attn_levels=cfg["attn_levels"], n_cond_blocks=6)
    dev = "cuda:0"
    pipe = DenoisePipeline(prepare(uw, torch.float16, dev, heads=cfg["heads"]), prepare(cw, torch.float16, dev, heads=cfg["heads"]), None, 20, 5.0)
    pipe.unet.cfg = pcfg; pipe.controlnet.cfg = pcfg
    g = torch.Generator().manual_seed(3)
    lat = torch.randn(4, 4, 8, 8, generator=g).to(dev); disp = torch.rand(4, 3, 64, 64, generator=g).to(dev)
    cn = torch.randn(1, cfg["text_len"], cfg["cross_dim"], generator=g).to(dev); cp = torch.randn(1, cfg["text_len"], cfg["cross_dim"], generator=g).to(dev)
    return pipe, (lat, disp, cn, cp)


@pytest.mark.parametrize("world", [4, 8])
def test_ref_shard_world4_and_8_end_to_end(world, monkeypatch):
    """dist.RefShard with ONE frame (world 4) and ONE SAMPLE = a single CFG half (world 8) per rank -- rep = 1 in _begin, the eps pairs
    all-gathered every DDIM step, a one-row text K / V^T cache -- through DenoisePipeline.begin_ref_bank_sharded on `world` ranks sharing one
    GPU over gloo (small SD-topology networks): every rank ends with the complete bank, BIT-identical (batch-invariant planning) to the bank
    build_ref_bank computes on one rank, and an edit chunk served from it is bit-identical too (advisor finding, round 4: the world-8 path
    had only a CPU layout test)."""
    from gaussctrl_amd.sd import ops as sdops
    monkeypatch.setattr(sdops, "BATCH_INVARIANT", True)
    pipe, inputs = _tiny_pipe()
    ref_bank = pipe.build_ref_bank(*inputs, steps=3)
    g = torch.Generator().manual_seed(9)
    lat = torch.randn(2, 4, 8, 8, generator=g).to("cuda:0"); disp = torch.rand(2, 3, 64, 64, generator=g).to("cuda:0")
    ref_out = pipe.edit_chunk_cached(lat, disp, inputs[2], inputs[3], ref_bank, steps=3).cpu().numpy()
    ref = {str(k): (_h(v[0]), _h(v[1]), tuple(v[0].shape)) for k, v in ref_bank.store.items()}
    del pipe
    torch.cuda.empty_cache()
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_shard_worker, args=(world, 29900 + os.getpid() % 90 + world, ret), nprocs=world, join=True)
    assert len(ref) > 0
    for r in range(world):
        bank, out = ret[r]
        assert set(bank) == set(ref), r
        bad = [k for k in ref if bank[k] != ref[k]]
        assert not bad, (r, len(bad), bad[:3])
        assert np.array_equal(out, ref_out), (r, float(np.abs(out - ref_out).max()))


def test_sharded_adam_slice_update_is_the_fused_adam_kernel():
    """dist.ShardedAdam's product slice update (gc_adam_step on views of the flat parameter / gradient / moment buffers) against train_ops.FusedAdam on
    separate tensors with the same gradients and per-group lr / eps: bit-identical parameters after 3 steps (world 1: the collectives are the gloo
    world-2 tests' business, tests/test_dist_cpu.py); on CPU tensors the slice update raises instead of falling back."""
    from gaussctrl_amd.dist import FlatGrads, FlatParams, ShardedAdam
    from gaussctrl_amd.train_ops import FusedAdam
    dev = "cuda:0"
    shapes = dict(means=(1001, 3), scales=(1001, 3), quats=(1001, 4), opacities=(1001, 1), features_dc=(1001, 3), features_rest=(1001, 15, 3))
    hyper = {"means": (1.6e-4, 1e-15), "scales": (5e-3, 1e-15), "quats": (1e-3, 1e-15), "opacities": (5e-2, 1e-15), "features_dc": (2.5e-3, 1e-15),
             "features_rest": (1.25e-4, 1e-15)}
    g = torch.Generator().manual_seed(0)
    init = {k: torch.randn(*s, generator=g) for k, s in shapes.items()}
    a = {k: torch.nn.Parameter(v.clone().to(dev)) for k, v in init.items()}
    b = {k: torch.nn.Parameter(v.clone().to(dev)) for k, v in init.items()}
    fp = FlatParams(a, 1); fg = FlatGrads(a, pad_to=fp.flat.numel()); sa = ShardedAdam(fp, fg, 1, 0)
    opts = {k: FusedAdam([b[k]], lr=hyper[k][0], eps=hyper[k][1]) for k in b}
    for step in range(3):
        for k in a:
            gk = torch.randn(*shapes[k], generator=g).to(dev)
            fg.views[k].copy_(gk); b[k].grad = gk.clone()
        sa.step(hyper)
        for o in opts.values():
            o.step()
    for k in a:
        assert torch.equal(a[k].detach(), b[k].detach()), k
    cpu = {k: torch.nn.Parameter(v.clone()) for k, v in init.items()}
    fpc = FlatParams(cpu, 1)
    with pytest.raises(Exception):
        ShardedAdam(fpc, FlatGrads(cpu, pad_to=fpc.flat.numel()), 1, 0).step(hyper)
