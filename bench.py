#!/usr/bin/env python
"""bench.py -- edited views/sec @512x512 (ControlNet denoise + splat render+bwd) on N MI355X GPUs.

One "step" = one chunk of `chunk_size` views pushed through the whole hot path:
  (a) eval render of each view (rgb + depth + alpha, one fused compositing sweep)            [rasterizer fwd]
  (b) 20-step CFG ControlNet+UNet cross-view denoise of the chunk against the 4 reference views' K/V.
      The reference trajectory (4 views) is computed INSIDE the timed region once per ceil(V/chunk) steps,
      i.e. each chunk pays its share of the reference work (V=40, c=3 -> every 14 steps).
  (c) VAE decode of the chunk's edited latents
  (d) one training render of each view, forward + backward to the six Gaussian parameter tensors with an
      L1 loss gradient against the edited image, gradients all-reduced over ranks (RCCL) when N > 1.
Workload (BASELINE.json configs[1]): "bear"-like scene, V=40 views, 4 reference views, chunk_size=3, ~1M synthetic
Gaussians, SD1.5 + ControlNet-depth shapes with seeded random weights (no checkpoints / network), bf16.

Prints ONE JSON line on rank 0.  N > 1: launched by torch.distributed.run, one rank per GPU, views sharded over
ranks (weak scaling: every rank edits its own `chunk_size` views per step; no data-path collective in the denoise
half -- reference K/V are replicated -- and one gradient all-reduce per step in the render half).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

UNET_GFLOP_XVIEW = 1293.3      # per sample-forward, SURVEY.md Appendix B (analytic, 2*MAC)
CN_GFLOP_XVIEW = 430.0         # ControlNet with the weight-0 self term skipped
PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "fp8": 2500.0}      # fp8 run: the dominant kernel (attention) still computes in bf16


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=14)   # one scene of V=40 views at chunk_size 3 = 14 chunks (+ 1 reference trajectory)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--views", type=int, default=None)     # default: 40 (edit, BASELINE configs[1]) / 256 cameras (raster, configs[4])
    ap.add_argument("--chunk-size", type=int, default=3)
    ap.add_argument("--denoise-steps", type=int, default=20)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "fp8"])     # fp8: e4m3 resnet convs on the block-scaled MFMA, bf16 elsewhere
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="edit", choices=["edit", "raster"])
    ap.add_argument("--mask", action="store_true")   # BASELINE configs[3]: edits composited through a (synthetic elliptical) mask, gc_pipeline.py:226-234
    return ap.parse_args()


class GemmProfiler:
    """Brackets every gc_dn_gemm launch with HIP events on the launch stream (one instrumented step)."""

    def __init__(self, dt="BF16"):
        self.rec = []
        self.dt = dt

    def wrap(self, ops):
        self._lin, self._conv, self._att = ops.linear, ops.conv3x3, ops.attention
        prof = self

        def att(q, k, vt, heads, sets, fph, Lk=None, **kw):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = prof._att(q, k, vt, heads, sets, fph, Lk=Lk, **kw)
            e.record()
            lk = k.shape[1] if Lk is None else Lk
            D = q.shape[2] // heads
            fast = D in (40, 80) and len(sets) * ((lk + 63) // 64) >= 4                      # mirrors launch_attn() in dn_attn.hip
            variant = getattr(ops, "KERNEL_VARIANT", {}).get("attn", 0)
            if fast and D == 40 and not (variant & 2):
                name = f"k_attn4<{prof.dt},40,3,{8 if variant & 4 else 4}>"
            else:
                name = f"k_attn3<{prof.dt},{D},{2 if D == 40 else 1},3>" if fast else f"k_attn<{prof.dt},{D},{1 if D == 160 else 2}>"
            prof.rec.append((name, 4.0 * q.shape[0] * q.shape[1] * lk * q.shape[2] * len(sets), s, e))
            return out

        def lin(x, w, *a, **k):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = prof._lin(x, w, *a, **k)
            e.record()
            K = x.shape[-1]
            N = w.shape[0]
            ntw = 5 if (N % 160 == 0 and N % 128 != 0 and not k.get("geglu", False)) else 4      # mirrors plan() in dn_gemm.hip
            prof.rec.append((f"gemm<{prof.dt},linear,BN={32 * ntw}>", 2.0 * (x.numel() // K) * N * K, s, e))
            return out

        def conv(x, w, *a, **k):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = prof._conv(x, w, *a, **k)
            e.record()
            N = w.shape[0]
            ntw = 5 if (N % 160 == 0 and N % 128 != 0) else 4
            mode = 2 if x.shape[-1] % 64 == 0 else 1
            prof.rec.append((f"gemm<{prof.dt},conv3x3{'' if mode == 2 else ' generic'},BN={32 * ntw}>", 2.0 * (out.numel() // out.shape[-1]) * N * w.shape[1], s, e))
            return out

        ops.linear, ops.conv3x3, ops.attention = lin, conv, att

    def unwrap(self, ops):
        ops.linear, ops.conv3x3, ops.attention = self._lin, self._conv, self._att

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for kind, fl, s, e in self.rec:
            d = out.setdefault(kind, {"launches": 0, "flop": 0.0, "ms": 0.0})
            d["launches"] += 1; d["flop"] += fl; d["ms"] += s.elapsed_time(e)
        return out


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # GC_BENCH_ONE_GPU=1 + GC_BENCH_BACKEND=gloo: every rank on cuda:0 over gloo -- exercises the N > 1 code path on a 1-GPU box
    # (a functional check only; its number means nothing).  The driver's multi-GPU runs use neither.
    if os.environ.get("GC_BENCH_ONE_GPU", "0") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("GC_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        dist = None

    from gaussctrl_amd import gsplat_ops as gops, synthetic as syn
    from gaussctrl_amd.camera import camera_to_gsplat
    from gaussctrl_amd.sd import arch, ops as sdops
    from gaussctrl_amd.sd.pipeline import DenoisePipeline
    from gaussctrl_amd.sd.vae import prepare_vae_weights
    from gaussctrl_amd.sd.weights import prepare

    dt = torch.float16 if args.dtype == "f16" else torch.bfloat16
    if args.views is None:
        args.views = 40 if args.workload == "edit" else 256
    c, V, nsteps = args.chunk_size, args.views, args.denoise_steps
    H = W = 512
    K = syn.BEAR_INTRINSICS if args.workload == "edit" else syn.ROUND_INTRINSICS      # config 5 uses fx=fy=540, cx=cy=256

    # ---------------------------------------------------------------- scene, cameras, networks (untimed setup)
    P = syn.make_gaussians(args.gaussians, seed=0)
    params = {k: torch.tensor(v, device=dev, requires_grad=True) for k, v in P.items()}
    cams = syn.make_cameras(V * world, seed=1)
    my_cams = [camera_to_gsplat(cams[rank * V + i], K["fx"], K["fy"], K["cx"], K["cy"], W, H) for i in range(V)]
    bg = torch.zeros(3, device=dev)
    pipe = None
    if args.workload == "edit":
        fold_ln = os.environ.get("GC_DN_FOLD_LN", "0") != "0"          # A/B switches of the round-2 normalisation fusions (default: off)
        usd, csd = arch.random_state_dict(arch.unet_shapes(), 100, dev), arch.random_state_dict(arch.controlnet_shapes(), 200, dev)
        uw = prepare(usd, dt, dev, heads=8, fold_ln=fold_ln)
        cw = prepare(csd, dt, dev, heads=8, fold_ln=fold_ln)
        if args.dtype == "fp8":
            from gaussctrl_amd.sd.weights import add_fp8_convs
            add_fp8_convs(uw, usd, dev); add_fp8_convs(cw, csd, dev)
        del usd, csd
        vw = prepare_vae_weights(arch.random_state_dict(arch.vae_decoder_shapes(), 300, dev), dt, dev)
        pipe = DenoisePipeline(uw, cw, vw, nsteps, 5.0)
        pipe.unet.fuse_stats = pipe.controlnet.fuse_stats = os.environ.get("GC_DN_FUSE_GN", "0") != "0"
        pipe.unet.gn_two_pass = pipe.controlnet.gn_two_pass = os.environ.get("GC_DN_GN2", "0") != "0"
    g = torch.Generator(device=dev).manual_seed(2 + rank)
    ctx_neg = torch.randn(1, 77, 768, device=dev, generator=g)
    ctx_pos = torch.randn(1, 77, 768, device=dev, generator=g)
    z0 = torch.randn(V, 4, 64, 64, device=dev, generator=g)                   # stand-in for the DDIM-inverted latents
    ref_idx = [4, 11, 29, 31]                                                  # gc_pipeline.py:109-113 for V=40
    ref_idx = [min(i, V - 1) for i in ref_idx]
    chunks_per_scene = math.ceil(V / c)
    stats = {"M": [], "n_visible": [], "dev": []}
    state = {"bank": None, "next": None}

    syncfree = os.environ.get("GC_BENCH_SYNCFREE", "1") != "0"
    from gaussctrl_amd.train_ops import l1_ssim_loss
    raster_target = torch.rand(H, W, 3, device=dev, generator=g)      # raster-only workload: a fixed synthetic target image
    edit_mask = torch.tensor(syn.elliptical_mask(H, W, soft=True), device=dev, dtype=torch.float32) if args.mask else None

    grad_buf = {k: torch.zeros_like(v) for k, v in params.items()}        # the chunk's summed leaf gradients (what an optimizer step reads)

    def new_aux():
        aux = gops.RenderAux()
        if syncfree and state.get("cap"):
            aux.m_cap = state["cap"]          # device-side intersection count + capacity: no host round trip in the frame
        return aux

    def note_m(aux):
        if isinstance(aux.M, tuple):
            stats["dev"].append(aux.M)        # (count, overflow) device tensors: read after the timed region
        else:
            stats["M"].append(aux.M)
            state["cap"] = max(state.get("cap") or 0, int(aux.M * 1.3) + 1024)

    def render_eval(i):
        aux = new_aux()
        with torch.no_grad():
            rgb, alpha, depth = gops.render_view(params["means"], params["scales"], params["quats"], params["opacities"],
                                                 params["features_dc"], params["features_rest"], my_cams[i], bg, True, 3, aux)
        note_m(aux)
        return rgb, depth, aux

    def disparity_of(depth):            # gc_pipeline.py:258-266 as one HIP kernel pair -> [H,W,8] control image (3 channels used)
        return sdops.depth_to_disparity(depth, dt)

    def start_ref_trajectory():
        rd = torch.stack([disparity_of(render_eval(i)[1]) for i in ref_idx])
        return pipe.begin_ref_bank(z0[ref_idx], rd, ctx_neg, ctx_pos)

    def step(s):
        """Chunk s of an endless stream of scenes (V views = chunks_per_scene chunks each).  A scene's reference trajectory (4 views
        x 20 DDIM steps, shared by its chunks) is computed while the PREVIOUS scene is edited, 20 / chunks_per_scene DDIM steps
        per chunk, so every step carries exactly its share of the reference work whatever K is."""
        j = s % chunks_per_scene
        views = list(range(j * c, min(V, (j + 1) * c)))          # the last chunk of a scene is short (40 = 13 x 3 + 1), gc_pipeline.py:190
        state["views_done"] = state.get("views_done", 0) + len(views)
        if args.workload == "edit":
            if state["bank"] is None:                # the very first scene (setup): its whole reference trajectory at once
                state["bank"] = pipe.advance_ref_bank(start_ref_trajectory())
            if state["next"] is None:                # the following scene's references start with this scene
                state["next"] = start_ref_trajectory()
            evals = [render_eval(i) for i in views]                                                         # (a)
            disp = torch.stack([disparity_of(e[1]) for e in evals])
            lat = pipe.edit_chunk_cached(z0[views], disp, ctx_neg, ctx_pos, state["bank"])                 # (b)
            edited = pipe.decode(lat)                                                                       # (c)
            quota = (nsteps * (j + 1)) // chunks_per_scene - (nsteps * j) // chunks_per_scene
            done = pipe.advance_ref_bank(state["next"], quota)
            if j == chunks_per_scene - 1:
                assert done is not None
                state["bank"], state["next"] = done, None
        else:
            edited = [None] * len(views)
        for j, i in enumerate(views):                                                                       # (d)
            aux = new_aux()
            aux.grad_into, aux.grad_accumulate = grad_buf, j > 0      # the batch's gradient sum is formed inside the backward kernel
            rgb, alpha, _ = gops.render_view(params["means"], params["scales"], params["quats"], params["opacities"],
                                             params["features_dc"], params["features_rest"], my_cams[i],
                                             torch.rand(3, device=dev), False, 3, aux)
            target = edited[j].permute(1, 2, 0).contiguous() if edited[j] is not None else raster_target
            if args.mask and edited[j] is not None:          # edited inside the mask, the un-edited render outside (one HIP kernel)
                target = sdops.mask_composite(target, evals[j][0].contiguous(), edit_mask)
            loss = l1_ssim_loss(rgb, target, 0.2)             # the product path's loss (fused L1 + SSIM value and gradient kernels)
            loss.backward()
            note_m(aux)
        if dist is not None and not state.get("rank0_only"):      # (the instrumented roofline steps run on rank 0 alone: no collective)
            flat = torch.cat([t.reshape(-1) for t in grad_buf.values()])
            dist.all_reduce(flat)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # setup (untimed, like weight init): two priming chunks so that the caching allocator, the kernel attribute calls and the
    # per-prompt / per-timestep caches are in their steady state before the W warm-up and K timed steps
    g = 0
    for _ in range(2 + args.warmup):
        step(g); g += 1
    barrier()
    v0 = state.get("views_done", 0)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(g); g += 1
    barrier()
    state["views_timed"] = state.get("views_done", 0) - v0
    dt_s = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt_s], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt_s = float(tt.item())
    if stats["dev"]:                      # sync-free frames: counts / overflow flags are read only now
        cnts = torch.stack([a for a, _ in stats["dev"]]).flatten().cpu()
        ovfs = torch.stack([b for _, b in stats["dev"]]).flatten().cpu()
        assert int(ovfs.max()) == 0, "intersection capacity exceeded in a sync-free frame: raise the capacity margin"
        stats["M"] += [int(v) for v in cnts]
    views_done = state["views_timed"] * world
    value = views_done / dt_s

    # ---------------------------------------------------------------- roofline of the dominant kernel (instrumented extra step)
    roof = None
    if args.workload == "edit" and rank == 0:
        prof = GemmProfiler("F16" if args.dtype == "f16" else "BF16")
        prof.wrap(sdops)
        two = pipe.two_streams
        pipe.two_streams = False          # HIP events bracket one kernel only when nothing else shares the GPU: single stream here
        try:
            lat = pipe.edit_chunk_cached(z0[:c], torch.rand(c, 3, H, W, device=dev), ctx_neg, ctx_pos, state["bank"])  # noqa: F841
        finally:
            prof.unwrap(sdops)
            pipe.two_streams = two
        sm = prof.summary()
        dom = max(sm.items(), key=lambda kv: kv[1]["ms"])
        kind, d = dom
        ach = d["flop"] / (d["ms"] * 1e-3) / 1e12
        # HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/r02_attn_traffic.json, scripts/
        # pmc_kernel_traffic.py: FETCH_SIZE / WRITE_SIZE in separate rocprofv3 passes on the UNet's 5-set launch at chunk_size 3)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r02_attn_traffic.json")
        if kind.startswith("k_attn4") and c == 3 and os.path.exists(tpath):
            for kn, v in json.load(open(tpath))["kernels"].items():
                if "k_attn4" in kn and "traffic_MB_per_dispatch" in v:
                    traffic = int(round(v["traffic_MB_per_dispatch"] * 1e6))
        roof = {"bound": "mfma", "kernel": kind + (" (multi-K/V-set flash attention, dn_attn.hip)" if kind.startswith("k_attn") else
                                                    " (k_gemm / k_gemm8 MFMA GEMM and implicit 3x3 conv, variant picked per grid, dn_gemm.hip)"),
                "achieved": round(ach, 2), "peak": PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s",
                "frac": round(ach / PEAK_TFLOPS[args.dtype], 4), "traffic": traffic,
                "launches": d["launches"], "avg_launch_us": round(1e3 * d["ms"] / d["launches"], 2),
                "algorithmic_gflop_per_launch": round(d["flop"] / d["launches"] / 1e9, 3),
                "other": {k: {"launches": v["launches"], "ms": round(v["ms"], 2),
                              "tflops": round(v["flop"] / (v["ms"] * 1e-3) / 1e12, 2)} for k, v in sm.items() if k != kind}}

    if args.workload == "raster" and rank == 0:
        state["rank0_only"] = True
        roof = raster_roofline(args, step, g, stats, H * W)

    # ---------------------------------------------------------------- CPU baseline (oracle, rank 0, bounded sample)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args)

    if rank == 0:
        metric = ("edited views/sec @512x512 (ControlNet denoise + splat render+bwd)" if args.workload == "edit" else
                  "raster-only train-render fwd+bwd views/sec @512x512 (BASELINE configs[4])")
        out = {"metric": metric, "value": round(value, 4),
               "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(1e3 * dt_s / args.steps, 2), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": args.dtype if args.workload == "edit" else "f32", "data": "synthetic",
               "config": {"workload": f"{'masked-edit' if args.mask else 'bear-like'} scene, {V} views/GPU, ref_view_num=4, chunk_size={c}, "
                                      f"{nsteps} DDIM steps, SD1.5+ControlNet-depth shapes (random weights), "
                                      f"{args.gaussians} Gaussians, 512x512" if args.workload == "edit" else
                                      f"raster-only fwd+bwd (+ fused L1+SSIM loss), {args.gaussians} Gaussians, {V} random cameras/GPU, 512x512, fx=fy=540",
                          "views_per_step": c * world, "parallelism": f"views sharded x{world}, reference K/V replicated, grad all-reduce; ControlNet || UNet encoder on 2 HIP streams",
                          "mean_intersections_M": int(np.mean(stats["M"])) if stats["M"] else 0,
                          "ref_trajectory_in_timed_region": bool(args.workload == "edit"),
                          "ref_trajectory_share_per_step": f"{nsteps}/{chunks_per_scene} DDIM steps of the next scene's 4 reference views" if args.workload == "edit" else None},
               "roofline": roof, "cpu_baseline": cpu}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()                      # ranks leave together (rank 0 ran one more instrumented step)
        dist.destroy_process_group()


class LibTimer:
    """Proxy of the ctypes library handle that brackets every compute entry point with HIP events on the launch stream
    (instrumented pass of the raster-only workload)."""
    SKIP = ("gc_last_error_string", "gc_abi_version", "gc_raster_read_count")

    def __init__(self, lib):
        self._l = lib
        self.rec = []

    def __getattr__(self, name):
        f = getattr(self._l, name)
        if not name.startswith("gc_") or name.endswith("_bytes") or name in self.SKIP:
            return f

        def timed(*a):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = f(*a)
            e.record()
            self.rec.append((name, s, e))
            return r
        return timed


# algorithmic HBM bytes of one training render (forward + backward) per stage: coefficients of (N Gaussians, M tile
# intersections, HW pixels) from SURVEY.md 8(d), which add up to N*868 + M*124 + HW*44 -- except k_project_sh_bwd: since round 2 it takes
# the forward colours instead of re-reading the 192-byte SH record (reads 108 B, writes 236 B per Gaussian: 344 instead of 552), so the
# chain total is N*660 + M*124 + HW*44
RASTER_STAGES = {
    "gc_project_sh_fwd": ("k_project_sh_fwd", 280, 0, 0),
    "gc_raster_depth_order": ("binning: depth keys + 4 radix passes + scan (raster_sort.hip)", 0, 0, 0),      # counted with the next row
    "gc_raster_bin_tiles_dev": ("binning: emit + 2 tile radix passes + bins (raster_sort.hip)", 0, 44, 0),
    "gc_raster_bin_tiles": ("binning: emit + 2 tile radix passes + bins (raster_sort.hip)", 0, 44, 0),
    "gc_rasterize_fwd": ("k_rasterize_fwd", 0, 40, 20),
    "gc_raster_finalize": ("k_raster_finalize", 0, 0, 0),
    "gc_l1_ssim_fwd_bwd": ("k_ssim_stats + k_ssim_grad (loss, not in the 8d byte count)", 0, 0, 0),
    "gc_rasterize_bwd": ("k_rasterize_bwd", 36, 40, 24),
    "gc_project_sh_bwd": ("k_project_sh_bwd", 344, 0, 0),
}


def raster_roofline(args, step, g, stats, HW):
    """One instrumented step of the raster-only workload: per-stage HIP-event durations -> achieved algorithmic GB/s per stage
    and for the whole forward + backward chain; `traffic` = HBM bytes per launch from the committed rocprofv3 PMC passes
    (profiles/r02_raster_traffic.json, FETCH_SIZE / WRITE_SIZE collected in separate passes by scripts/pmc_traffic.py) when the
    configuration matches, else null."""
    from gaussctrl_amd import _lib as L
    real = L.lib()
    timer = LibTimer(real)
    n_dev = len(stats["dev"])
    L._lib = timer
    try:
        step(g)
        torch.cuda.synchronize()
    finally:
        L._lib = real
    Ms = [int(a.item()) for a, _ in stats["dev"][n_dev:]] or stats["M"][-8:]        # intersections of the instrumented views
    nviews = max(1, sum(1 for n, _, _ in timer.rec if n == "gc_rasterize_bwd"))
    M = float(np.mean(Ms))
    N = args.gaussians
    per = {}
    for name, s, e in timer.rec:
        per.setdefault(name, []).append(s.elapsed_time(e) * 1e-3)
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r02_raster_traffic.json")
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        traffic = tj.get(str(N))
    stages, tot_b, tot_s = {}, 0.0, 0.0
    for name, (label, cn, cm, chw) in RASTER_STAGES.items():
        if name not in per:
            continue
        secs = float(np.mean(per[name]))
        b = cn * N + cm * M + chw * HW
        tot_b += b; tot_s += secs
        stages[name] = {"kernel": label, "avg_us": round(secs * 1e6, 1), "algorithmic_MB": round(b / 1e6, 1),
                        "GBps": round(b / secs / 1e9, 1) if b else None,
                        "traffic_MB": None if not traffic or name not in traffic else traffic[name]}
    dom = max((k for k in stages if stages[k]["GBps"]), key=lambda k: stages[k]["avg_us"])
    d = stages[dom]
    ach = d["algorithmic_MB"] * 1e6 / (d["avg_us"] * 1e-6) / 1e9
    return {"bound": "hbm", "kernel": f"{d['kernel']} ({dom}, dominant stage of the render fwd+bwd chain)", "achieved": round(ach, 1),
            "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4), "traffic": d["traffic_MB"] and d["traffic_MB"] * 1e6,
            "avg_launch_us": d["avg_us"], "algorithmic_bytes_per_launch": d["algorithmic_MB"] * 1e6,
            "chain": {"algorithmic_MB_per_view": round(tot_b / 1e6, 1), "kernel_us_per_view": round(tot_s * 1e6, 1),
                      "GBps": round(tot_b / tot_s / 1e9, 1), "frac": round(tot_b / tot_s / 8e12, 4), "views_in_sample": nviews,
                      "N": N, "M_mean": int(M), "formula": "N*660 + M*124 + HW*44 bytes per view (SURVEY.md 8d with the SH record no longer re-read in the backward)"},
            "stages": stages}


def cpu_baseline(args):
    """Oracle ("port") timed on the host cores on a BOUNDED sample: ONE CFG ControlNet+UNet cross-view step at the
    reference's CPU-runnable shape (configs[0]: chunk_size=1 -> f = 4 refs + 1 = 5 frames, batch 10) on 32x32 latents,
    scaled to 64x64 latents by the analytic FLOP ratio (conv / linear terms x4, attention core x16
    -> x6.28), + one C-oracle raster eval + train fwd+bwd at N=200k scaled linearly to N.  An edited view at chunk_size 1
    costs the whole f=5 batch for 20 steps."""
    from oracle import raster_c, sd15_torch as sd
    from gaussctrl_amd import synthetic as syn
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    torch.set_num_threads(threads)
    if args.workload == "raster":
        N = 200_000
        P = syn.make_gaussians(N, seed=0)
        c2w = syn.make_cameras(1, seed=1)[0]
        K = syn.ROUND_INTRINSICS
        t0 = time.perf_counter()
        raster_c.render(P, c2w, K["fx"], K["fy"], K["cx"], K["cy"], 512, 512, np.zeros(3, np.float32), training=True,
                        v_rgb=np.ones((512, 512, 3), np.float32))
        t = (time.perf_counter() - t0) * (args.gaussians / N)
        return {"value": round(1.0 / t, 4), "unit": "views/s", "cores": 1, "kind": "port",
                "sample": f"C oracle (oracle/raster_ref.c, 1 thread) train render fwd+bwd at N={N} scaled linearly to N={args.gaussians}: {t:.2f}s per view"}
    with torch.no_grad():
        uw = sd.make_unet_weights(sd.SD15, 100); cw = sd.make_controlnet_weights(sd.SD15, 200)
        f = 5
        lat = torch.randn(f, 4, 32, 32); disp = torch.rand(f, 3, 256, 256)
        cn, cp = torch.randn(1, 77, 768), torch.randn(1, 77, 768)
        t0 = time.perf_counter()
        sd.denoise_chunk(uw, cw, lat, disp, cn, cp, 5.0, 1, sd.SD15, 20)
        t32 = time.perf_counter() - t0
    del uw, cw
    scale = (1293.3 + 479.3) / ((1293.3 - 612.5) / 4 + 612.5 / 16 + (479.3 - 245.1) / 4 + 245.1 / 16)
    t_step = t32 * scale
    N = 200_000
    P = syn.make_gaussians(N, seed=0)
    c2w = syn.make_cameras(1, seed=1)[0]
    K = syn.BEAR_INTRINSICS
    bgc = np.zeros(3, np.float32)
    v_rgb = np.ones((512, 512, 3), np.float32)
    t0 = time.perf_counter()
    raster_c.render(P, c2w, K["fx"], K["fy"], K["cx"], K["cy"], 512, 512, bgc, training=False)
    raster_c.render(P, c2w, K["fx"], K["fy"], K["cx"], K["cy"], 512, 512, bgc, training=True, v_rgb=v_rgb)
    t_raster = (time.perf_counter() - t0) * (args.gaussians / N)
    per_view = 20 * t_step + t_raster
    return {"value": round(1.0 / per_view, 6), "unit": "views/s", "cores": threads, "kind": "port",
            "sample": f"1 of 20 CFG ControlNet+UNet cross-view steps, f=5 (batch 10) on 32x32 latents = {t32:.2f}s, x{scale:.2f} "
                      f"(analytic FLOP ratio) -> {t_step:.1f}s per 64x64 step, x20 steps; C raster eval + train fwd+bwd at "
                      f"N={N} (1 thread) scaled to N={args.gaussians} = {t_raster:.2f}s; VAE decode not included"}


if __name__ == "__main__":
    main()
