"""Stage timings of the splat rasterizer on bench.py's scene (1M Gaussians, 512x512, bear intrinsics).
usage: python scripts/bench_raster.py [n_gaussians]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaussctrl_amd import gsplat_ops as ops, synthetic as syn
from gaussctrl_amd.camera import camera_to_gsplat

dev = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
H = W = 512
K = syn.BEAR_INTRINSICS
P = syn.make_gaussians(N, seed=0)
tp = {k: torch.tensor(v, device=dev) for k, v in P.items()}
cam = camera_to_gsplat(syn.make_cameras(4, seed=1)[1], K["fx"], K["fy"], K["cx"], K["cy"], W, H)
V4 = torch.tensor(np.asarray(cam["viewmat4"], np.float32)); full = torch.tensor(np.asarray(cam["fullproj"], np.float32).reshape(4, 4))
q = tp["quats"] / tp["quats"].norm(dim=-1, keepdim=True)
xys, depths, radii, conics, nth, _ = ops.project_gaussians(tp["means"], torch.exp(tp["scales"]), 1, q, V4[:3], full, K["fx"], K["fy"],
                                                           K["cx"], K["cy"], H, W, cam["tile_bounds"])
tb = cam["tile_bounds"]


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


M, k1, i1, b1, _ = ops.bin_and_sort_gaussians_keys64(N, xys, depths, radii, nth, tb)
M2, k2, i2, b2, _ = ops.bin_and_sort_gaussians(N, xys, depths, radii, nth, tb, want_keys=True)
print(f"N={N} M={M} visible={(radii > 0).sum().item()}  equal: ids={torch.equal(i1, i2)} bins={torch.equal(b1, b2)} keys={torch.equal(k1, k2)}")
print(f"  64-bit key chain (scan+map+rocPRIM sort+bins): {timeit(lambda: ops.bin_and_sort_gaussians_keys64(N, xys, depths, radii, nth, tb)):8.1f} us")
print(f"  two-level binning (depth order + tile passes): {timeit(lambda: ops.bin_and_sort_gaussians(N, xys, depths, radii, nth, tb)):8.1f} us")

# fused render (project + SH + binning + compositing) forward and forward+backward
tq = {k: v.clone().requires_grad_(True) for k, v in tp.items()}
bg = torch.zeros(3, device=dev)
def fwd():
    with torch.no_grad():
        ops.render_view(tq["means"], tq["scales"], tq["quats"], tq["opacities"], tq["features_dc"], tq["features_rest"], cam, bg, True, 3, None)
def fwdbwd():
    for p in tq.values(): p.grad = None
    rgb, alpha, _ = ops.render_view(tq["means"], tq["scales"], tq["quats"], tq["opacities"], tq["features_dc"], tq["features_rest"], cam, bg, False, 3, None)
    rgb.abs().mean().backward()
print(f"  eval render (rgb + depth + alpha):             {timeit(fwd, 10):8.1f} us")
print(f"  train render fwd + bwd:                        {timeit(fwdbwd, 10):8.1f} us")
