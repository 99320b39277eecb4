"""diagnostic (round 6): leaf gradients of the batched render_views path at 1 M, C = 3 vs the C oracle, per view and per tensor"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from oracle import raster_c
from gaussctrl_amd import gsplat_ops as ops, synthetic as syn
from gaussctrl_amd.camera import camera_to_gsplat
raster_c.build()
DEV = "cuda:0"
BG = np.array([0.1, 0.2, 0.3], np.float32)
N = 1_000_000
K = syn.BEAR_INTRINSICS; W = H = 512
P = syn.make_gaussians(N, seed=0)
c2ws = syn.make_cameras(40, seed=1)
g = np.random.default_rng(5)
v_rgb = g.normal(size=(H, W, 3)).astype(np.float32); v_a = g.normal(size=(H, W)).astype(np.float32)
t = lambda a: torch.tensor(a, device=DEV)
views = [int(v) for v in (sys.argv[1:] or (7, 19, 33))]
og = {}
for v in views:
    o = raster_c.render(P, c2ws[v], K["fx"], K["fy"], K["cx"], K["cy"], W, H, BG, training=True, v_rgb=v_rgb, v_alpha=v_a)
    og[v] = o["grads"]
    tp = {k: t(a).requires_grad_(True) for k, a in P.items()}
    cam = camera_to_gsplat(c2ws[v], K["fx"], K["fy"], K["cx"], K["cy"], W, H)
    for rep in range(2):
        for p in tp.values(): p.grad = None
        rgb, alpha, _ = ops.render_view(tp["means"], tp["scales"], tp["quats"], tp["opacities"], tp["features_dc"], tp["features_rest"], cam, t(BG), False, 3, ops.RenderAux())
        ((rgb * t(v_rgb)).sum() + (alpha * t(v_a)).sum()).backward()
        line = []
        for k in P:
            d = np.abs(tp[k].grad.cpu().numpy().astype(np.float64) - og[v][k]); m = np.abs(og[v][k]).max()
            i = np.unravel_index(d.argmax(), d.shape)
            line.append(f"{k}: {d.max() / m:.2e} (max |ref| {m:.3g}, worst at Gaussian {i[0]})")
        print(f"single view {v} run {rep}: " + "; ".join(line))
    # the worst scales Gaussian: its parameters
    d = np.abs(tp["scales"].grad.cpu().numpy().astype(np.float64) - og[v]["scales"]); i = int(np.unravel_index(d.argmax(), d.shape)[0])
    print(f"   worst scales Gaussian {i}: scales(log) {P['scales'][i]}, ref grad {og[v]['scales'][i]}, got {tp['scales'].grad[i].cpu().numpy()}, radius {o['radii'][i]}, depth {o['depths'][i]:.4f}, xy {o['xys'][i]}")
