import sys, numpy as np, torch
sys.path.insert(0, '.')
from gaussctrl_amd import synthetic as syn, gsplat_ops as ops
from gaussctrl_amd.camera import camera_to_gsplat
from oracle import raster_c as rc, raster_torch as rt
BG = np.array([0.1, 0.2, 0.3], np.float32)
def t(a, dt=torch.float32): return torch.tensor(np.asarray(a), dtype=dt, device='cuda:0')
def run(N, W, H, fx, sm, training):
    P = syn.make_gaussians(N, seed=3, scale_mean=sm); c2w = syn.make_cameras(1, seed=4)[0]
    K = dict(fx=fx, fy=fx*0.99, cx=W/2+1.3, cy=H/2-2.1)
    g = np.random.default_rng(2)
    v_rgb = g.normal(size=(H, W, 3)).astype(np.float32); v_a = g.normal(size=(H, W)).astype(np.float32)
    o = rc.render(P, c2w, K["fx"], K["fy"], K["cx"], K["cy"], W, H, BG, training=training, v_rgb=v_rgb, v_alpha=v_a)
    cam = camera_to_gsplat(c2w, K["fx"], K["fy"], K["cx"], K["cy"], W, H)
    tp = {k: t(v).requires_grad_(True) for k, v in P.items()}
    aux = ops.RenderAux()
    rgb, alpha, depth = ops.render_view(tp["means"], tp["scales"], tp["quats"], tp["opacities"], tp["features_dc"], tp["features_rest"], cam, t(BG), not training, 3, aux)
    d = np.abs(rgb.detach().cpu().numpy() - o["rgb"])
    print(f"N={N} {W}x{H}: rgb maxdiff {d.max():.3e}, frac>1e-4: {(d>1e-4).mean():.3e}, frac>1e-5 {(d>1e-5).mean():.3e}; fidx mismatch {(aux.final_index.cpu().numpy()!=o['final_index']).mean():.3e}")
    ((rgb * t(v_rgb)).sum() + (alpha * t(v_a)).sum()).backward()
    if N <= 2000:
        tq = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in P.items()}
        to = rt.get_outputs(tq, torch.tensor(c2w), K["fx"], K["fy"], K["cx"], K["cy"], W, H, torch.tensor(BG), training=training, dtype=torch.float64)
        ((to["rgb"] * torch.tensor(v_rgb, dtype=torch.float64)).sum() + (to["accumulation"][..., 0] * torch.tensor(v_a, dtype=torch.float64)).sum()).backward()
    for k in P:
        got = tp[k].grad.cpu().numpy(); ref = o["grads"][k]
        e = np.abs(got - ref); mx = np.abs(ref).max()
        idx = np.unravel_index(np.argmax(e), e.shape)
        line = f"  {k}: max|ref| {mx:.3e} maxerr {e.max():.3e} at {idx} got {got[idx]:.6e} ref {ref[idx]:.6e} radius {o['radii'][idx[0]]} depth {o['depths'][idx[0]]:.3f}"
        if N <= 2000: line += f" f64 {tq[k].grad.numpy()[idx]:.6e}"
        print(line)
for a in [(7, 33, 17, 40.0, 0.3, False), (3, 40, 24, 50.0, 0.2, True), (5000, 200, 136, 180.0, 0.03, False), (200000, 512, 512, 540.0, 0.01, True)]:
    run(*a)
