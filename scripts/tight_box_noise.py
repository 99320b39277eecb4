"""What does tests/test_raster_gpu.py::test_tight_tile_boxes_same_images_and_gradients measure?  (round-3 review item 1a)

For each of the test's three stress scenes (needles, giants, faint and opaque Gaussians): the fused render + backward with tight tile boxes
twice, with gsplat's boxes twice, and the fp64 gradients of the independent torch restatement (oracle/raster_torch.py's projection / SH
plus its per-tile compositing, evaluated tile by tile ON THE GPU in float64 over the bit-exact gsplat lists).  Prints, per leaf tensor,
the max-norm and relative-L2 distance of every pair as a fraction of max|fp64 gradient|.  TEST INFRASTRUCTURE (imports oracle/).

    python scripts/tight_box_noise.py > profiles/r04_tight_box_noise.txt
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussctrl_amd import gsplat_ops as ops           # noqa: E402
from gaussctrl_amd import synthetic as syn            # noqa: E402
from gaussctrl_amd.camera import camera_to_gsplat     # noqa: E402
from oracle import raster_torch as RT                 # noqa: E402

DEV = "cuda:0"
BG = np.array([0.1, 0.2, 0.3], np.float32)
KEYS = ("means", "scales", "quats", "opacities", "features_dc", "features_rest")


def stress_scene(seed, sm, N=60000):
    P = syn.make_gaussians(N, seed=seed, scale_mean=sm)
    g = np.random.default_rng(seed)
    P["scales"][::7, 0] += 2.0
    P["scales"][::11] += 1.5
    P["opacities"][::5] = g.normal(-5.0, 1.0, size=P["opacities"][::5].shape).astype(np.float32)
    P["opacities"][::13] = 8.0
    return P


def hip_run(P, cam, v, tight):
    tp = {k: torch.tensor(x, device=DEV).requires_grad_(True) for k, x in P.items()}
    aux = ops.RenderAux(); aux.tight_boxes = tight
    rgb, alpha, _ = ops.render_view(tp["means"], tp["scales"], tp["quats"], tp["opacities"], tp["features_dc"], tp["features_rest"],
                                    cam, torch.tensor(BG, device=DEV), False, 3, aux)
    ((rgb * v).sum() + alpha.sum()).backward()
    return rgb.detach(), {k: tp[k].grad.double().cpu().numpy() for k in KEYS}, aux


def fp64_grads(P, c2w, K, v, ids, bins):
    g = RT.render_grads_tiled(P, c2w, K["fx"], K["fy"], K["cx"], K["cy"], K["W"], K["H"], BG, v.cpu(), ids, bins, device=DEV)
    return {k: g[k].numpy() for k in KEYS}


def dist(a, b, ref_max):
    d = np.abs(a - b)
    return d.max() / ref_max, np.sqrt((d * d).sum() / max((b * b).sum(), 1e-300))


def main():
    W, H = 320, 240
    K = dict(fx=300.0, fy=290.0, cx=161.3, cy=118.2, W=W, H=H)
    print("# distances as (max|a-b| / max|fp64|,  ||a-b||_2 / ||b||_2);  T = tight boxes, G = gsplat boxes, 1/2 = first / second run")
    for seed, sm in ((0, 0.02), (1, 0.08), (2, 0.004)):
        P = stress_scene(seed, sm)
        c2w = syn.make_cameras(2, seed=seed + 3)[1]
        cam = camera_to_gsplat(c2w, K["fx"], K["fy"], K["cx"], K["cy"], W, H)
        v = torch.randn(H, W, 3, generator=torch.Generator().manual_seed(seed)).to(DEV)
        _, T1, auxT = hip_run(P, cam, v, True)
        _, T2, _ = hip_run(P, cam, v, True)
        _, G1, auxG = hip_run(P, cam, v, False)
        _, G2, _ = hip_run(P, cam, v, False)
        R = fp64_grads(P, c2w, K, v, auxG.gaussian_ids_sorted, auxG.tile_bins)
        print(f"\nseed {seed} scale_mean {sm}: M tight {int(auxT.M)} / gsplat {int(auxG.M)}")
        print(f"{'tensor':14s} {'max|fp64|':>10s} | {'T1-T2':>19s} | {'G1-G2':>19s} | {'T1-G1':>19s} | {'T1-fp64':>19s} | {'G1-fp64':>19s}")
        for k in KEYS:
            rm = np.abs(R[k]).max()
            cols = [dist(T1[k], T2[k], rm), dist(G1[k], G2[k], rm), dist(T1[k], G1[k], rm), dist(T1[k], R[k], rm), dist(G1[k], R[k], rm)]
            print(f"{k:14s} {rm:10.4g} | " + " | ".join(f"{a:9.2e} {b:9.2e}" for a, b in cols))
            # where the largest T1-G1 difference sits: is it a cancelling sum?
            i = np.unravel_index(np.abs(T1[k] - G1[k]).argmax(), T1[k].shape)
            print(f"{'':14s} worst T1-G1 element {i}: T1 {T1[k][i]:.6g} G1 {G1[k][i]:.6g} fp64 {R[k][i]:.6g}")


if __name__ == "__main__":
    main()
