"""A/B of the compositing kernels with and without block culling (raster_composite.hip, -DGC_NO_BLOCK_CULL build loaded through
GC_HIP_LIB): renders one training view (forward + backward) of a random scene and dumps image / alpha / leaf gradients.
usage: python scripts/culling_ab.py dump <out.npz> [N]      (run once per library)
       python scripts/culling_ab.py cmp <a.npz> <b.npz>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if sys.argv[1] == "cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        x, y = a[k].astype(np.float64), b[k].astype(np.float64)
        print(f"{k:14s} bit-identical={np.array_equal(a[k], b[k])}  max|a-b|={np.abs(x - y).max():.3e}  max|a|={np.abs(x).max():.3e}")
    sys.exit(0)

import torch  # noqa: E402
from gaussctrl_amd import gsplat_ops as ops, synthetic as syn  # noqa: E402
from gaussctrl_amd.camera import camera_to_gsplat  # noqa: E402

N = int(sys.argv[3]) if len(sys.argv) > 3 else 300000
dev = "cuda:0"
K = syn.ROUND_INTRINSICS
P = syn.make_gaussians(N, seed=0)
tp = {k: torch.tensor(v, device=dev, requires_grad=True) for k, v in P.items()}
cam = camera_to_gsplat(syn.make_cameras(4, seed=1)[2], K["fx"], K["fy"], K["cx"], K["cy"], 512, 512)
g = torch.Generator(device="cpu").manual_seed(3)
v_rgb = torch.randn(512, 512, 3, generator=g).to(dev); v_a = torch.randn(512, 512, generator=g).to(dev)
aux = ops.RenderAux()
rgb, alpha, _ = ops.render_view(tp["means"], tp["scales"], tp["quats"], tp["opacities"], tp["features_dc"], tp["features_rest"],
                                cam, torch.tensor([0.1, 0.2, 0.3], device=dev), False, 3, aux)
((rgb * v_rgb).sum() + (alpha * v_a).sum()).backward()
np.savez(sys.argv[2], rgb=rgb.detach().cpu().numpy(), alpha=alpha.detach().cpu().numpy(), **{"g_" + k: v.grad.cpu().numpy() for k, v in tp.items()})
print("dumped", sys.argv[2], "M", aux.M if hasattr(aux, "M") else None)
