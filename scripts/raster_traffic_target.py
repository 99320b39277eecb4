"""Target of scripts/pmc_traffic.py: `views` training renders (forward + fused L1+SSIM loss + backward) of the raster-only
workload (BASELINE configs[4]: N random Gaussians, random 512x512 cameras, fx=fy=540) followed by known-size calibration copies.
usage: python scripts/raster_traffic_target.py <N> <views> [views per batched launch set, default 1 = one camera per launch set]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussctrl_amd import gsplat_ops as ops, synthetic as syn  # noqa: E402
from gaussctrl_amd.camera import camera_to_gsplat  # noqa: E402
from gaussctrl_amd.train_ops import l1_ssim_loss, l1_ssim_loss_views  # noqa: E402

N = int(sys.argv[1]); views = int(sys.argv[2]); batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = "cuda:0"
K = syn.ROUND_INTRINSICS
P = syn.make_gaussians(N, seed=0)
tp = {k: torch.tensor(v, device=dev, requires_grad=True) for k, v in P.items()}
cams = syn.make_cameras(256, seed=1)
target = torch.rand(512, 512, 3, device=dev)
Ms = []
for i0 in range(0, views if batch > 1 else 0, batch):          # batched views (gsplat_ops.render_views): `batch` cameras per launch set
    cs = [camera_to_gsplat(cams[(37 * i) % 256], K["fx"], K["fy"], K["cx"], K["cy"], 512, 512) for i in range(i0, min(views, i0 + batch))]
    for p in tp.values():
        p.grad = None
    aux = ops.RenderAux()
    rgb, alpha, _ = ops.render_views(tp["means"], tp["scales"], tp["quats"], tp["opacities"], tp["features_dc"], tp["features_rest"],
                                     cs, torch.rand(len(cs), 3, device=dev), False, 3, aux)
    l1_ssim_loss_views(rgb, target.expand(len(cs), -1, -1, -1), 0.2).sum().backward()
    Ms += [int(v) for v in aux.M[0].cpu()]
for i in range(views if batch <= 1 else 0):
    cam = camera_to_gsplat(cams[(37 * i) % 256], K["fx"], K["fy"], K["cx"], K["cy"], 512, 512)
    for p in tp.values():
        p.grad = None
    aux = ops.RenderAux()
    rgb, alpha, _ = ops.render_view(tp["means"], tp["scales"], tp["quats"], tp["opacities"], tp["features_dc"], tp["features_rest"],
                                    cam, torch.rand(3, device=dev), False, 3, aux)
    l1_ssim_loss(rgb, target, 0.2).backward()
    Ms.append(aux.M)
torch.cuda.synchronize()
print("M_per_view", Ms)
lib = C.CDLL(os.path.join(ROOT, "scripts", "ubench", "libhbm_calib.so"))
nbytes = 768 * 1024 * 1024          # > 256 MiB Infinity Cache: the copy really streams from / to HBM
a = torch.empty(nbytes, dtype=torch.uint8, device=dev).random_(0, 255)
b = torch.empty_like(a)
for w in (16, 12, 8, 4):
    for _ in range(2):
        rc = lib.hbm_calib_copy(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_int64(nbytes), w,
                                C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
torch.cuda.synchronize()
print("calib_bytes", nbytes)
