"""How many of gsplat's (tile, Gaussian) intersections can never reach alpha >= 1/255 anywhere in their tile?  (The exact tile-level
form of the block mask of raster_composite.hip, evaluated in torch over all M pairs of one view.)
usage (GPU box): python scripts/dead_pairs.py [N]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussctrl_amd import gsplat_ops as ops, synthetic as syn  # noqa: E402
from gaussctrl_amd.camera import camera_to_gsplat  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
dev = "cuda:0"
K = syn.ROUND_INTRINSICS
P = syn.make_gaussians(N, seed=0)
tp = {k: torch.tensor(v, device=dev) for k, v in P.items()}
cam = camera_to_gsplat(syn.make_cameras(4, seed=1)[2], K["fx"], K["fy"], K["cx"], K["cy"], 512, 512)
aux = ops.RenderAux()
with torch.no_grad():
    ops.render_view(tp["means"], tp["scales"], tp["quats"], tp["opacities"], tp["features_dc"], tp["features_rest"], cam,
                    torch.zeros(3, device=dev), False, 3, aux)
ids = aux.gaussian_ids_sorted.long(); bins = aux.tile_bins.long()
M = ids.numel()
tile = torch.repeat_interleave(torch.arange(bins.shape[0], device=dev), bins[:, 1] - bins[:, 0])
tx0 = (tile % 32).float() * 16; ty0 = (tile // 32).float() * 16
xy = aux.xys[ids]; con = aux.conics[ids] if hasattr(aux, "conics") and aux.conics is not None else None
if con is None:
    # recompute conics through the operator surface
    q = tp["quats"] / tp["quats"].norm(dim=-1, keepdim=True)
    V4 = torch.tensor(cam["viewmat4"], device=dev); full = torch.tensor(cam["fullproj"], device=dev).reshape(4, 4)
    xys, depths, radii, conics, nth, _ = ops.project_gaussians(tp["means"], torch.exp(tp["scales"]), 1, q, V4[:3], full, K["fx"], K["fy"], K["cx"], K["cy"], 512, 512, cam["tile_bounds"])
    con = conics[ids]; xy = xys[ids]
op = torch.sigmoid(tp["opacities"][:, 0])[ids]
A, B, C = con[:, 0], con[:, 1], con[:, 2]
tau = torch.log(255.0 * op) * 1.001 + 0.01


def edge_min(a, b, c, e, lo, hi):
    v = torch.minimum(torch.maximum(-b * e / c, lo), hi)
    return 0.5 * (a * e * e + c * v * v) + b * e * v


def box_min(x0, x1, y0, y1):
    dx0, dx1, dy0, dy1 = xy[:, 0] - x1, xy[:, 0] - x0, xy[:, 1] - y1, xy[:, 1] - y0
    inside = (dx0 <= 0) & (dx1 >= 0) & (dy0 <= 0) & (dy1 >= 0)
    m = torch.minimum(torch.minimum(edge_min(A, B, C, dx0, dy0, dy1), edge_min(A, B, C, dx1, dy0, dy1)),
                      torch.minimum(edge_min(C, B, A, dy0, dx0, dx1), edge_min(C, B, A, dy1, dx0, dx1)))
    return torch.where(inside, torch.zeros_like(m), m)


dead_tile = box_min(tx0, tx0 + 15, ty0, ty0 + 15) > tau
live_blocks = 0
for w in range(4):
    bx, by = tx0 + 8 * (w & 1), ty0 + 8 * (w >> 1)
    live_blocks = live_blocks + (~(box_min(bx, bx + 7, by, by + 7) > tau)).float()
print(f"N={N} M={M}: tile-level dead pairs {float(dead_tile.float().mean()):.3f}; live 8x8 blocks per pair {float(live_blocks.mean()):.2f} of 4 "
      f"(per live pair {float(live_blocks[~dead_tile].mean()):.2f})")
