"""Synthetic workloads of SURVEY.md section 8(d): random Gaussians, look-at cameras, diffusion inputs.

The reference ships no trained splat checkpoint and no weights can be downloaded, so every BASELINE
config runs on seeded synthetic stand-ins with the reference's shapes.  Camera intrinsics follow
/root/reference/data/bear/transforms.json (fx 539.05 ... 512x512, distortion dropped) or the round
fx=fy=540, cx=cy=256 figures of SURVEY 8d.
"""
from __future__ import annotations

import math

import numpy as np

BEAR_INTRINSICS = dict(fx=539.05, fy=538.17, cx=258.74, cy=239.35, W=512, H=512)
GARDEN_INTRINSICS = dict(fx=585.78, fy=585.51, cx=255.98, cy=253.71, W=512, H=512)
ROUND_INTRINSICS = dict(fx=540.0, fy=540.0, cx=256.0, cy=256.0, W=512, H=512)


def make_gaussians(n: int, seed: int = 0, sh_degree: int = 3, scale_mean: float = 0.01) -> dict:
    """Six splatfacto parameter tensors (numpy float32), SURVEY 8d distribution."""
    g = np.random.default_rng(seed)
    means = g.uniform(-1.0, 1.0, size=(n, 3)).astype(np.float32)
    log_s = np.clip(g.normal(math.log(scale_mean), 0.5, size=(n, 3)), math.log(1e-3), math.log(0.1)).astype(np.float32)
    quats = g.normal(0.0, 1.0, size=(n, 4)).astype(np.float32)
    opac = g.normal(0.0, 2.0, size=(n, 1)).astype(np.float32)
    dc = g.normal(0.0, 0.5, size=(n, 3)).astype(np.float32)
    k = (sh_degree + 1) ** 2 - 1
    rest = g.normal(0.0, 0.05, size=(n, k, 3)).astype(np.float32)
    return {"means": means, "scales": log_s, "quats": quats, "opacities": opac, "features_dc": dc,
            "features_rest": rest}


def look_at_c2w(origin: np.ndarray, target: np.ndarray, up=(0.0, 0.0, 1.0)) -> np.ndarray:
    """OpenGL/nerfstudio camera-to-world [3,4]: columns = right, up, back(+z), origin; camera looks along -z."""
    f = target - origin
    f = f / np.linalg.norm(f)
    upv = np.asarray(up, np.float64)
    r = np.cross(f, upv)
    if np.linalg.norm(r) < 1e-6:
        r = np.cross(f, np.array([1.0, 0.0, 0.0]))
    r = r / np.linalg.norm(r)
    u = np.cross(r, f)
    c2w = np.stack([r, u, -f, origin], axis=1)
    return c2w.astype(np.float32)


def make_cameras(v: int, seed: int = 1, rmin: float = 2.0, rmax: float = 3.0) -> np.ndarray:
    """[V,3,4] camera-to-world matrices on an upper-hemisphere shell looking at the origin (+- 0.2)."""
    g = np.random.default_rng(seed)
    out = []
    for _ in range(v):
        d = g.normal(size=3)
        d[2] = abs(d[2])
        d = d / np.linalg.norm(d)
        rad = g.uniform(rmin, rmax)
        tgt = g.uniform(-0.2, 0.2, size=3)
        out.append(look_at_c2w(d * rad, tgt))
    return np.stack(out, 0)


def elliptical_mask(h: int, w: int, soft: bool = False) -> np.ndarray:
    """Synthetic stand-in for a LangSAM object mask (BASELINE configs[3]): an axis-aligned ellipse covering the
    image centre, [H,W] float32 in {0,1} (or with a smooth 0..1 rim when `soft`)."""
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    r = ((x - 0.5 * w) / (0.32 * w)) ** 2 + ((y - 0.47 * h) / (0.40 * h)) ** 2
    if soft:
        return np.clip((1.15 - r) / 0.3, 0.0, 1.0).astype(np.float32)
    return (r <= 1.0).astype(np.float32)
