"""Camera glue of GaussCtrlModel.get_outputs (/root/reference/gaussctrl/gc_model.py:97-121), on the host.

The reference builds viewmat/projmat with ~10 tiny device kernels and three `.item()` syncs per
render; here the 3x4 pose is host data (numpy float32), the 28 matrix floats travel to the GPU as
kernel arguments, and nothing synchronises.
"""
from __future__ import annotations

import math

import numpy as np

TILE = 16


def projection_matrix(znear: float, zfar: float, fovx: float, fovy: float) -> np.ndarray:
    """nerfstudio.models.splatfacto.projection_matrix (imported at gc_model.py:31, called :115)."""
    t = znear * math.tan(0.5 * fovy)
    b = -t
    r = znear * math.tan(0.5 * fovx)
    l = -r
    n, f = znear, zfar
    return np.array([[2 * n / (r - l), 0.0, (r + l) / (r - l), 0.0],
                     [0.0, 2 * n / (t - b), (t + b) / (t - b), 0.0],
                     [0.0, 0.0, (f + n) / (f - n), -1.0 * f * n / (f - n)],
                     [0.0, 0.0, 1.0, 0.0]], dtype=np.float32)


def camera_to_gsplat(c2w, fx: float, fy: float, cx: float, cy: float, W: int, H: int) -> dict:
    """c2w: [3,4] (or [4,4]) OpenGL camera-to-world.  Returns the `cam` dict render_view consumes."""
    c2w = np.asarray(c2w, dtype=np.float32)
    R = c2w[:3, :3] @ np.diag(np.array([1.0, -1.0, -1.0], dtype=np.float32))   # gc_model.py:100-102
    T = c2w[:3, 3:4]
    R_inv = R.T                                                                     # :104-105
    T_inv = -R_inv @ T
    viewmat = np.eye(4, dtype=np.float32)
    viewmat[:3, :3] = R_inv
    viewmat[:3, 3:4] = T_inv
    fovx = 2 * math.atan(W / (2 * fx))                                              # :111-112
    fovy = 2 * math.atan(H / (2 * fy))
    projmat = projection_matrix(0.001, 1000, fovx, fovy)                            # :115
    full = (projmat @ viewmat).astype(np.float32)
    return {"viewmat": viewmat[:3].reshape(-1).tolist(), "fullproj": full.reshape(-1).tolist(),
            "origin": c2w[:3, 3].tolist(), "fx": float(fx), "fy": float(fy), "cx": float(cx), "cy": float(cy),
            "H": int(H), "W": int(W),
            "tile_bounds": ((W + TILE - 1) // TILE, (H + TILE - 1) // TILE, 1),
            "viewmat4": viewmat, "projmat4": projmat}
