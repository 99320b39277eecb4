"""oracle/raster_c.py -- TEST INFRASTRUCTURE ONLY.  numpy/ctypes front-end of oracle/raster_ref.c
(the CPU restatement of the gsplat-0.1.3 semantics reached from
/root/reference/gaussctrl/gc_model.py:140-154,166,174-186,191-202; see the C file's header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_raster.so")
TILE = 16


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "raster_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle_raster.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_cumsum_tiles.restype = C.c_int64
    return _lib


def use_threads(on: bool) -> None:
    """bench.py's cpu_baseline leg only: switch to the OpenMP build of the same source (liboracle_raster_mt.so; thread count from
    OMP_NUM_THREADS / all cores).  Its gradient sums use atomics, so the parity tests never load it."""
    global _lib
    if not on:
        _lib = None
        return
    so = os.path.join(_HERE, "liboracle_raster_mt.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE, "liboracle_raster_mt.so"], stdout=subprocess.DEVNULL)
    _lib = C.CDLL(so)
    _lib.orc_cumsum_tiles.restype = C.c_int64


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def project_gaussians(means3d, scales, glob_scale, quats, viewmat, projmat, fx, fy, cx, cy, H, W, tile_bounds,
                      clip_thresh=0.01):
    means3d, scales, quats = _f32(means3d), _f32(scales), _f32(quats)
    viewmat = _f32(np.asarray(viewmat).reshape(-1))
    projmat = _f32(np.asarray(projmat).reshape(-1))
    N = means3d.shape[0]
    cov3d = np.zeros((N, 6), np.float32); xys = np.zeros((N, 2), np.float32); depths = np.zeros(N, np.float32)
    radii = np.zeros(N, np.int32); conics = np.zeros((N, 3), np.float32); nth = np.zeros(N, np.int32)
    lib().orc_project_gaussians_fwd(C.c_int64(N), _p(means3d), _p(scales), C.c_float(glob_scale), _p(quats),
                                    _p(viewmat), _p(projmat), C.c_float(fx), C.c_float(fy), C.c_float(cx),
                                    C.c_float(cy), C.c_int(H), C.c_int(W), C.c_int(tile_bounds[0]),
                                    C.c_int(tile_bounds[1]), C.c_float(clip_thresh), _p(cov3d), _p(xys), _p(depths),
                                    _p(radii), _p(conics), _p(nth))
    return xys, depths, radii, conics, nth, cov3d


def project_gaussians_bwd(means3d, scales, glob_scale, quats, viewmat, projmat, fx, fy, cx, cy, H, W, radii, conics,
                          v_xy, v_depth, v_conic):
    means3d, scales, quats = _f32(means3d), _f32(scales), _f32(quats)
    viewmat = _f32(np.asarray(viewmat).reshape(-1)); projmat = _f32(np.asarray(projmat).reshape(-1))
    N = means3d.shape[0]
    radii, conics = _i32(radii), _f32(conics)
    v_xy, v_conic = _f32(v_xy), _f32(v_conic)
    v_depth = None if v_depth is None else _f32(v_depth)
    vm = np.zeros((N, 3), np.float32); vs = np.zeros((N, 3), np.float32); vq = np.zeros((N, 4), np.float32)
    lib().orc_project_gaussians_bwd(C.c_int64(N), _p(means3d), _p(scales), C.c_float(glob_scale), _p(quats),
                                    _p(viewmat), _p(projmat), C.c_float(fx), C.c_float(fy), C.c_float(cx),
                                    C.c_float(cy), C.c_int(H), C.c_int(W), _p(radii), _p(conics), _p(v_xy),
                                    _p(v_depth), _p(v_conic), _p(vm), _p(vs), _p(vq))
    return vm, vs, vq


def spherical_harmonics(degrees_to_use, viewdirs, coeffs):
    viewdirs, coeffs = _f32(viewdirs), _f32(coeffs)
    N, K = coeffs.shape[0], coeffs.shape[1]
    degree = int(round(K ** 0.5)) - 1
    out = np.zeros((N, 3), np.float32)
    lib().orc_sh_fwd(C.c_int64(N), C.c_int(degree), C.c_int(degrees_to_use), _p(viewdirs), _p(coeffs), _p(out))
    return out


def spherical_harmonics_bwd(degrees_to_use, viewdirs, K, v_colors):
    viewdirs, v_colors = _f32(viewdirs), _f32(v_colors)
    N = viewdirs.shape[0]
    degree = int(round(K ** 0.5)) - 1
    out = np.zeros((N, K, 3), np.float32)
    lib().orc_sh_bwd(C.c_int64(N), C.c_int(degree), C.c_int(degrees_to_use), _p(viewdirs), _p(v_colors), _p(out))
    return out


def bin_and_sort(xys, depths, radii, num_tiles_hit, tile_bounds):
    """-> (cum_tiles_hit, isect_ids_sorted int64[M], gaussian_ids_sorted int32[M], tile_bins int32[T,2])."""
    xys, depths, radii, nth = _f32(xys), _f32(depths), _i32(radii), _i32(num_tiles_hit)
    N = xys.shape[0]
    cum = np.zeros(N, np.int32)
    M = int(lib().orc_cumsum_tiles(C.c_int64(N), _p(nth), _p(cum)))
    keys = np.zeros(M, np.int64); ids = np.zeros(M, np.int32)
    lib().orc_map_gaussian_to_intersects(C.c_int64(N), _p(xys), _p(depths), _p(radii), _p(cum),
                                         C.c_int(tile_bounds[0]), C.c_int(tile_bounds[1]), _p(keys), _p(ids))
    ks = np.zeros(M, np.int64); isd = np.zeros(M, np.int32)
    lib().orc_sort_intersects(C.c_int64(M), _p(keys), _p(ids), _p(ks), _p(isd))
    T = tile_bounds[0] * tile_bounds[1]
    bins = np.zeros((T, 2), np.int32)
    lib().orc_get_tile_bin_edges(C.c_int64(M), _p(ks), C.c_int(T), _p(bins))
    return cum, ks, isd, bins


def rasterize_fwd(xys, conics, colors, opacities, ids_sorted, tile_bins, H, W, tile_bounds, background, extra=None):
    xys, conics, colors, opacities = _f32(xys), _f32(conics), _f32(colors), _f32(np.asarray(opacities).reshape(-1))
    ids_sorted, tile_bins, background = _i32(ids_sorted), _i32(tile_bins), _f32(background)
    extra = None if extra is None else _f32(extra)
    out = np.zeros((H, W, 3), np.float32); out_e = np.zeros((H, W), np.float32) if extra is not None else None
    fT = np.zeros((H, W), np.float32); fi = np.zeros((H, W), np.int32)
    lib().orc_rasterize_fwd(C.c_int(H), C.c_int(W), C.c_int(tile_bounds[0]), C.c_int(tile_bounds[1]),
                            _p(ids_sorted), _p(tile_bins), _p(xys), _p(conics), _p(colors), _p(opacities),
                            _p(extra), _p(background), _p(out), _p(out_e), _p(fT), _p(fi))
    return out, out_e, fT, fi


def rasterize_bwd(xys, conics, colors, opacities, ids_sorted, tile_bins, H, W, tile_bounds, background,
                  final_Ts, final_index, v_out, v_out_alpha=None):
    xys, conics, colors, opacities = _f32(xys), _f32(conics), _f32(colors), _f32(np.asarray(opacities).reshape(-1))
    ids_sorted, tile_bins, background = _i32(ids_sorted), _i32(tile_bins), _f32(background)
    final_Ts, final_index, v_out = _f32(final_Ts), _i32(final_index), _f32(v_out)
    v_out_alpha = None if v_out_alpha is None else _f32(v_out_alpha)
    N = xys.shape[0]
    v_xy = np.zeros((N, 2), np.float32); v_conic = np.zeros((N, 3), np.float32)
    v_col = np.zeros((N, 3), np.float32); v_op = np.zeros(N, np.float32)
    lib().orc_rasterize_bwd(C.c_int(H), C.c_int(W), C.c_int(tile_bounds[0]), C.c_int64(N), _p(ids_sorted),
                            _p(tile_bins), _p(xys), _p(conics), _p(colors), _p(opacities), _p(background),
                            _p(final_Ts), _p(final_index), _p(v_out), _p(v_out_alpha), _p(v_xy), _p(v_conic),
                            _p(v_col), _p(v_op))
    return v_xy, v_conic, v_col, v_op


# ----------------------------------------------------------------------------------------------------------
# GaussCtrlModel.get_outputs restated on top of the C oracle (gc_model.py:57-206), forward + leaf gradients.
def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x.astype(np.float64)))


def camera_glue(c2w, fx, fy, W, H):
    """gc_model.py:97-115 in float32 numpy.  Returns (viewmat[4,4], fullproj[4,4])."""
    import math
    c2w = np.asarray(c2w, np.float32)
    R = c2w[:3, :3] @ np.diag(np.array([1, -1, -1], np.float32))
    T = c2w[:3, 3:4]
    Rinv = R.T
    Tinv = -Rinv @ T
    V = np.eye(4, dtype=np.float32)
    V[:3, :3] = Rinv
    V[:3, 3:4] = Tinv
    fovx = 2 * math.atan(W / (2 * fx)); fovy = 2 * math.atan(H / (2 * fy))
    n, f = 0.001, 1000.0
    t = n * math.tan(0.5 * fovy); b = -t; r = n * math.tan(0.5 * fovx); l = -r
    P = np.array([[2 * n / (r - l), 0, (r + l) / (r - l), 0], [0, 2 * n / (t - b), (t + b) / (t - b), 0],
                  [0, 0, (f + n) / (f - n), -f * n / (f - n)], [0, 0, 1, 0]], np.float32)
    return V, (P @ V).astype(np.float32)


def render(params, c2w, fx, fy, cx, cy, W, H, background, training, sh_degree_to_use=3, v_rgb=None, v_alpha=None):
    """Forward (and, when v_rgb is given, backward to the six leaf tensors) of get_outputs.
    params are numpy arrays: means[N,3], scales(log)[N,3], quats[N,4], opacities(logit)[N,1],
    features_dc[N,3], features_rest[N,15,3]."""
    means = _f32(params["means"]); log_s = _f32(params["scales"]); quats = _f32(params["quats"])
    op_logit = _f32(params["opacities"]).reshape(-1)
    coeffs = _f32(np.concatenate([params["features_dc"][:, None, :], params["features_rest"]], 1))
    K = coeffs.shape[1]
    V, full = camera_glue(c2w, fx, fy, W, H)
    tb = ((W + TILE - 1) // TILE, (H + TILE - 1) // TILE, 1)
    scales = np.exp(log_s).astype(np.float32)
    qn = (quats / np.linalg.norm(quats, axis=-1, keepdims=True)).astype(np.float32)
    xys, depths, radii, conics, nth, _ = project_gaussians(means, scales, 1.0, qn, V[:3], full, fx, fy, cx, cy, H, W, tb)
    bg = _f32(background)
    if int(radii.sum()) == 0:
        return {"rgb": np.broadcast_to(bg, (H, W, 3)).copy()}
    vd = means - np.asarray(c2w, np.float32)[:3, 3]
    vd = (vd / np.linalg.norm(vd, axis=-1, keepdims=True)).astype(np.float32)
    if sh_degree_to_use < 0:           # config.sh_degree == 0: rgbs = sigmoid(colors[:, 0, :])   (gc_model.py:169)
        sh = None
        rgbs = _sigmoid(coeffs[:, 0, :]).astype(np.float32)
    else:
        sh = spherical_harmonics(sh_degree_to_use, vd, coeffs)
        rgbs = np.maximum(sh + np.float32(0.5), 0).astype(np.float32)
    opac = _sigmoid(op_logit).astype(np.float32)
    cum, keys, ids, bins = bin_and_sort(xys, depths, radii, nth, tb)
    img, dep, fT, fi = rasterize_fwd(xys, conics, rgbs, opac, ids, bins, H, W, tb, bg, extra=None if training else depths)
    alpha = (1 - fT)[..., None]
    out = {"rgb": np.minimum(img, 1.0), "accumulation": alpha, "depth": None, "xys": xys, "radii": radii,
           "num_tiles_hit": nth, "isect_ids_sorted": keys, "gaussian_ids_sorted": ids, "tile_bins": bins,
           "final_index": fi, "final_Ts": fT, "M": int(keys.shape[0]), "conics": conics, "depths": depths,
           "rgbs": rgbs}
    if not training:
        d = dep[..., None].copy()
        pos = alpha > 0
        d[pos] = d[pos] / alpha[pos]
        d[~pos] = 1000.0
        out["depth"] = d
    if v_rgb is not None:
        v_img = _f32(v_rgb) * (img <= 1.0)                      # clamp(max=1) backward (gc_model.py:188)
        v_xy, v_conic, v_col, v_op = rasterize_bwd(xys, conics, rgbs, opac, ids, bins, H, W, tb, bg, fT, fi, v_img,
                                                   None if v_alpha is None else _f32(v_alpha).reshape(H, W))
        if sh is None:
            v_coeffs = np.zeros_like(coeffs)
            v_coeffs[:, 0, :] = v_col * rgbs * (1 - rgbs)
        else:
            v_sh = v_col * ((sh + np.float32(0.5)) >= 0)            # clamp(min=0) backward (gc_model.py:167)
            v_coeffs = spherical_harmonics_bwd(sh_degree_to_use, vd, K, v_sh)
        vm, vs, vq = project_gaussians_bwd(means, scales, 1.0, qn, V[:3], full, fx, fy, cx, cy, H, W, radii, conics,
                                           v_xy, None, v_conic)
        # chain through exp(scales), quat normalisation (gc_model.py:142-144), sigmoid (gc_model.py:181)
        qnorm = np.linalg.norm(quats, axis=-1, keepdims=True)
        vq_raw = (vq - qn * (qn * vq).sum(-1, keepdims=True)) / qnorm
        out["grads"] = {"means": vm, "scales": vs * scales, "quats": vq_raw.astype(np.float32),
                        "opacities": (v_op * opac * (1 - opac))[:, None].astype(np.float32),
                        "features_dc": v_coeffs[:, 0, :], "features_rest": v_coeffs[:, 1:, :], "xys": v_xy}
    return out
