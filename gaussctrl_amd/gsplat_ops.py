"""Operator surface of the rasterizer: drop-in for the three gsplat 0.1.3 functions the reference calls.

    project_gaussians    /root/reference/gaussctrl/gc_model.py:140-154  (import :35)
    spherical_harmonics  /root/reference/gaussctrl/gc_model.py:166      (import :32)
    rasterize_gaussians  /root/reference/gaussctrl/gc_model.py:174-186,191-202 (import :36)

Same names, argument order, tuple arity (6-tuple from project_gaussians, gsplat <= 0.1.3) and
autograd behaviour; every call goes through the C ABI of include/gaussctrl_hip.h (hand-written HIP
kernels for gfx950).  There is no CPU / eager fallback: tensors must live on the GPU.

`render_view` is the fused product path used by gaussctrl_amd.gc_model.GaussCtrlModel.get_outputs:
one launch over the 59-float parameter record (exp/normalise/project/SH/sigmoid), one sort, one
compositing sweep for RGB + depth + alpha, and a fused backward to the six leaf tensors.
"""
from __future__ import annotations

import torch

from . import _lib as L

TILE = 16


def num_sh_bases(degree: int) -> int:
    """gsplat.sh.num_sh_bases"""
    if degree == 0:
        return 1
    if degree == 1:
        return 4
    if degree == 2:
        return 9
    if degree == 3:
        return 16
    return 25


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.GaussCtrlHipError("gaussctrl_amd ops need GPU tensors (HIP path only; no CPU fallback)")


def _c(t, dtype=torch.float32):
    return t.detach().to(dtype).contiguous()


def _host_mat(m, n):
    """small camera matrix (device or host tensor / array) -> ctypes float array of n entries."""
    if isinstance(m, torch.Tensor):
        m = m.detach().to("cpu", torch.float32).reshape(-1)
    else:
        import numpy as np
        m = torch.from_numpy(np.asarray(m, dtype="float32").reshape(-1))
    if m.numel() < n:
        raise ValueError(f"camera matrix needs >= {n} entries")
    return L.host_floats(m[:n].tolist())


# --------------------------------------------------------------------------------------------- project
class _ProjectGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3d, scales, glob_scale, quats, viewmat, projmat, fx, fy, cx, cy, H, W, tile_bounds, clip):
        _need_gpu(means3d, scales, quats)
        N = means3d.shape[0]
        dev = means3d.device
        m, s, q = _c(means3d), _c(scales), _c(quats)
        V, P = _host_mat(viewmat, 12), _host_mat(projmat, 16)
        cov3d = torch.empty(N, 6, device=dev); xys = torch.empty(N, 2, device=dev); depths = torch.empty(N, device=dev)
        radii = torch.empty(N, dtype=torch.int32, device=dev); conics = torch.empty(N, 3, device=dev)
        nth = torch.empty(N, dtype=torch.int32, device=dev)
        L.check(L.lib().gc_project_gaussians_fwd(
            L.i64(N), L.ptr(m), L.ptr(s), L.f32(glob_scale), L.ptr(q), V, P, L.f32(fx), L.f32(fy), L.f32(cx), L.f32(cy),
            L.i32(H), L.i32(W), L.i32(tile_bounds[0]), L.i32(tile_bounds[1]), L.f32(clip), L.ptr(cov3d), L.ptr(xys),
            L.ptr(depths), L.ptr(radii), L.ptr(conics), L.ptr(nth), L.stream_ptr()), "gc_project_gaussians_fwd")
        ctx.save_for_backward(m, s, q, radii, conics)
        ctx.cam = (V, P, float(glob_scale), float(fx), float(fy), float(cx), float(cy), int(H), int(W))
        ctx.mark_non_differentiable(radii, nth)
        return xys, depths, radii, conics, nth, cov3d

    @staticmethod
    def backward(ctx, v_xys, v_depths, v_radii, v_conics, v_nth, v_cov3d):
        m, s, q, radii, conics = ctx.saved_tensors
        V, P, glob, fx, fy, cx, cy, H, W = ctx.cam
        N = m.shape[0]
        dev = m.device
        v_xy = _c(v_xys) if v_xys is not None else torch.zeros(N, 2, device=dev)
        v_con = _c(v_conics) if v_conics is not None else torch.zeros(N, 3, device=dev)
        v_dep = _c(v_depths) if v_depths is not None else None
        vm = torch.empty(N, 3, device=dev); vs = torch.empty(N, 3, device=dev); vq = torch.empty(N, 4, device=dev)
        L.check(L.lib().gc_project_gaussians_bwd(
            L.i64(N), L.ptr(m), L.ptr(s), L.f32(glob), L.ptr(q), V, P, L.f32(fx), L.f32(fy), L.f32(cx), L.f32(cy),
            L.i32(H), L.i32(W), L.ptr(radii), L.ptr(conics), L.ptr(v_xy), L.ptr(v_dep), L.ptr(v_con), L.ptr(vm),
            L.ptr(vs), L.ptr(vq), L.stream_ptr()), "gc_project_gaussians_bwd")
        return (vm, vs, None, vq) + (None,) * 10


def project_gaussians(means3d, scales, glob_scale, quats, viewmat, projmat, fx, fy, cx, cy, img_height, img_width,
                      tile_bounds, clip_thresh=0.01):
    """gsplat 0.1.3 signature; returns (xys, depths, radii, conics, num_tiles_hit, cov3d)."""
    return _ProjectGaussians.apply(means3d, scales, glob_scale, quats, viewmat, projmat, fx, fy, cx, cy,
                                   img_height, img_width, tile_bounds, clip_thresh)


# --------------------------------------------------------------------------------------------- SH
class _SphericalHarmonics(torch.autograd.Function):
    @staticmethod
    def forward(ctx, degrees_to_use, viewdirs, coeffs):
        _need_gpu(viewdirs, coeffs)
        N, K = coeffs.shape[0], coeffs.shape[1]
        degree = {1: 0, 4: 1, 9: 2, 16: 3}[K]
        d, c = _c(viewdirs), _c(coeffs)
        out = torch.empty(N, 3, device=coeffs.device)
        L.check(L.lib().gc_sh_fwd(L.i64(N), L.i32(degree), L.i32(degrees_to_use), L.ptr(d), L.ptr(c), L.ptr(out),
                                  L.stream_ptr()), "gc_sh_fwd")
        ctx.save_for_backward(d)
        ctx.meta = (degree, int(degrees_to_use), K)
        return out

    @staticmethod
    def backward(ctx, v_colors):
        (d,) = ctx.saved_tensors
        degree, n, K = ctx.meta
        N = d.shape[0]
        vc = _c(v_colors)
        out = torch.empty(N, K, 3, device=d.device)
        L.check(L.lib().gc_sh_bwd(L.i64(N), L.i32(degree), L.i32(n), L.ptr(d), L.ptr(vc), L.ptr(out), L.stream_ptr()),
                "gc_sh_bwd")
        return None, None, out


def spherical_harmonics(degrees_to_use, viewdirs, coeffs):
    """gsplat.sh.spherical_harmonics(degrees_to_use, viewdirs[N,3], coeffs[N,K,3]) -> colors[N,3]"""
    return _SphericalHarmonics.apply(degrees_to_use, viewdirs, coeffs)


# --------------------------------------------------------------------------------------------- binning
def bin_and_sort_gaussians(N, xys, depths, radii, num_tiles_hit, tile_bounds, want_keys=False, m_cap=None, tile_boxes=None):
    """Two-level binning (csrc/raster_sort.hip): depth-sort the Gaussians, emit (tile, id) in depth order, stable radix
    passes over the tile id, tile bins.  Same gaussian_ids_sorted / tile_bins as gsplat's bin_and_sort_gaussians
    (reference call sites gc_model.py:174-202).  Returns (M, isect_ids_sorted | None, gaussian_ids_sorted, tile_bins,
    cum_tiles_hit_in_depth_order).  One 4-byte readback (M).
    tile_boxes: the packed tight boxes of gc_project_sh_fwd_boxes (num_tiles_hit must be the counts it wrote): the emission walks
    them instead of the boxes recomputed from xys / radii -- fewer pairs, bit-identical images (fused product path)."""
    lib = L.lib()
    dev = xys.device
    st = L.stream_ptr()
    T = tile_bounds[0] * tile_bounds[1]
    order = torch.empty(N, dtype=torch.int32, device=dev)
    cum = torch.empty(N, dtype=torch.int32, device=dev)
    cnt = torch.empty(1, dtype=torch.int32, device=dev)
    wb = int(lib.gc_raster_depth_order_workspace_bytes(L.i64(N)))
    ws = torch.empty(wb, dtype=torch.uint8, device=dev)
    L.check(lib.gc_raster_depth_order(L.i64(N), L.ptr(depths), L.ptr(radii), L.ptr(num_tiles_hit), L.ptr(order), L.ptr(cum),
                                      L.ptr(cnt), L.ptr(ws), L.C.c_size_t(wb), st), "gc_raster_depth_order")
    bins = torch.empty(T, 2, dtype=torch.int32, device=dev)
    if m_cap is not None:
        # sync-free: buffers sized for the caller's capacity, count and overflow flag stay on the device.
        # Returns M = (count tensor, overflow tensor) instead of a host int.
        M = int(m_cap)
        ids_s = torch.empty(M, dtype=torch.int32, device=dev)
        keys_s = torch.empty(M, dtype=torch.int64, device=dev) if want_keys else None
        ovf = torch.empty(1, dtype=torch.int32, device=dev)
        bb = int(lib.gc_raster_bin_workspace_bytes(L.i64(M)))
        bws = torch.empty(bb, dtype=torch.uint8, device=dev)
        if tile_boxes is not None:
            L.check(lib.gc_raster_bin_tiles_boxes(L.i64(N), L.i64(M), L.ptr(cnt), L.ptr(ovf), L.ptr(order), L.ptr(cum), L.ptr(tile_boxes),
                                                  L.ptr(depths), L.i32(tile_bounds[0]), L.i32(tile_bounds[1]), L.ptr(ids_s), L.ptr(bins),
                                                  L.ptr(keys_s), L.ptr(bws), L.C.c_size_t(bb), st), "gc_raster_bin_tiles_boxes")
        else:
            L.check(lib.gc_raster_bin_tiles_dev(L.i64(N), L.i64(M), L.ptr(cnt), L.ptr(ovf), L.ptr(order), L.ptr(cum), L.ptr(xys),
                                                L.ptr(depths), L.ptr(radii), L.i32(tile_bounds[0]), L.i32(tile_bounds[1]), L.ptr(ids_s),
                                                L.ptr(bins), L.ptr(keys_s), L.ptr(bws), L.C.c_size_t(bb), st), "gc_raster_bin_tiles_dev")
        return (cnt, ovf), keys_s, ids_s, bins, cum
    m_host = L.C.c_int32(0)
    L.check(lib.gc_raster_read_count(L.ptr(cnt), L.C.byref(m_host), st), "gc_raster_read_count")
    M = int(m_host.value)
    ids_s = torch.empty(M, dtype=torch.int32, device=dev)
    keys_s = torch.empty(M, dtype=torch.int64, device=dev) if want_keys else None
    bb = int(lib.gc_raster_bin_workspace_bytes(L.i64(M)))
    bws = torch.empty(bb, dtype=torch.uint8, device=dev)
    if tile_boxes is not None:
        L.check(lib.gc_raster_bin_tiles_boxes(L.i64(N), L.i64(M), None, None, L.ptr(order), L.ptr(cum), L.ptr(tile_boxes), L.ptr(depths),
                                              L.i32(tile_bounds[0]), L.i32(tile_bounds[1]), L.ptr(ids_s), L.ptr(bins), L.ptr(keys_s),
                                              L.ptr(bws), L.C.c_size_t(bb), st), "gc_raster_bin_tiles_boxes")
    else:
        L.check(lib.gc_raster_bin_tiles(L.i64(N), L.i64(M), L.ptr(order), L.ptr(cum), L.ptr(xys), L.ptr(depths), L.ptr(radii),
                                        L.i32(tile_bounds[0]), L.i32(tile_bounds[1]), L.ptr(ids_s), L.ptr(bins), L.ptr(keys_s),
                                        L.ptr(bws), L.C.c_size_t(bb), st), "gc_raster_bin_tiles")
    return M, keys_s, ids_s, bins, cum


def bin_and_sort_gaussians_keys64(N, xys, depths, radii, num_tiles_hit, tile_bounds):
    """gsplat-shaped chain: cumsum -> map -> 64-bit key sort -> tile bins (kept for API parity / cross-checks).
    Returns (M, isect_ids_sorted, gaussian_ids_sorted, tile_bins, cum_tiles_hit)."""
    lib = L.lib()
    dev = xys.device
    st = L.stream_ptr()
    T = tile_bounds[0] * tile_bounds[1]
    cum = torch.empty(N, dtype=torch.int32, device=dev)
    cnt = torch.empty(1, dtype=torch.int32, device=dev)
    ws_bytes = lib.gc_raster_scan_workspace_bytes(L.i64(N))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    L.check(lib.gc_raster_scan_tiles(L.i64(N), L.ptr(num_tiles_hit), L.ptr(cum), L.ptr(cnt), L.ptr(ws),
                                     L.C.c_size_t(ws_bytes), st), "gc_raster_scan_tiles")
    m_host = L.C.c_int32(0)
    L.check(lib.gc_raster_read_count(L.ptr(cnt), L.C.byref(m_host), st), "gc_raster_read_count")
    M = int(m_host.value)
    bins = torch.empty(T, 2, dtype=torch.int32, device=dev)
    keys = torch.empty(M, dtype=torch.int64, device=dev); ids = torch.empty(M, dtype=torch.int32, device=dev)
    keys_s = torch.empty(M, dtype=torch.int64, device=dev); ids_s = torch.empty(M, dtype=torch.int32, device=dev)
    if M > 0:
        L.check(lib.gc_raster_map_intersects(L.i64(N), L.i64(M), L.ptr(xys), L.ptr(depths), L.ptr(radii), L.ptr(cum),
                                             L.i32(tile_bounds[0]), L.i32(tile_bounds[1]), L.ptr(keys), L.ptr(ids), st),
                "gc_raster_map_intersects")
        sb = lib.gc_raster_sort_workspace_bytes(L.i64(M), L.i32(T))
        sws = torch.empty(max(int(sb), 1), dtype=torch.uint8, device=dev)
        L.check(lib.gc_raster_sort_intersects(L.i64(M), L.i32(T), L.ptr(keys), L.ptr(ids), L.ptr(keys_s), L.ptr(ids_s),
                                              L.ptr(sws), L.C.c_size_t(int(sb)), st), "gc_raster_sort_intersects")
    L.check(lib.gc_raster_tile_bins(L.i64(M), L.i32(T), L.ptr(keys_s), L.ptr(bins), st), "gc_raster_tile_bins")
    return M, keys_s, ids_s, bins, cum


def _rasterize_fwd(H, W, tb, ids_s, bins, xys, conics, colors, opac, extra, background):
    dev = xys.device
    out = torch.empty(H, W, 3, device=dev)
    out_e = torch.empty(H, W, device=dev) if extra is not None else None
    fT = torch.empty(H, W, device=dev)
    fi = torch.empty(H, W, dtype=torch.int32, device=dev)
    L.check(L.lib().gc_rasterize_fwd(L.i32(H), L.i32(W), L.i32(tb[0]), L.i32(tb[1]), L.ptr(ids_s), L.ptr(bins),
                                     L.ptr(xys), L.ptr(conics), L.ptr(colors), L.ptr(opac), L.ptr(extra),
                                     L.ptr(background), L.ptr(out), L.ptr(out_e), L.ptr(fT), L.ptr(fi), L.stream_ptr()),
            "gc_rasterize_fwd")
    return out, out_e, fT, fi


def _rasterize_bwd(H, W, tb, N, ids_s, bins, xys, conics, colors, opac, background, fT, fi, v_out, v_alpha, pre_clamp=None):
    """pre_clamp: the un-clamped composited image -> the clamp(max=1) backward of gc_model.py:188 happens in the kernel's pixel load"""
    dev = xys.device
    v_xy = torch.zeros(N, 2, device=dev); v_conic = torch.zeros(N, 3, device=dev)
    v_col = torch.zeros(N, 3, device=dev); v_op = torch.zeros(N, device=dev)
    if pre_clamp is None:
        L.check(L.lib().gc_rasterize_bwd(L.i32(H), L.i32(W), L.i32(tb[0]), L.i32(tb[1]), L.i64(N), L.ptr(ids_s), L.ptr(bins),
                                         L.ptr(xys), L.ptr(conics), L.ptr(colors), L.ptr(opac), L.ptr(background), L.ptr(fT),
                                         L.ptr(fi), L.ptr(v_out), L.ptr(v_alpha), L.ptr(v_xy), L.ptr(v_conic), L.ptr(v_col),
                                         L.ptr(v_op), L.stream_ptr()), "gc_rasterize_bwd")
    else:
        L.check(L.lib().gc_rasterize_bwd_clamped(L.i32(H), L.i32(W), L.i32(tb[0]), L.i32(tb[1]), L.i64(N), L.ptr(ids_s), L.ptr(bins),
                                                 L.ptr(xys), L.ptr(conics), L.ptr(colors), L.ptr(opac), L.ptr(background), L.ptr(fT),
                                                 L.ptr(fi), L.ptr(v_out), L.ptr(v_alpha), L.ptr(pre_clamp), L.ptr(v_xy), L.ptr(v_conic),
                                                 L.ptr(v_col), L.ptr(v_op), L.stream_ptr()), "gc_rasterize_bwd_clamped")
    return v_xy, v_conic, v_col, v_op


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xys, depths, radii, conics, num_tiles_hit, colors, opacity, H, W, background, return_alpha):
        _need_gpu(xys, colors)
        if colors.dim() != 2 or colors.shape[1] != 3:
            raise ValueError("rasterize_gaussians: colors must be [N,3] (the reference only uses 3 channels)")
        N = xys.shape[0]
        dev = xys.device
        tb = ((W + TILE - 1) // TILE, (H + TILE - 1) // TILE, 1)
        x, d, r, c, nth = _c(xys), _c(depths), _c(radii, torch.int32), _c(conics), _c(num_tiles_hit, torch.int32)
        col, op = _c(colors), _c(opacity).reshape(-1)
        bg = _c(background) if background is not None else torch.ones(3, device=dev)
        M, keys_s, ids_s, bins, _ = bin_and_sort_gaussians(N, x, d, r, nth, tb)
        out, _, fT, fi = _rasterize_fwd(H, W, tb, ids_s, bins, x, c, col, op, None, bg)
        ctx.save_for_backward(ids_s, bins, x, c, col, op, bg, fT, fi)
        ctx.meta = (H, W, tb, N)
        if return_alpha:
            return out, 1 - fT
        return out

    @staticmethod
    def backward(ctx, v_out, v_alpha=None):
        ids_s, bins, x, c, col, op, bg, fT, fi = ctx.saved_tensors
        H, W, tb, N = ctx.meta
        vo = _c(v_out)
        va = _c(v_alpha) if v_alpha is not None else None
        v_xy, v_conic, v_col, v_op = _rasterize_bwd(H, W, tb, N, ids_s, bins, x, c, col, op, bg, fT, fi, vo, va)
        return v_xy, None, None, v_conic, None, v_col, v_op[:, None], None, None, None, None


def rasterize_gaussians(xys, depths, radii, conics, num_tiles_hit, colors, opacity, img_height, img_width,
                        background=None, return_alpha=False):
    """gsplat 0.1.3 signature; out_img[H,W,3] (, out_alpha[H,W])."""
    return _RasterizeGaussians.apply(xys, depths, radii, conics, num_tiles_hit, colors, opacity, img_height, img_width,
                                     background, return_alpha)


# --------------------------------------------------------------------------------------------- fused path
TIGHT_BOXES = True          # default of RenderAux.tight_boxes for the fused render_view path


class RenderAux:
    """Side outputs of render_view (non-differentiable state the model keeps, gc_model.py:140,159-160)."""
    xys = None
    radii = None
    num_tiles_hit = None
    depths = None
    xys_grad = None
    M = 0
    gaussian_ids_sorted = None
    tile_bins = None
    final_index = None
    isect_ids_sorted = None
    m_cap = None          # set to an intersection capacity for the sync-free path: M then is (count, overflow) device tensors
    # Gradient accumulation inside the backward kernel: grad_into = {"means", "scales", "quats", "opacities", "features_dc",
    # "features_rest"} -> fp32 tensors shaped like the parameters.  The backward then writes (grad_accumulate = False) or adds
    # (True) the six leaf gradients there and hands autograd None for them -- no separate read-add-write pass per tensor and view.
    grad_into = None
    grad_accumulate = False
    # Tight tile boxes (round 3): bin a Gaussian only into the tiles of the bounding box of its alpha >= 1/255 ellipse (inside gsplat's
    # 3-sigma box): ~31 % fewer (tile, Gaussian) pairs, images and gradients bit-identical.  num_tiles_hit / M / gaussian_ids_sorted /
    # tile_bins then describe the SHORTER lists; set False for the lists gsplat would build (the gsplat-shaped operators always do).
    tight_boxes = None            # None: module default TIGHT_BOXES
    tile_boxes = None
    # render_views (round 6): True = the gather-free depth order (gc_raster_order_boxes_views + gc_raster_bin_sorted_views: the packed boxes ride
    # through the radix passes, culled Gaussians drop out in the first pass); False = the round-5 chain (same lists bit for bit; A/B and cross-checks)
    sorted_boxes = True


class _RenderView(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means, log_scales, quats, opacities, features_dc, features_rest, cam, background, want_depth,
                sh_degree_to_use, aux):
        _need_gpu(means)
        lib = L.lib()
        st = L.stream_ptr()
        N = means.shape[0]
        dev = means.device
        H, W = cam["H"], cam["W"]
        tb = ((W + TILE - 1) // TILE, (H + TILE - 1) // TILE, 1)
        K = features_rest.shape[1] + 1
        sh_degree = {1: 0, 4: 1, 9: 2, 16: 3}[K]
        m, ls, q = _c(means), _c(log_scales), _c(quats)
        op, dc, rest = _c(opacities).reshape(-1), _c(features_dc), _c(features_rest)
        V, P, O = L.host_floats(cam["viewmat"]), L.host_floats(cam["fullproj"]), L.host_floats(cam["origin"])
        xys = torch.empty(N, 2, device=dev); depths = torch.empty(N, device=dev)
        radii = torch.empty(N, dtype=torch.int32, device=dev); conics = torch.empty(N, 3, device=dev)
        nth = torch.empty(N, dtype=torch.int32, device=dev)
        rgbs = torch.empty(N, 3, device=dev); opac = torch.empty(N, device=dev)
        tight = TIGHT_BOXES if (aux is None or aux.tight_boxes is None) else bool(aux.tight_boxes)
        tight = tight and tb[0] <= 255 and tb[1] <= 255
        boxes = torch.empty(N, dtype=torch.int32, device=dev) if tight else None
        if tight:
            L.check(lib.gc_project_sh_fwd_boxes(
                L.i64(N), L.ptr(m), L.ptr(ls), L.ptr(q), L.ptr(op), L.ptr(dc), L.ptr(rest), L.i32(sh_degree),
                L.i32(sh_degree_to_use), V, P, O, L.f32(cam["fx"]), L.f32(cam["fy"]), L.f32(cam["cx"]), L.f32(cam["cy"]),
                L.i32(H), L.i32(W), L.i32(tb[0]), L.i32(tb[1]), L.f32(0.01), L.ptr(xys), L.ptr(depths), L.ptr(radii),
                L.ptr(conics), L.ptr(nth), L.ptr(rgbs), L.ptr(opac), L.ptr(boxes), st), "gc_project_sh_fwd_boxes")
        else:
            L.check(lib.gc_project_sh_fwd(
                L.i64(N), L.ptr(m), L.ptr(ls), L.ptr(q), L.ptr(op), L.ptr(dc), L.ptr(rest), L.i32(sh_degree),
                L.i32(sh_degree_to_use), V, P, O, L.f32(cam["fx"]), L.f32(cam["fy"]), L.f32(cam["cx"]), L.f32(cam["cy"]),
                L.i32(H), L.i32(W), L.i32(tb[0]), L.i32(tb[1]), L.f32(0.01), L.ptr(xys), L.ptr(depths), L.ptr(radii),
                L.ptr(conics), L.ptr(nth), L.ptr(rgbs), L.ptr(opac), st), "gc_project_sh_fwd")
        M, keys_s, ids_s, bins, _ = bin_and_sort_gaussians(N, xys, depths, radii, nth, tb, m_cap=None if aux is None else aux.m_cap,
                                                           tile_boxes=boxes)
        bg = _c(background)
        extra = depths if want_depth else None
        img, dep, fT, fi = _rasterize_fwd(H, W, tb, ids_s, bins, xys, conics, rgbs, opac, extra, bg)
        alpha = torch.empty(H, W, device=dev)
        if any(ctx.needs_input_grad[:6]):        # differentiable: the raw image stays for the clamp's backward, no copy pass
            pre_clamp, img = img, torch.empty_like(img)
            L.check(lib.gc_raster_finalize_into(L.i64(H * W), L.ptr(pre_clamp), L.ptr(img), L.ptr(dep), L.ptr(fT), L.ptr(alpha), st),
                    "gc_raster_finalize_into")
        else:
            pre_clamp = None
            L.check(lib.gc_raster_finalize(L.i64(H * W), L.ptr(img), L.ptr(dep), L.ptr(fT), L.ptr(alpha), st),
                    "gc_raster_finalize")
        if aux is not None:
            aux.xys, aux.radii, aux.num_tiles_hit, aux.M, aux.depths = xys, radii, nth, M, depths
            aux.tile_boxes = boxes
            aux.gaussian_ids_sorted, aux.tile_bins, aux.final_index, aux.isect_ids_sorted = ids_s, bins, fi, keys_s
            aux.xys_grad = None
        ctx.save_for_backward(m, ls, q, op, dc, rest, radii, conics, xys, rgbs, opac, ids_s, bins, bg, fT, fi, pre_clamp)
        ctx.meta = (cam, tb, N, sh_degree, int(sh_degree_to_use), V, P, O)
        ctx.aux = aux
        ctx.mark_non_differentiable(*( [dep] if dep is not None else []))
        if dep is None:
            dep = torch.empty(0, device=dev)
        return img, alpha, dep

    @staticmethod
    def backward(ctx, v_img, v_alpha, v_dep):
        (m, ls, q, op, dc, rest, radii, conics, xys, rgbs, opac, ids_s, bins, bg, fT, fi, pre_clamp) = ctx.saved_tensors
        cam, tb, N, sh_degree, n_use, V, P, O = ctx.meta
        H, W = cam["H"], cam["W"]
        dev = m.device
        vo = _c(v_img) if v_img is not None else torch.zeros(H, W, 3, device=dev)
        va = _c(v_alpha) if v_alpha is not None else None
        # (clamp(max=1) backward, gc_model.py:188: applied by the kernel when it loads the pixel's gradient)
        v_xy, v_conic, v_col, v_op = _rasterize_bwd(H, W, tb, N, ids_s, bins, xys, conics, rgbs, opac, bg, fT, fi, vo, va, pre_clamp)
        if ctx.aux is not None:
            ctx.aux.xys_grad = v_xy
        into = ctx.aux.grad_into if ctx.aux is not None else None
        if into is not None:
            vm, vls, vq, vop, vdc, vrest = (into[k] for k in ("means", "scales", "quats", "opacities", "features_dc", "features_rest"))
            for t, ref in ((vm, m), (vls, ls), (vq, q), (vdc, dc), (vrest, rest)):
                assert t.is_contiguous() and t.dtype == torch.float32 and t.shape == ref.shape and t.device == dev
            assert vop.is_contiguous() and vop.dtype == torch.float32 and vop.numel() == N and vop.device == dev
            fn = L.lib().gc_project_sh_bwd_accumulate if ctx.aux.grad_accumulate else L.lib().gc_project_sh_bwd
        else:
            vm = torch.empty(N, 3, device=dev); vls = torch.empty(N, 3, device=dev); vq = torch.empty(N, 4, device=dev)
            vop = torch.empty(N, device=dev); vdc = torch.empty(N, 3, device=dev)
            vrest = torch.empty(rest.shape, device=dev)
            fn = L.lib().gc_project_sh_bwd
        L.check(fn(
            L.i64(N), L.ptr(m), L.ptr(ls), L.ptr(q), L.ptr(op), L.ptr(rgbs), L.i32(sh_degree), L.i32(n_use),
            V, P, O, L.f32(cam["fx"]), L.f32(cam["fy"]), L.f32(cam["cx"]), L.f32(cam["cy"]), L.i32(H), L.i32(W),
            L.ptr(radii), L.ptr(conics), L.ptr(v_xy), L.ptr(v_conic), L.ptr(v_col), L.ptr(v_op), L.ptr(vm), L.ptr(vls),
            L.ptr(vq), L.ptr(vop), L.ptr(vdc), L.ptr(vrest), L.stream_ptr()), "gc_project_sh_bwd")
        if into is not None:
            return (None,) * 11
        return vm, vls, vq, vop[:, None], vdc, vrest, None, None, None, None, None


def render_view(means, log_scales, quats, opacities, features_dc, features_rest, cam: dict, background, want_depth: bool,
                sh_degree_to_use: int = 3, aux: RenderAux | None = None):
    """Fused get_outputs core.  cam: dict(viewmat[12], fullproj[16], origin[3], fx, fy, cx, cy, H, W) of HOST floats.
    Returns (rgb[H,W,3] clamped to <=1, alpha[H,W], depth[H,W] (normalised, 1000 where alpha==0) or empty)."""
    return _RenderView.apply(means, log_scales, quats, opacities, features_dc, features_rest, cam, background, want_depth,
                             sh_degree_to_use, aux)


# --------------------------------------------------------------------------------------------- batched views (round 5)
def _cams_host(cams: list):
    """list of `cam` dicts (camera_to_gsplat) -> ctypes float array [C][GC_VIEW_CAM_FLOATS = 35]"""
    flat = []
    for cam in cams:
        flat += list(cam["viewmat"])[:12] + list(cam["fullproj"])[:16] + list(cam["origin"])[:3] + [cam["fx"], cam["fy"], cam["cx"], cam["cy"]]
    return L.host_floats(flat)


class _RenderViews(torch.autograd.Function):
    """C cameras of one scene through ONE set of launches (include/gaussctrl_hip.h "Batched views"): the parameter record is read once
    per batch by the projection / SH kernel and once by its backward, the sort / binning / compositing kernels run with the view as a
    grid dimension.  Per view the result is bit-identical to _RenderView (same device code); the leaf gradients are the sum over the
    views.  The reference has no such call: it renders one camera per get_outputs (gc_pipeline.py:124-130, gc_trainer.py:186-201)."""

    @staticmethod
    def forward(ctx, means, log_scales, quats, opacities, features_dc, features_rest, cams, backgrounds, want_depth, sh_degree_to_use, aux):
        _need_gpu(means)
        lib = L.lib()
        st = L.stream_ptr()
        N = means.shape[0]
        dev = means.device
        C = len(cams)
        H, W = cams[0]["H"], cams[0]["W"]
        assert all(c["H"] == H and c["W"] == W for c in cams), "the views of a batch share one image size"
        tb = ((W + TILE - 1) // TILE, (H + TILE - 1) // TILE, 1)
        T = tb[0] * tb[1]
        if tb[0] > 255 or tb[1] > 255:
            raise ValueError("render_views: at most 255 x 255 tiles (packed tile boxes)")
        K = features_rest.shape[1] + 1
        sh_degree = {1: 0, 4: 1, 9: 2, 16: 3}[K]
        m, ls, q = _c(means), _c(log_scales), _c(quats)
        op, dc, rest = _c(opacities).reshape(-1), _c(features_dc), _c(features_rest)
        CH = _cams_host(cams)
        f32 = dict(device=dev, dtype=torch.float32); i32 = dict(device=dev, dtype=torch.int32)
        xys = torch.empty(C, N, 2, **f32); depths = torch.empty(C, N, **f32); radii = torch.empty(C, N, **i32)
        conics = torch.empty(C, N, 3, **f32); nth = torch.empty(C, N, **i32); rgbs = torch.empty(C, N, 3, **f32)
        opac = torch.empty(N, **f32); boxes = torch.empty(C, N, **i32); pairs = torch.empty(C, N, 2, **i32)
        L.check(lib.gc_project_sh_fwd_views(
            L.i64(N), L.i32(C), L.ptr(m), L.ptr(ls), L.ptr(q), L.ptr(op), L.ptr(dc), L.ptr(rest), L.i32(sh_degree), L.i32(sh_degree_to_use),
            CH, L.i32(H), L.i32(W), L.i32(tb[0]), L.i32(tb[1]), L.f32(0.01), L.ptr(xys), L.ptr(depths), L.ptr(radii), L.ptr(conics),
            L.ptr(nth), L.ptr(rgbs), L.ptr(opac), L.ptr(boxes), L.ptr(pairs), st), "gc_project_sh_fwd_views")
        order = torch.empty(C, N, **i32); cum = torch.empty(C, N, **i32); cnt = torch.empty(C, **i32)
        sorted_boxes = bool(getattr(aux, "sorted_boxes", True)) if aux is not None else True
        if sorted_boxes:      # round 6: the boxes ride through the depth sort, culled Gaussians drop out in its first pass -- no per-Gaussian gathers
            bxs = torch.empty(C, N, **i32); nvis = torch.empty(C, **i32)
            wb = int(lib.gc_raster_order_boxes_views_workspace_bytes(L.i64(N), L.i32(C)))
            ws = torch.empty(wb, dtype=torch.uint8, device=dev)
            L.check(lib.gc_raster_order_boxes_views(L.i64(N), L.i32(C), L.ptr(pairs), L.ptr(boxes), L.ptr(order), L.ptr(bxs), L.ptr(cum), L.ptr(cnt),
                                                    L.ptr(nvis), L.ptr(ws), L.C.c_size_t(wb), st), "gc_raster_order_boxes_views")
        else:                 # the round-5 chain (A/B, cross-check tests)
            wb = int(lib.gc_raster_depth_order_views_workspace_bytes(L.i64(N), L.i32(C)))
            ws = torch.empty(wb, dtype=torch.uint8, device=dev)
            L.check(lib.gc_raster_depth_order_views(L.i64(N), L.i32(C), None, None, L.ptr(pairs), L.ptr(nth), L.ptr(order), L.ptr(cum), L.ptr(cnt),
                                                    L.ptr(ws), L.C.c_size_t(wb), st), "gc_raster_depth_order_views")
        del pairs
        m_cap = None if aux is None else aux.m_cap
        if m_cap is None:                     # one readback of the C counts sizes the lists exactly (the sync-free form passes a capacity)
            M_cap = max(int(cnt.max().item()), 1)
        else:
            M_cap = int(m_cap)
        ids_s = torch.empty(C, M_cap, **i32); bins = torch.empty(C, T, 2, **i32); ovf = torch.empty(C, **i32)
        bb = int(lib.gc_raster_bin_views_workspace_bytes(L.i64(M_cap), L.i32(C)))
        del ws
        bws = torch.empty(bb, dtype=torch.uint8, device=dev)
        if sorted_boxes:
            L.check(lib.gc_raster_bin_sorted_views(L.i64(N), L.i32(C), L.i64(M_cap), L.ptr(cnt), L.ptr(ovf), L.ptr(nvis), L.ptr(order), L.ptr(bxs),
                                                   L.ptr(cum), L.i32(tb[0]), L.i32(tb[1]), L.ptr(ids_s), L.ptr(bins), L.ptr(bws), L.C.c_size_t(bb), st),
                    "gc_raster_bin_sorted_views")
            del bxs
        else:
            L.check(lib.gc_raster_bin_tiles_views(L.i64(N), L.i32(C), L.i64(M_cap), L.ptr(cnt), L.ptr(ovf), L.ptr(order), L.ptr(cum), L.ptr(boxes),
                                                  L.ptr(depths), L.i32(tb[0]), L.i32(tb[1]), L.ptr(ids_s), L.ptr(bins), None, L.ptr(bws),
                                                  L.C.c_size_t(bb), st), "gc_raster_bin_tiles_views")
        del bws
        bg = _c(backgrounds)
        shared_bg = 1 if bg.dim() == 1 else 0
        assert bg.numel() == (3 if shared_bg else 3 * C)
        img = torch.empty(C, H, W, 3, **f32); fT = torch.empty(C, H, W, **f32); fi = torch.empty(C, H, W, **i32)
        dep = torch.empty(C, H, W, **f32) if want_depth else None
        L.check(lib.gc_rasterize_fwd_views(L.i32(C), L.i64(N), L.i64(M_cap), L.i32(1), L.i32(shared_bg), L.i32(H), L.i32(W), L.i32(tb[0]),
                                           L.i32(tb[1]), L.ptr(ids_s), L.ptr(bins), L.ptr(xys), L.ptr(conics), L.ptr(rgbs), L.ptr(opac),
                                           L.ptr(depths if want_depth else None), L.ptr(bg), L.ptr(img), L.ptr(dep), L.ptr(fT), L.ptr(fi), st),
                "gc_rasterize_fwd_views")
        alpha = torch.empty(C, H, W, **f32)
        if any(ctx.needs_input_grad[:6]):
            pre_clamp, img = img, torch.empty_like(img)
            L.check(lib.gc_raster_finalize_into(L.i64(C * H * W), L.ptr(pre_clamp), L.ptr(img), L.ptr(dep), L.ptr(fT), L.ptr(alpha), st),
                    "gc_raster_finalize_into")
        else:
            pre_clamp = None
            L.check(lib.gc_raster_finalize(L.i64(C * H * W), L.ptr(img), L.ptr(dep), L.ptr(fT), L.ptr(alpha), st), "gc_raster_finalize")
        if aux is not None:
            aux.xys, aux.radii, aux.num_tiles_hit, aux.depths, aux.tile_boxes = xys, radii, nth, depths, boxes
            aux.M = (cnt, ovf)                          # per-view device counts / overflow flags ([C] each)
            aux.gaussian_ids_sorted, aux.tile_bins, aux.final_index, aux.isect_ids_sorted = ids_s, bins, fi, None
            aux.xys_grad = None
        ctx.save_for_backward(m, ls, q, op, dc, rest, radii, conics, xys, rgbs, opac, ids_s, bins, bg, fT, fi, pre_clamp)
        ctx.meta = (cams, CH, tb, N, C, M_cap, shared_bg, sh_degree, int(sh_degree_to_use))
        ctx.aux = aux
        if dep is not None:
            ctx.mark_non_differentiable(dep)
        else:
            dep = torch.empty(0, device=dev)
        return img, alpha, dep

    @staticmethod
    def backward(ctx, v_img, v_alpha, v_dep):
        (m, ls, q, op, dc, rest, radii, conics, xys, rgbs, opac, ids_s, bins, bg, fT, fi, pre_clamp) = ctx.saved_tensors
        cams, CH, tb, N, C, M_cap, shared_bg, sh_degree, n_use = ctx.meta
        H, W = cams[0]["H"], cams[0]["W"]
        dev = m.device
        lib = L.lib()
        st = L.stream_ptr()
        vo = _c(v_img) if v_img is not None else torch.zeros(C, H, W, 3, device=dev)
        va = _c(v_alpha) if v_alpha is not None else None
        vbuf = torch.zeros(C * N * 9, device=dev)                      # v_xy | v_conic | v_colors | v_opacity, each [C][N][..]
        v_xy = vbuf[:2 * C * N].view(C, N, 2); v_conic = vbuf[2 * C * N:5 * C * N].view(C, N, 3)
        v_col = vbuf[5 * C * N:8 * C * N].view(C, N, 3); v_op = vbuf[8 * C * N:].view(C, N)
        L.check(lib.gc_rasterize_bwd_views(L.i32(C), L.i64(N), L.i64(M_cap), L.i32(1), L.i32(shared_bg), L.i32(H), L.i32(W), L.i32(tb[0]),
                                           L.i32(tb[1]), L.ptr(ids_s), L.ptr(bins), L.ptr(xys), L.ptr(conics), L.ptr(rgbs), L.ptr(opac), L.ptr(bg),
                                           L.ptr(fT), L.ptr(fi), L.ptr(vo), L.ptr(va), L.ptr(pre_clamp), L.ptr(v_xy), L.ptr(v_conic),
                                           L.ptr(v_col), L.ptr(v_op), st), "gc_rasterize_bwd_views")
        if ctx.aux is not None:
            ctx.aux.xys_grad = v_xy
        into = ctx.aux.grad_into if ctx.aux is not None else None
        acc = 0
        if into is not None:
            vm, vls, vq, vop, vdc, vrest = (into[k] for k in ("means", "scales", "quats", "opacities", "features_dc", "features_rest"))
            for t, ref in ((vm, m), (vls, ls), (vq, q), (vdc, dc), (vrest, rest)):
                assert t.is_contiguous() and t.dtype == torch.float32 and t.shape == ref.shape and t.device == dev
            assert vop.is_contiguous() and vop.dtype == torch.float32 and vop.numel() == N and vop.device == dev
            acc = 1 if ctx.aux.grad_accumulate else 0
        else:
            vm = torch.empty(N, 3, device=dev); vls = torch.empty(N, 3, device=dev); vq = torch.empty(N, 4, device=dev)
            vop = torch.empty(N, device=dev); vdc = torch.empty(N, 3, device=dev); vrest = torch.empty(rest.shape, device=dev)
        L.check(lib.gc_project_sh_bwd_views(
            L.i64(N), L.i32(C), L.i32(acc), L.ptr(m), L.ptr(ls), L.ptr(q), L.ptr(op), L.ptr(rgbs), L.i32(sh_degree), L.i32(n_use), CH,
            L.i32(H), L.i32(W), L.ptr(radii), L.ptr(conics), L.ptr(v_xy), L.ptr(v_conic), L.ptr(v_col), L.ptr(v_op), L.ptr(vm), L.ptr(vls),
            L.ptr(vq), L.ptr(vop), L.ptr(vdc), L.ptr(vrest), st), "gc_project_sh_bwd_views")
        if into is not None:
            return (None,) * 11
        return vm, vls, vq, vop[:, None], vdc, vrest, None, None, None, None, None


def render_views(means, log_scales, quats, opacities, features_dc, features_rest, cams: list, backgrounds, want_depth: bool,
                 sh_degree_to_use: int = 3, aux: RenderAux | None = None):
    """render_view for a LIST of cameras of one scene in one set of launches.  backgrounds: [3] (shared) or [C,3].
    Returns (rgb [C,H,W,3] clamped to <= 1, alpha [C,H,W], depth [C,H,W] or empty).  aux.M = (count [C], overflow [C]) device tensors;
    aux.m_cap = per-view intersection capacity for the sync-free form (else the C counts are read back once to size the lists)."""
    return _RenderViews.apply(means, log_scales, quats, opacities, features_dc, features_rest, list(cams), backgrounds, want_depth,
                              sh_degree_to_use, aux)
