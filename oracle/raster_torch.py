"""oracle/raster_torch.py -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

Independent, vectorised, *differentiable* PyTorch restatement (fp32 or fp64, CPU) of the
rasterizer semantics of SURVEY.md Appendix A, used to

  * pin the C oracle (oracle/raster_ref.c): forward values and -- through torch.autograd in
    fp64 -- every backward formula (the C backward is hand-derived);
  * restate the reference's glue arithmetic around the three gsplat calls:
      camera -> viewmat / projmat           /root/reference/gaussctrl/gc_model.py:97-121
      colours = cat(dc, rest); SH; +0.5 clamp  gc_model.py:138,162-169
      rgb clamp(max=1), depth / alpha, 1000 where alpha == 0   gc_model.py:188-204

gsplat 0.1.3 itself (project_gaussians / spherical_harmonics / rasterize_gaussians; call sites
gc_model.py:140-154,166,174-186,191-202) is third-party and absent from /root/reference:
PARITY UNPINNED at that boundary (see oracle/raster_ref.c header).
"""
from __future__ import annotations

import math

import torch

TILE = 16
ALPHA_CAP = 0.999
ALPHA_MIN = 1.0 / 255.0
T_STOP = 1e-4

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


# ------------------------------------------------------------------ camera glue (gc_model.py:97-121)
def projection_matrix(znear, zfar, fovx, fovy, dtype=torch.float32):
    """nerfstudio splatfacto.projection_matrix [recall, SURVEY.md 8a/A2]."""
    t = znear * math.tan(0.5 * fovy)
    b = -t
    r = znear * math.tan(0.5 * fovx)
    l = -r
    n, f = znear, zfar
    return torch.tensor(
        [[2 * n / (r - l), 0.0, (r + l) / (r - l), 0.0],
         [0.0, 2 * n / (t - b), (t + b) / (t - b), 0.0],
         [0.0, 0.0, (f + n) / (f - n), -1.0 * f * n / (f - n)],
         [0.0, 0.0, 1.0, 0.0]], dtype=dtype)


def camera_to_gsplat(c2w, fx, fy, W, H, dtype=torch.float32):
    """gc_model.py:97-115: c2w [3,4] (OpenGL, -z forward) -> (viewmat[4,4], projmat[4,4], fullproj)."""
    c2w = c2w.to(dtype)
    R = c2w[:3, :3]
    T = c2w[:3, 3:4]
    R = R @ torch.diag(torch.tensor([1.0, -1.0, -1.0], dtype=dtype))
    R_inv = R.T
    T_inv = -R_inv @ T
    viewmat = torch.eye(4, dtype=dtype)
    viewmat[:3, :3] = R_inv
    viewmat[:3, 3:4] = T_inv
    fovx = 2 * math.atan(W / (2 * fx))
    fovy = 2 * math.atan(H / (2 * fy))
    projmat = projection_matrix(0.001, 1000, fovx, fovy, dtype=dtype)
    return viewmat, projmat, projmat @ viewmat


# ------------------------------------------------------------------ A.1 projection
def quat_to_rotmat(q):
    q = q / q.norm(dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=-1)
    return R.reshape(q.shape[:-1] + (3, 3))


def project_gaussians(means3d, scales, glob_scale, quats, viewmat, projmat, fx, fy, cx, cy, H, W,
                      tile_bounds, clip_thresh=0.01):
    """Returns (xys, depths, radii, conics, num_tiles_hit, cov3d) like gsplat 0.1.3's 6-tuple."""
    dt = means3d.dtype
    N = means3d.shape[0]
    V = viewmat.to(dt)
    P = projmat.to(dt)
    W3 = V[:3, :3]
    t = means3d @ W3.T + V[:3, 3]
    tz = t[:, 2]
    keep = tz > clip_thresh
    R = quat_to_rotmat(quats)
    M = R * (glob_scale * scales)[:, None, :]
    Sigma = M @ M.transpose(1, 2)
    cov3d = torch.stack([Sigma[:, 0, 0], Sigma[:, 0, 1], Sigma[:, 0, 2], Sigma[:, 1, 1], Sigma[:, 1, 2], Sigma[:, 2, 2]], -1)
    lim_x = 1.3 * (0.5 * W / fx)
    lim_y = 1.3 * (0.5 * H / fy)
    tzs = torch.where(keep, tz, torch.ones_like(tz))
    txc = tzs * torch.clamp(t[:, 0] / tzs, -lim_x, lim_x)
    tyc = tzs * torch.clamp(t[:, 1] / tzs, -lim_y, lim_y)
    rz = 1.0 / tzs
    rz2 = rz * rz
    zero = torch.zeros_like(rz)
    J = torch.stack([fx * rz, zero, -fx * txc * rz2, zero, fy * rz, -fy * tyc * rz2], -1).reshape(N, 2, 3)
    Tm = J @ W3
    cov = Tm @ Sigma @ Tm.transpose(1, 2)
    a = cov[:, 0, 0] + 0.3
    b = cov[:, 0, 1]
    d = cov[:, 1, 1] + 0.3
    det = a * d - b * b
    keep = keep & (det != 0)
    dets = torch.where(det != 0, det, torch.ones_like(det))
    conics = torch.stack([d / dets, -b / dets, a / dets], -1)
    mid = 0.5 * (a + d)
    disc = torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3.0 * torch.sqrt(torch.maximum(mid + disc, mid - disc)))
    ph = torch.cat([means3d, torch.ones_like(means3d[:, :1])], -1) @ P.T
    rw = 1.0 / (ph[:, 3] + 1e-6)
    px = 0.5 * W * (ph[:, 0] * rw) + cx - 0.5
    py = 0.5 * H * (ph[:, 1] * rw) + cy - 0.5
    xys = torch.stack([px, py], -1)
    tx_b, ty_b = tile_bounds[0], tile_bounds[1]
    with torch.no_grad():
        tcx, tcy, tr = px / TILE, py / TILE, radius / TILE
        minx = torch.clamp((tcx - tr).to(torch.int64), 0, tx_b)      # trunc toward zero, like C
        maxx = torch.clamp((tcx + tr + 1).to(torch.int64), 0, tx_b)
        miny = torch.clamp((tcy - tr).to(torch.int64), 0, ty_b)
        maxy = torch.clamp((tcy + tr + 1).to(torch.int64), 0, ty_b)
        area = (maxx - minx) * (maxy - miny)
        keep = keep & (area > 0)
        num_tiles_hit = torch.where(keep, area, torch.zeros_like(area)).to(torch.int32)
        radii = torch.where(keep, radius.to(torch.int32), torch.zeros_like(radius, dtype=torch.int32))
    kf = keep.to(dt)
    xys = xys * kf[:, None]
    depths = tz * kf
    conics = conics * kf[:, None]
    return xys, depths, radii, conics, num_tiles_hit, cov3d


# ------------------------------------------------------------------ A.2 SH
def num_sh_bases(degree: int) -> int:
    return (degree + 1) ** 2


def sh_basis(n, dirs):
    x, y, z = dirs.unbind(-1)
    B = [torch.full_like(x, SH_C0)]
    if n >= 1:
        B += [-SH_C1 * y, SH_C1 * z, -SH_C1 * x]
    if n >= 2:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        B += [SH_C2[0] * xy, SH_C2[1] * yz, SH_C2[2] * (2 * zz - xx - yy), SH_C2[3] * xz, SH_C2[4] * (xx - yy)]
    if n >= 3:
        B += [SH_C3[0] * y * (3 * xx - yy), SH_C3[1] * xy * z, SH_C3[2] * y * (4 * zz - xx - yy),
              SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy), SH_C3[4] * x * (4 * zz - xx - yy),
              SH_C3[5] * z * (xx - yy), SH_C3[6] * x * (xx - 3 * yy)]
    return torch.stack(B, -1)


def spherical_harmonics(degrees_to_use, viewdirs, coeffs):
    B = sh_basis(degrees_to_use, viewdirs)                     # [N, Ku]
    Ku = B.shape[-1]
    return (B[:, :, None] * coeffs[:, :Ku, :]).sum(1)


# ------------------------------------------------------------------ A.3 bin & sort
def bin_and_sort(xys, depths, radii, num_tiles_hit, tile_bounds):
    """Returns (isect_ids_sorted int64[M], gaussian_ids_sorted int32[M], tile_bins int32[T,2])."""
    tx_b, ty_b = tile_bounds[0], tile_bounds[1]
    keys, ids = [], []
    xs = xys.detach().to(torch.float32)
    dbits = depths.detach().to(torch.float32).view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    for i in torch.nonzero(radii > 0).flatten().tolist():
        tcx = xs[i, 0] / TILE
        tcy = xs[i, 1] / TILE
        tr = radii[i].to(torch.float32) / TILE
        minx = min(max(int(tcx - tr), 0), tx_b)
        maxx = min(max(int(tcx + tr + 1), 0), tx_b)
        miny = min(max(int(tcy - tr), 0), ty_b)
        maxy = min(max(int(tcy + tr + 1), 0), ty_b)
        for ty in range(miny, maxy):
            for tx in range(minx, maxx):
                keys.append(((ty * tx_b + tx) << 32) | int(dbits[i]))
                ids.append(i)
    keys = torch.tensor(keys, dtype=torch.int64)
    ids = torch.tensor(ids, dtype=torch.int32)
    order = torch.sort(keys, stable=True).indices
    keys, ids = keys[order], ids[order]
    T = tx_b * ty_b
    bins = torch.zeros(T, 2, dtype=torch.int32)
    if keys.numel():
        tiles = (keys >> 32).to(torch.int64)
        for t in torch.unique(tiles).tolist():
            idx = torch.nonzero(tiles == t).flatten()
            bins[t, 0] = int(idx[0])
            bins[t, 1] = int(idx[-1]) + 1
    return keys, ids, bins


# ------------------------------------------------------------------ A.4 rasterize (differentiable)
def composite_tile(xys, conics, colors, opacities, gid, s, h0, h1, w0, w1, extra=None):
    """Front-to-back compositing of the splats `gid` (already in list order; `s` = list position of gid[0]) over the pixel block
    [h0,h1) x [w0,w1).  Returns (img[P,C] WITHOUT background, T_final[P], last contributing list index[P], extra[P] or None)."""
    dt = xys.dtype
    py, px = torch.meshgrid(torch.arange(h0, h1, dtype=dt), torch.arange(w0, w1, dtype=dt), indexing="ij")
    px, py = px.reshape(-1, 1), py.reshape(-1, 1)                 # [P,1]
    dx = xys[gid, 0][None, :] - px
    dy = xys[gid, 1][None, :] - py
    cn = conics[gid]
    sigma = 0.5 * (cn[:, 0] * dx * dx + cn[:, 2] * dy * dy) + cn[:, 1] * dx * dy
    alpha = torch.clamp(opacities[gid][None, :] * torch.exp(-sigma), max=ALPHA_CAP)
    valid = (sigma >= 0) & (alpha >= ALPHA_MIN)
    a_eff = torch.where(valid, alpha, torch.zeros_like(alpha))
    one_m = 1 - a_eff
    T_incl = torch.cumprod(one_m, dim=1)                          # T after k
    T_excl = torch.cat([torch.ones_like(T_incl[:, :1]), T_incl[:, :-1]], 1)
    stop = valid & (T_incl <= T_STOP)                             # first such k terminates the pixel
    alive = torch.cumsum(stop.to(torch.int32), 1) == 0            # k strictly before the stop
    contrib = valid & alive
    wgt = torch.where(contrib, a_eff * T_excl, torch.zeros_like(a_eff))
    img = wgt @ colors[gid]
    Tfin = torch.prod(torch.where(contrib, one_m, torch.ones_like(one_m)), 1)
    kk = torch.arange(s, s + gid.numel(), dtype=torch.int32)[None, :].expand_as(contrib)
    last = torch.where(contrib, kk, torch.zeros_like(kk)).max(1).values
    return img, Tfin, last, (wgt @ extra[gid] if extra is not None else None)


def rasterize(xys, conics, colors, opacities, gaussian_ids_sorted, tile_bins, H, W, tile_bounds, background,
              extra=None):
    """Vectorised per tile.  colors [N,C]; returns (out_img[H,W,C], out_alpha[H,W], final_index[H,W],
    out_extra[H,W] or None)."""
    dt = xys.dtype
    C = colors.shape[1]
    out = torch.zeros(H, W, C, dtype=dt)
    out_T = torch.ones(H, W, dtype=dt)
    out_e = torch.zeros(H, W, dtype=dt) if extra is not None else None
    fidx = torch.zeros(H, W, dtype=torch.int32)
    tx_b, ty_b = tile_bounds[0], tile_bounds[1]
    rows = []
    for ty in range(ty_b):
        cols_img, cols_T, cols_e, cols_i = [], [], [], []
        for tx in range(tx_b):
            s, e = int(tile_bins[ty * tx_b + tx, 0]), int(tile_bins[ty * tx_b + tx, 1])
            h0, w0 = ty * TILE, tx * TILE
            h1, w1 = min(h0 + TILE, H), min(w0 + TILE, W)
            ph, pw = h1 - h0, w1 - w0
            if e <= s:
                cols_img.append(torch.zeros(ph, pw, C, dtype=dt)); cols_T.append(torch.ones(ph, pw, dtype=dt))
                cols_e.append(torch.zeros(ph, pw, dtype=dt)); cols_i.append(torch.zeros(ph, pw, dtype=torch.int32))
                continue
            gid = gaussian_ids_sorted[s:e].to(torch.int64)
            img, Tfin, last, ext = composite_tile(xys, conics, colors, opacities, gid, s, h0, h1, w0, w1, extra)
            cols_img.append(img.reshape(ph, pw, C)); cols_T.append(Tfin.reshape(ph, pw)); cols_i.append(last.reshape(ph, pw))
            cols_e.append(ext.reshape(ph, pw) if extra is not None else torch.zeros(ph, pw, dtype=dt))
        rows.append((torch.cat(cols_img, 1), torch.cat(cols_T, 1), torch.cat(cols_e, 1), torch.cat(cols_i, 1)))
    out = torch.cat([r[0] for r in rows], 0)
    out_T = torch.cat([r[1] for r in rows], 0)
    out_e = torch.cat([r[2] for r in rows], 0) if extra is not None else None
    fidx = torch.cat([r[3] for r in rows], 0)
    out = out + out_T[..., None] * background.to(dt)
    return out, 1 - out_T, fidx, out_e


# ------------------------------------------------------------------ gc_model.get_outputs restated
def get_outputs(params, c2w, fx, fy, cx, cy, W, H, background, training, sh_degree_to_use=3, dtype=torch.float32):
    """GaussCtrlModel.get_outputs (gc_model.py:57-206) on the six splatfacto parameter tensors.
    params: dict(means, scales(log), quats, opacities(logit)[N,1], features_dc[N,3], features_rest[N,15,3])."""
    p = {k: v.to(dtype) for k, v in params.items()}
    viewmat, projmat, full = camera_to_gsplat(c2w, fx, fy, W, H, dtype)
    tile_bounds = ((W + TILE - 1) // TILE, (H + TILE - 1) // TILE, 1)
    colors = torch.cat([p["features_dc"][:, None, :], p["features_rest"]], 1)
    quats = p["quats"] / p["quats"].norm(dim=-1, keepdim=True)
    xys, depths, radii, conics, nth, _ = project_gaussians(
        p["means"], torch.exp(p["scales"]), 1.0, quats, viewmat[:3, :], full, fx, fy, cx, cy, H, W, tile_bounds)
    if int(radii.sum()) == 0:
        return {"rgb": background.to(dtype).repeat(H, W, 1)}
    viewdirs = p["means"].detach() - c2w[:3, 3].to(dtype)
    viewdirs = viewdirs / viewdirs.norm(dim=-1, keepdim=True)
    rgbs = torch.clamp(spherical_harmonics(sh_degree_to_use, viewdirs, colors) + 0.5, min=0.0)
    opac = torch.sigmoid(p["opacities"])[:, 0]
    _, ids, bins = bin_and_sort(xys, depths, radii, nth, tile_bounds)
    extra = None if training else depths
    rgb, alpha, fidx, dep = rasterize(xys, conics, rgbs, opac, ids, bins, H, W, tile_bounds, background, extra=extra)
    rgb = torch.clamp(rgb, max=1.0)
    alpha = alpha[..., None]
    depth_im = None
    if not training:
        depth_im = dep[..., None].clone()
        pos = alpha > 0
        depth_im = torch.where(pos, depth_im / torch.where(pos, alpha, torch.ones_like(alpha)), torch.full_like(depth_im, 1000.0))
    return {"rgb": rgb, "depth": depth_im, "accumulation": alpha, "xys": xys, "radii": radii,
            "gaussian_ids_sorted": ids, "tile_bins": bins, "final_index": fidx}


# ------------------------------------------------------------------ fp64 gradients at scene sizes the vectorised form cannot hold
def render_grads_tiled(params, c2w, fx, fy, cx, cy, W, H, background, v_rgb, gaussian_ids_sorted, tile_bins, sh_degree_to_use=3,
                       alpha_weight=1.0, dtype=torch.float64, device="cpu"):
    """d/d(params) of  sum(rgb * v_rgb) + alpha_weight * sum(alpha)  for the TRAINING render of get_outputs above (rgb clamped to <= 1,
    gc_model.py:188), evaluated tile by tile so that a 60 k-Gaussian stress scene with multi-million-entry lists fits: the per-tile
    compositing graph (same expressions as `rasterize`) is differentiated on its own down to (xys, conics, rgbs, opacities) and the
    sums are pushed through projection / SH / sigmoid once.  `gaussian_ids_sorted` / `tile_bins`: the gsplat-box lists (bit-exact
    between the C oracle and the HIP binning, tests/test_raster_gpu.py) -- the per-pixel tests do not depend on the lists' length.
    `device` may be a GPU: this is the checker running in float64 there, not the product."""
    with torch.device(device):
        p = {k: torch.as_tensor(v).to(device=device, dtype=dtype).requires_grad_(True) for k, v in params.items()}
        c2w_t = torch.as_tensor(c2w).to(device=device, dtype=dtype)[:3]
        viewmat, projmat, full = camera_to_gsplat(c2w_t, fx, fy, W, H, dtype)
        tb = ((W + TILE - 1) // TILE, (H + TILE - 1) // TILE, 1)
        colors = torch.cat([p["features_dc"][:, None, :], p["features_rest"]], 1)
        quats = p["quats"] / p["quats"].norm(dim=-1, keepdim=True)
        xys, depths, radii, conics, nth, _ = project_gaussians(p["means"], torch.exp(p["scales"]), 1.0, quats, viewmat[:3], full,
                                                               fx, fy, cx, cy, H, W, tb)
        viewdirs = p["means"].detach() - c2w_t[:3, 3]
        viewdirs = viewdirs / viewdirs.norm(dim=-1, keepdim=True)
        rgbs = torch.clamp(spherical_harmonics(sh_degree_to_use, viewdirs, colors) + 0.5, min=0.0)
        opac = torch.sigmoid(p["opacities"])[:, 0]
        mid = [t.detach().requires_grad_(True) for t in (xys, conics, rgbs, opac)]
        acc = [torch.zeros_like(t) for t in mid]
        bg = torch.as_tensor(background).to(device=device, dtype=dtype)
        v = torch.as_tensor(v_rgb).to(device=device, dtype=dtype)
        ids = torch.as_tensor(gaussian_ids_sorted).to(device=device, dtype=torch.int64)
        bins_h = torch.as_tensor(tile_bins).cpu().numpy()
        for ty in range(tb[1]):
            for tx in range(tb[0]):
                s, e = int(bins_h[ty * tb[0] + tx, 0]), int(bins_h[ty * tb[0] + tx, 1])
                if e <= s:
                    continue
                h0, w0 = ty * TILE, tx * TILE
                h1, w1 = min(h0 + TILE, H), min(w0 + TILE, W)
                img, Tfin, _, _ = composite_tile(mid[0], mid[1], mid[2], mid[3], ids[s:e], s, h0, h1, w0, w1)
                rgb = torch.clamp(img + Tfin[:, None] * bg, max=1.0).reshape(h1 - h0, w1 - w0, 3)
                loss = (rgb * v[h0:h1, w0:w1]).sum() + alpha_weight * (1 - Tfin).sum()
                for a, g in zip(acc, torch.autograd.grad(loss, mid)):
                    a += g
        torch.autograd.backward([xys, conics, rgbs, opac], acc)
    return {k: t.grad.detach().cpu() for k, t in p.items()}

