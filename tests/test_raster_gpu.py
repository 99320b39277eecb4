"""GPU parity tests of the rasterizer: HIP kernels (through the C ABI) vs the CPU oracle on the same
seeded inputs.  Bars (BASELINE.json north_star): tile indices / sort keys bit-exact; rasterized
RGB / depth within 1e-4 relative; gradients within 1e-3 of the per-tensor max (float atomics)."""
import math

import numpy as np
import pytest
import torch

from _margins import within
from gaussctrl_amd import synthetic as syn

pytestmark = pytest.mark.gpu
BG = np.array([0.1, 0.2, 0.3], np.float32)
DEV = "cuda:0"


def _t(a, dtype=torch.float32):
    return torch.tensor(np.asarray(a), dtype=dtype, device=DEV)


def _scene(N, W, H, fx, seed=3, scale_mean=0.03):
    P = syn.make_gaussians(N, seed=seed, scale_mean=scale_mean)
    c2w = syn.make_cameras(1, seed=seed + 1)[0]
    return P, c2w, dict(fx=fx, fy=fx * 0.99, cx=W / 2 + 1.3, cy=H / 2 - 2.1, W=W, H=H)


def _relerr(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-12)


def _img_close(got, ref, rel=1e-4, frac_max=1e-5):
    """|diff| <= rel * max|ref| on every pixel, except for knife-edge pixels where a 1-ulp difference in
    exp() flips one of the discrete decisions of SURVEY A.4 (alpha >= 1/255, T' <= 1e-4): at most 1e-5
    of all values may exceed the bound, and then by no more than one dropped splat (1/255 * max colour)."""
    got = np.asarray(got, np.float64); ref = np.asarray(ref, np.float64)
    d = np.abs(got - ref); bound = rel * (np.abs(ref).max() + 1e-12)
    frac = float((d > bound).mean())
    within(f"img: fraction of values beyond {rel} rel", frac, frac_max)
    within("img: worst value (one dropped splat)", d.max(), (1.0 / 255.0) * max(1.0, np.abs(ref).max()))


def _grad_close(got, ref, scale):
    """float-atomic accumulation order differs from the oracle: |diff| <= 1e-3 * max|ref| + 1e-6 * scale on every Gaussian (row)
    (scale = largest gradient magnitude of the whole parameter set; guards exactly-zero gradients) -- except knife-edge Gaussians, the
    gradient-side twin of _img_close's knife-edge pixels: where a splat's alpha at ONE pixel sits within an ulp of exp() of the discrete
    1/255 test (measured round 6, scripts/diag_views_grad.py: 1 M bear scene, camera 33, Gaussian 122961 at pixel (360, 7): alpha =
    0.00392156607 vs 1/255 = 0.00392156863, 6.5e-7 relative), product and oracle decide differently and that pixel's whole contribution
    enters one Gaussian's six gradients on one side only (every other Gaussian of the 1 M agreed to 6e-6).  At most 1e-5 of the rows (and
    never more than a handful; none in the scenes below 500 k Gaussians) may exceed the bar, each by no more than 2e-2 of the tensor's max
    (one (pixel, splat) pair)."""
    got = np.asarray(got, np.float64); ref = np.asarray(ref, np.float64)
    bar = 1e-3 * np.abs(ref).max() + 1e-6 * scale
    d = np.abs(got - ref).reshape(ref.shape[0], -1).max(axis=1) if ref.ndim > 1 else np.abs(got - ref)
    bad = d > bar
    nbad = int(bad.sum())
    allowed = max(2.0, 1e-5 * d.shape[0]) if d.shape[0] >= 500_000 else 0.0       # (only the full-size scenes get the allowance)
    within("grad: rows beyond 1e-3 of max (knife-edge Gaussians)", nbad, allowed)
    within("grad max-norm, rows within the bar", d[~bad].max() if nbad < d.shape[0] else 0.0, bar)
    if nbad:
        within("grad: worst knife-edge row", d.max(), 2e-2 * np.abs(ref).max() + 1e-6 * scale)


@pytest.mark.parametrize("N,W,H,fx", [(5000, 200, 136, 180.0), (200000, 512, 512, 540.0), (1, 64, 64, 100.0)])
def test_project_gaussians_bit_exact(oracle_c, N, W, H, fx):
    from gaussctrl_amd import gsplat_ops as ops
    from gaussctrl_amd.camera import camera_to_gsplat
    P, c2w, K = _scene(N, W, H, fx)
    cam = camera_to_gsplat(c2w, K["fx"], K["fy"], K["cx"], K["cy"], W, H)
    scales = np.exp(P["scales"]).astype(np.float32)
    qn = (P["quats"] / np.linalg.norm(P["quats"], axis=-1, keepdims=True)).astype(np.float32)
    tb = cam["tile_bounds"]
    V4 = cam["viewmat4"]; full = np.asarray(cam["fullproj"], np.float32).reshape(4, 4)
    ref = oracle_c.project_gaussians(P["means"], scales, 1.0, qn, V4[:3], full, K["fx"], K["fy"], K["cx"], K["cy"], H, W, tb)
    got = ops.project_gaussians(_t(P["means"]), _t(scales), 1.0, _t(qn), _t(V4[:3]), _t(full), K["fx"], K["fy"],
                                K["cx"], K["cy"], H, W, tb)
    names = ["xys", "depths", "radii", "conics", "num_tiles_hit", "cov3d"]
    for n, r, g in zip(names, ref, got):
        g = g.cpu().numpy()
        assert np.array_equal(r.view(np.int32) if r.dtype == np.float32 else r,
                              g.view(np.int32) if g.dtype == np.float32 else g), f"{n} not bit-exact"


def test_sh_fwd_bwd(oracle_c):
    from gaussctrl_amd import gsplat_ops as ops
    g = np.random.default_rng(0)
    N = 10007
    d = g.normal(size=(N, 3)); d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
    for K, n in [(16, 3), (16, 2), (16, 0), (9, 2), (4, 1), (1, 0)]:
        c = g.normal(size=(N, K, 3)).astype(np.float32)
        ref = oracle_c.spherical_harmonics(n, d, c)
        ct = _t(c).requires_grad_(True)
        out = ops.spherical_harmonics(n, _t(d), ct)
        within("_relerr(out.detach().cpu().numpy(), ref)", _relerr(out.detach().cpu().numpy(), ref), 1e-6, strict=True)
        v = g.normal(size=(N, 3)).astype(np.float32)
        out.backward(_t(v))
        refb = oracle_c.spherical_harmonics_bwd(n, d, K, v)
        within("_relerr(ct.grad.cpu().numpy(), refb)", _relerr(ct.grad.cpu().numpy(), refb), 1e-6, strict=True)


# tile counts chosen to take every branch of the tile passes (raster_sort.hip): 28 tiles -> one 5-bit staged pass, 63 -> one 6-bit,
# 117 / 1024 -> 5 + 5, 3185 -> 6 + 6, 7500 -> 8 + 8 (direct scatter)
@pytest.mark.parametrize("N,W,H,fx", [(5000, 200, 136, 180.0), (300000, 512, 512, 540.0), (3000, 100, 60, 90.0), (4000, 130, 100, 110.0),
                                      (60000, 1040, 784, 900.0), (60000, 1600, 1200, 1400.0)])
def test_bin_and_sort_bit_exact(oracle_c, N, W, H, fx):
    from gaussctrl_amd import gsplat_ops as ops
    P, c2w, K = _scene(N, W, H, fx)
    o = oracle_c.render(P, c2w, K["fx"], K["fy"], K["cx"], K["cy"], W, H, BG, training=True)
    tb = ((W + 15) // 16, (H + 15) // 16, 1)
    args = (N, _t(o["xys"]), _t(o["depths"]), _t(o["radii"], torch.int32), _t(o["num_tiles_hit"], torch.int32), tb)
    # gsplat-shaped chain (64-bit key sort)
    M, keys, ids, bins, cum = ops.bin_and_sort_gaussians_keys64(*args)
    assert M == o["M"]
    assert np.array_equal(cum.cpu().numpy(), np.cumsum(o["num_tiles_hit"]).astype(np.int32))
    assert np.array_equal(keys.cpu().numpy(), o["isect_ids_sorted"])
    assert np.array_equal(ids.cpu().numpy(), o["gaussian_ids_sorted"])
    assert np.array_equal(bins.cpu().numpy(), o["tile_bins"])
    # product path: two-level binning (depth order, then stable tile passes) must give the identical order
    M2, keys2, ids2, bins2, cum2 = ops.bin_and_sort_gaussians(*args, want_keys=True)
    assert M2 == o["M"]
    assert np.array_equal(keys2.cpu().numpy(), o["isect_ids_sorted"])
    assert np.array_equal(ids2.cpu().numpy(), o["gaussian_ids_sorted"])
    assert np.array_equal(bins2.cpu().numpy(), o["tile_bins"])
    assert int(cum2[-1]) == o["M"]
    M3, keys3, ids3, bins3, _ = ops.bin_and_sort_gaussians(*args)
    assert keys3 is None and torch.equal(ids3, ids2) and torch.equal(bins3, bins2)
    # sync-free variant: capacity-sized buffers, the count and the overflow flag stay on the device
    cap = int(o["M"] * 1.3) + 7
    (cnt, ovf), keys4, ids4, bins4, _ = ops.bin_and_sort_gaussians(*args, want_keys=True, m_cap=cap)
    assert int(cnt) == o["M"] and int(ovf) == 0
    assert torch.equal(ids4[:M], ids2) and torch.equal(keys4[:M], keys2) and torch.equal(bins4, bins2)
    if M > 64:                       # too small a capacity is reported, not silently truncated
        (cnt5, ovf5), _, _, _, _ = ops.bin_and_sort_gaussians(*args, m_cap=M // 2)
        assert int(cnt5) == o["M"] and int(ovf5) == 1


def test_block_culling_edge_cases(oracle_c):
    """The compositing kernels skip (8x8 block, Gaussian) pairs whose exact minimum of sigma over the block exceeds ln(255 opacity)
    (raster_composite.hip::block_mask).  Crafted cases where the bound is tight or degenerate, against the unculled oracle: needle-thin
    rotated Gaussians crossing block corners, opacities on both sides of 1/255 and near 1, Gaussians centred exactly on block / tile
    borders, giants covering the whole image, and points smaller than a pixel."""
    from gaussctrl_amd import gsplat_ops as ops
    from gaussctrl_amd.camera import camera_to_gsplat
    W, H, fx = 160, 128, 150.0
    g = np.random.default_rng(11)
    N = 4000
    P, c2w, K = _scene(N, W, H, fx, seed=5, scale_mean=0.02)
    P = {k: v.copy() for k, v in P.items()}
    s = P["scales"]                                           # log-scales
    s[:800] = np.log(np.stack([g.uniform(0.15, 0.6, 800), g.uniform(0.004, 0.01, 800), g.uniform(0.004, 0.01, 800)], 1)).astype(np.float32)  # needles (aspect <= 150: beyond, the fp32 scale gradient itself cancels catastrophically in either implementation)
    s[800:1000] = np.log(g.uniform(0.8, 2.0, (200, 3))).astype(np.float32)       # giants
    s[1000:1400] = np.log(g.uniform(1e-4, 6e-4, (400, 3))).astype(np.float32)    # sub-pixel points
    op = P["opacities"]                                        # logits
    lg = lambda p: np.log(p / (1 - p))
    op[1400:1800, 0] = lg(g.uniform(1 / 255 * 0.9, 1 / 255 * 1.3, 400)).astype(np.float32)      # around the alpha threshold
    op[1800:2000, 0] = lg(np.full(200, 0.9995)).astype(np.float32)                                 # above the 0.999 cap
    v_rgb = g.normal(size=(H, W, 3)).astype(np.float32); v_a = g.normal(size=(H, W)).astype(np.float32)
    o = oracle_c.render(P, c2w, K["fx"], K["fy"], K["cx"], K["cy"], W, H, BG, training=True, v_rgb=v_rgb, v_alpha=v_a)
    cam = camera_to_gsplat(c2w, K["fx"], K["fy"], K["cx"], K["cy"], W, H)
    tp = {k: _t(v).requires_grad_(True) for k, v in P.items()}
    colors = torch.cat([tp["features_dc"][:, None, :], tp["features_rest"]], 1)
    q = tp["quats"] / tp["quats"].norm(dim=-1, keepdim=True)
    V4 = _t(cam["viewmat4"]); full = _t(np.asarray(cam["fullproj"], np.float32).reshape(4, 4))
    xys, depths, radii, conics, nth, _ = ops.project_gaussians(tp["means"], torch.exp(tp["scales"]), 1, q, V4[:3], full,
                                                               K["fx"], K["fy"], K["cx"], K["cy"], H, W, cam["tile_bounds"])
    vd = tp["means"].detach() - _t(c2w[:3, 3]); vd = vd / vd.norm(dim=-1, keepdim=True)
    rgbs = torch.clamp(ops.spherical_harmonics(3, vd, colors) + 0.5, min=0.0)
    rgb, alpha = ops.rasterize_gaussians(xys, depths, radii, conics, nth, rgbs, torch.sigmoid(tp["opacities"]), H, W,
                                         background=_t(BG), return_alpha=True)
    rgbc = torch.clamp(rgb, max=1.0)
    _img_close(rgbc.detach().cpu().numpy(), o["rgb"], frac_max=2e-4)
    _img_close(alpha.detach().cpu().numpy(), o["accumulation"][..., 0], frac_max=2e-4)
    ((rgbc * _t(v_rgb)).sum() + (alpha * _t(v_a)).sum()).backward()
    # The crafted Gaussians put many pixels ON the discrete alpha >= 1/255 decision (threshold opacities; the far ends of needles, whose
    # long-axis gradient is dominated by exactly those pixels), where a 1-ulp difference of exp() flips a contribution in or out.  Their
    # gradients are therefore compared in aggregate (relative L2 over each tensor); the 2000 ordinary Gaussians of the same scene with the
    # max-norm bound, loosened to 1e-2 because a flipped splat in front changes T by 1/255 = 0.4 % for everything behind it at that pixel.
    # (That culling itself changes nothing is measured separately: scripts/culling_ab.py against a -DGC_NO_BLOCK_CULL build gives
    # bit-identical images and gradients equal to within the run-to-run noise of the float atomics, profiles/r02_block_culling_ab.txt.)
    plain = np.zeros(N, bool); plain[2000:] = True
    scale = max(np.abs(o["grads"][k][plain]).max() for k in P)
    for k in P:
        got, ref = tp[k].grad.cpu().numpy().astype(np.float64), o["grads"][k].astype(np.float64)
        assert np.isfinite(got).all()
        d = np.abs(got[plain] - ref[plain])
        within("d.max()", d.max(), 1e-2 * np.abs(ref[plain]).max() + 1e-6 * scale)
        within("np.linalg.norm(got - ref)", np.linalg.norm(got - ref), 5e-2 * np.linalg.norm(ref) + 1e-6 * scale)


@pytest.mark.parametrize("N,W,H,fx,sm", [(5000, 200, 136, 180.0, 0.03), (100000, 512, 512, 540.0, 0.01), (3, 40, 24, 50.0, 0.2)])
def test_rasterize_ops_fwd_bwd(oracle_c, N, W, H, fx, sm):
    """gsplat-surface chain project -> SH -> rasterize with autograd, vs the C oracle."""
    from gaussctrl_amd import gsplat_ops as ops
    from gaussctrl_amd.camera import camera_to_gsplat
    P, c2w, K = _scene(N, W, H, fx, scale_mean=sm)
    g = np.random.default_rng(1)
    v_rgb = g.normal(size=(H, W, 3)).astype(np.float32); v_a = g.normal(size=(H, W)).astype(np.float32)
    o = oracle_c.render(P, c2w, K["fx"], K["fy"], K["cx"], K["cy"], W, H, BG, training=True, v_rgb=v_rgb, v_alpha=v_a)
    cam = camera_to_gsplat(c2w, K["fx"], K["fy"], K["cx"], K["cy"], W, H)
    tp = {k: _t(v).requires_grad_(True) for k, v in P.items()}
    colors = torch.cat([tp["features_dc"][:, None, :], tp["features_rest"]], 1)
    q = tp["quats"] / tp["quats"].norm(dim=-1, keepdim=True)
    V4 = _t(cam["viewmat4"]); full = _t(np.asarray(cam["fullproj"], np.float32).reshape(4, 4))
    xys, depths, radii, conics, nth, _ = ops.project_gaussians(tp["means"], torch.exp(tp["scales"]), 1, q, V4[:3], full,
                                                               K["fx"], K["fy"], K["cx"], K["cy"], H, W, cam["tile_bounds"])
    xys.retain_grad()
    vd = tp["means"].detach() - _t(c2w[:3, 3]); vd = vd / vd.norm(dim=-1, keepdim=True)
    rgbs = torch.clamp(ops.spherical_harmonics(3, vd, colors) + 0.5, min=0.0)
    rgb, alpha = ops.rasterize_gaussians(xys, depths, radii, conics, nth, rgbs, torch.sigmoid(tp["opacities"]), H, W,
                                         background=_t(BG), return_alpha=True)
    rgbc = torch.clamp(rgb, max=1.0)
    _img_close(rgbc.detach().cpu().numpy(), o["rgb"])
    _img_close(alpha.detach().cpu().numpy(), o["accumulation"][..., 0])
    ((rgbc * _t(v_rgb)).sum() + (alpha * _t(v_a)).sum()).backward()
    scale = max(np.abs(o["grads"][k]).max() for k in P)
    for k in P:
        _grad_close(tp[k].grad.cpu().numpy(), o["grads"][k], scale)
    _grad_close(xys.grad.cpu().numpy(), o["grads"]["xys"], scale)


@pytest.mark.parametrize("N,W,H,fx,sm,training", [(5000, 200, 136, 180.0, 0.03, False), (200000, 512, 512, 540.0, 0.01, True),
                                                  (200000, 512, 512, 540.0, 0.01, False), (7, 33, 17, 40.0, 0.3, False)])
def test_fused_render_view(oracle_c, N, W, H, fx, sm, training):
    """Product path (one fused pass + one compositing sweep) vs get_outputs restated on the oracle."""
    from gaussctrl_amd import gsplat_ops as ops
    from gaussctrl_amd.camera import camera_to_gsplat
    P, c2w, K = _scene(N, W, H, fx, scale_mean=sm)
    g = np.random.default_rng(2)
    v_rgb = g.normal(size=(H, W, 3)).astype(np.float32); v_a = g.normal(size=(H, W)).astype(np.float32)
    o = oracle_c.render(P, c2w, K["fx"], K["fy"], K["cx"], K["cy"], W, H, BG, training=training, v_rgb=v_rgb, v_alpha=v_a)
    cam = camera_to_gsplat(c2w, K["fx"], K["fy"], K["cx"], K["cy"], W, H)
    tp = {k: _t(v).requires_grad_(True) for k, v in P.items()}
    aux = ops.RenderAux()
    rgb, alpha, depth = ops.render_view(tp["means"], tp["scales"], tp["quats"], tp["opacities"], tp["features_dc"],
                                        tp["features_rest"], cam, _t(BG), not training, 3, aux)
    _img_close(rgb.detach().cpu().numpy(), o["rgb"])
    _img_close(alpha.detach().cpu().numpy(), o["accumulation"][..., 0])
    # integer state: identical up to exp()/sqrt ulp differences in the fused front end
    mism = (aux.radii.cpu().numpy() != o["radii"]).mean()
    assert mism < 1e-4, mism
    # the product path bins on TIGHT tile boxes (RenderAux.tight_boxes, default on): fewer (tile, Gaussian) pairs than gsplat's lists ...
    assert aux.tile_boxes is not None and aux.M <= o["M"] + 4
    # ... and with gsplat's boxes the SAME image bit for bit, on lists of gsplat's length
    auxg = ops.RenderAux(); auxg.tight_boxes = False
    with torch.no_grad():
        rgb_g, alpha_g, depth_g = ops.render_view(tp["means"], tp["scales"], tp["quats"], tp["opacities"], tp["features_dc"],
                                                  tp["features_rest"], cam, _t(BG), not training, 3, auxg)
    assert abs(auxg.M - o["M"]) <= max(4, 1e-4 * o["M"])
    assert torch.equal(rgb_g, rgb.detach()) and torch.equal(alpha_g, alpha.detach()) and (training or torch.equal(depth_g, depth))
    if not training:
        d = depth.cpu().numpy(); od = o["depth"][..., 0]
        far = (od == 1000.0)
        assert np.array_equal(far, d == 1000.0)
        _img_close(np.where(far, 0, d), np.where(far, 0, od))
    ((rgb * _t(v_rgb)).sum() + (alpha * _t(v_a)).sum()).backward()
    scale = max(np.abs(o["grads"][k]).max() for k in P)
    for k in P:
        _grad_close(tp[k].grad.cpu().numpy(), o["grads"][k], scale)
    _grad_close(aux.xys_grad.cpu().numpy(), o["grads"]["xys"], scale)
    # sync-free frame (device-side intersection count, capacity-sized buffers): same image, no overflow; grads within atomics noise
    tq = {k: _t(v).requires_grad_(True) for k, v in P.items()}
    aux2 = ops.RenderAux(); aux2.m_cap = int(aux.M * 1.25) + 16
    rgb2, alpha2, depth2 = ops.render_view(tq["means"], tq["scales"], tq["quats"], tq["opacities"], tq["features_dc"],
                                           tq["features_rest"], cam, _t(BG), not training, 3, aux2)
    cnt, ovf = aux2.M
    assert int(cnt) == aux.M and int(ovf) == 0
    assert torch.equal(rgb2, rgb) and torch.equal(alpha2, alpha)
    ((rgb2 * _t(v_rgb)).sum() + (alpha2 * _t(v_a)).sum()).backward()
    for k in P:
        _grad_close(tq[k].grad.cpu().numpy(), o["grads"][k], scale)


def test_psnr_vs_oracle_full_size(oracle_c):
    """north_star: PSNR vs reference render >= 45 dB at 512x512 / large N."""
    from gaussctrl_amd import gsplat_ops as ops
    from gaussctrl_amd.camera import camera_to_gsplat
    N, W, H = 1000000, 512, 512
    c2w = full_cams(1, 1)[0]
    K = syn.ROUND_INTRINSICS
    o = oracle_full(oracle_c, N, (1, 1), 0, "round", False)
    cam = camera_to_gsplat(c2w, K["fx"], K["fy"], K["cx"], K["cy"], W, H)
    tp = full_scene_gpu(N)
    rgb, alpha, depth = ops.render_view(tp["means"], tp["scales"], tp["quats"], tp["opacities"], tp["features_dc"],
                                        tp["features_rest"], cam, _t(BG), True, 3, None)
    mse = float(((rgb.cpu().numpy().astype(np.float64) - o["rgb"]) ** 2).mean())
    psnr = 10 * math.log10(1.0 / max(mse, 1e-20))
    assert psnr >= 45.0, psnr
    _img_close(rgb.cpu().numpy(), o["rgb"])


def test_empty_and_all_culled():
    from gaussctrl_amd import gsplat_ops as ops
    from gaussctrl_amd.camera import camera_to_gsplat
    c2w = syn.look_at_c2w(np.array([0.0, -2.0, 0.0]), np.zeros(3))
    cam = camera_to_gsplat(c2w, 100.0, 100.0, 32.0, 32.0, 64, 64)
    P = syn.make_gaussians(50, seed=1)
    P["means"][:] = [0, -5, 0]
    tp = {k: _t(v).requires_grad_(True) for k, v in P.items()}
    rgb, alpha, depth = ops.render_view(tp["means"], tp["scales"], tp["quats"], tp["opacities"], tp["features_dc"],
                                        tp["features_rest"], cam, _t(BG), True, 3, None)
    assert torch.allclose(rgb, _t(BG).expand(64, 64, 3)) and float(alpha.abs().max()) == 0.0
    assert bool((depth == 1000.0).all())
    rgb.sum().backward()
    assert float(tp["means"].grad.abs().max()) == 0.0


# ---------------------------------------------------------------------------------------- BASELINE configs 3 and 5
# Full-size scenes and their oracle renders are shared by every test that needs them (this file and test_raster_views_gpu.py): building
# 4 M Gaussians takes 10 s and one serial oracle render 5-15 s, and the same (scene, camera, mode) is wanted by several tests.
_SCENES, _SCENES_GPU, _CAMS, _ORACLE = {}, {}, {}, {}
INTRINSICS = {"bear": syn.BEAR_INTRINSICS, "garden": syn.GARDEN_INTRINSICS, "round": syn.ROUND_INTRINSICS}


def full_scene(N, seed=0):
    if (N, seed) not in _SCENES:
        _SCENES[(N, seed)] = syn.make_gaussians(N, seed=seed)
    return _SCENES[(N, seed)]


def full_scene_gpu(N, seed=0):
    """the scene's six tensors on the GPU (no grad; tests that differentiate make their own leaves with .clone().requires_grad_())"""
    if (N, seed) not in _SCENES_GPU:
        _SCENES_GPU[(N, seed)] = {k: _t(v) for k, v in full_scene(N, seed).items()}
    return _SCENES_GPU[(N, seed)]


def full_cams(n, seed):
    if (n, seed) not in _CAMS:
        _CAMS[(n, seed)] = syn.make_cameras(n, seed=seed)
    return _CAMS[(n, seed)]


def upstream(H, W, seed=5):
    g = np.random.default_rng(seed)
    return g.normal(size=(H, W, 3)).astype(np.float32), g.normal(size=(H, W)).astype(np.float32)


def oracle_full(oracle_c, N, cams, ci, kname, training):
    """oracle_c.render of camera `ci` of full_cams(*cams) on full_scene(N) with background BG and the upstream gradients of upstream();
    memoised for the session"""
    key = (N, cams, ci, kname, training)
    if key not in _ORACLE:
        K = INTRINSICS[kname]
        v_rgb, v_a = upstream(K["H"], K["W"])
        _ORACLE[key] = oracle_c.render(full_scene(N), full_cams(*cams)[ci], K["fx"], K["fy"], K["cx"], K["cy"], K["W"], K["H"], BG,
                                       training=training, v_rgb=v_rgb, v_alpha=v_a)
    return _ORACLE[key]


def _full_parity(oracle_c, N, cams, ci, kname, training):
    """fused product render (fwd [+ depth] + bwd to the six leaf tensors) vs the C oracle at full size."""
    from gaussctrl_amd import gsplat_ops as ops
    from gaussctrl_amd.camera import camera_to_gsplat
    K = INTRINSICS[kname]
    W, H = K["W"], K["H"]
    P = full_scene(N)
    v_rgb, v_a = upstream(H, W)
    o = oracle_full(oracle_c, N, cams, ci, kname, training)
    cam = camera_to_gsplat(full_cams(*cams)[ci], K["fx"], K["fy"], K["cx"], K["cy"], W, H)
    tp = {k: v.clone().requires_grad_(True) for k, v in full_scene_gpu(N).items()}
    aux = ops.RenderAux()
    rgb, alpha, depth = ops.render_view(tp["means"], tp["scales"], tp["quats"], tp["opacities"], tp["features_dc"],
                                        tp["features_rest"], cam, _t(BG), not training, 3, aux)
    _img_close(rgb.detach().cpu().numpy(), o["rgb"])
    _img_close(alpha.detach().cpu().numpy(), o["accumulation"][..., 0])
    mse = float(((rgb.detach().cpu().numpy().astype(np.float64) - o["rgb"]) ** 2).mean())
    assert 10 * math.log10(1.0 / max(mse, 1e-20)) >= 45.0
    assert (aux.radii.cpu().numpy() != o["radii"]).mean() < 1e-4
    assert aux.M <= o["M"] + 4            # tight tile boxes: never more pairs than gsplat's box (the oracle's M)
    print(f"tight tile boxes: M = {aux.M} of gsplat's {o['M']} ({aux.M / max(o['M'], 1):.3f})")
    if not training:
        d = depth.cpu().numpy(); od = o["depth"][..., 0]
        far = (od == 1000.0)
        assert np.array_equal(far, d == 1000.0)
        _img_close(np.where(far, 0, d), np.where(far, 0, od))
    ((rgb * _t(v_rgb)).sum() + (alpha * _t(v_a)).sum()).backward()
    scale = max(np.abs(o["grads"][k]).max() for k in P)
    for k in P:
        _grad_close(tp[k].grad.cpu().numpy(), o["grads"][k], scale)
    return aux.M


@pytest.mark.parametrize("training", [False, True])
def test_config3_garden_2m(oracle_c, training):
    """BASELINE configs[2]: garden intrinsics (/root/reference/data/garden/transforms.json), ~2 M Gaussians."""
    M = _full_parity(oracle_c, 2_000_000, (3, 11), 2, "garden", training)
    print(f"config 3: M = {M}")


@pytest.mark.parametrize("training", [False, True])
def test_config2_bear_1m(oracle_c, training):
    """BASELINE configs[1] exactly as bench.py renders it: 1 M Gaussians, the bear intrinsics of
    /root/reference/data/bear/transforms.json (fx 539.05, fy 538.17, cx 258.74, cy 239.35: non-square focal, off-centre
    principal point), eval (rgb + depth) and training render with backward, against the C oracle."""
    M = _full_parity(oracle_c, 1_000_000, (40, 1), 7, "bear", training)
    print(f"config 2 (bear intrinsics, 1 M): M = {M}")


def test_config5_raster_4m(oracle_c):
    """BASELINE configs[4]: 4 M random Gaussians, random 512x512 cameras (fx=fy=540, cx=cy=256): one camera against the C
    oracle (image, depth, leaf gradients), further cameras through size-independent properties of the binning and the
    compositing: tile bins partition [0, M), keys sorted, tiles inside the image, alpha = 1 - T."""
    from gaussctrl_amd import gsplat_ops as ops
    from gaussctrl_amd.camera import camera_to_gsplat
    N = 4_000_000
    cams = full_cams(256, 1)
    K = syn.ROUND_INTRINSICS
    M = _full_parity(oracle_c, N, (256, 1), 17, "round", True)
    print(f"config 5: M = {M}")
    tp = full_scene_gpu(N)
    W, H = K["W"], K["H"]
    tb = ((W + 15) // 16, (H + 15) // 16, 1)
    T = tb[0] * tb[1]
    for ci in (0, 100, 255):
        cam = camera_to_gsplat(cams[ci], K["fx"], K["fy"], K["cx"], K["cy"], W, H)
        aux = ops.RenderAux()
        rgb, alpha, depth = ops.render_view(tp["means"], tp["scales"], tp["quats"], tp["opacities"], tp["features_dc"],
                                            tp["features_rest"], cam, _t(BG), True, 3, aux)
        assert bool(torch.isfinite(rgb).all()) and float(alpha.min()) >= 0.0 and float(alpha.max()) <= 1.0
        nth = aux.num_tiles_hit
        Mi, keys, ids, bins, cum = ops.bin_and_sort_gaussians(N, aux.xys, aux.depths, aux.radii, nth, tb, want_keys=True,
                                                              tile_boxes=aux.tile_boxes)
        assert Mi == aux.M == int(nth.sum())
        k = keys.cpu().numpy()
        assert np.all(k[1:] >= k[:-1])                                    # tile-major, depth-minor order
        b = bins.cpu().numpy().astype(np.int64)
        tiles = (k >> 32).astype(np.int64)
        assert tiles.min() >= 0 and tiles.max() < T
        nz = b[:, 1] > b[:, 0]
        assert int((b[nz, 1] - b[nz, 0]).sum()) == Mi                     # bins partition [0, M)
        starts = np.sort(b[nz, 0]); ends = np.sort(b[nz, 1])
        assert starts[0] == 0 and ends[-1] == Mi and np.array_equal(starts[1:], ends[:-1])
        idn = ids.cpu().numpy()
        assert idn.min() >= 0 and idn.max() < N
        # every intersection's Gaussian is visible and its depth is the key's low word
        dbits = aux.depths.cpu().numpy().view(np.int32)[idn[:: max(1, Mi // 100000)]]
        assert np.array_equal(dbits.astype(np.int64), (k[:: max(1, Mi // 100000)] & 0xFFFFFFFF))


def test_tight_tile_boxes_same_images_and_gradients():
    """Tight tile boxes (gc_project_sh_fwd_boxes / gc_raster_bin_tiles_boxes) vs gsplat's 3-sigma boxes on scenes that stress the box:
    needles, giants, faint and sub-pixel Gaussians, opacities around the 1/255 threshold.  Images must be BIT-identical (a dropped tile
    contains no pixel that passes the per-pixel test) and the lists get shorter.

    Gradients.  These scenes are ill-conditioned on purpose: a needle's scale / rotation gradient is a cancelling sum whose fp32
    float-atomic order noise is amplified by the projection backward.  Measured (scripts/tight_box_noise.py ->
    profiles/r04_tight_box_noise.txt; same build, same box): the SAME variant run twice differs by up to 6.4e-4 of the tensor's max
    (relative L2 2.9e-4), tight vs gsplat boxes by 1.03e-3 / 4.6e-4, and both sit 8.5e-4 / 5.0e-4 from the float64 gradients of the
    independent restatement -- i.e. the two variants are equally far from the truth and the round-3 bar (1e-3 of max between the two
    variants) sat inside the noise (a second box: same variant twice 1.47e-3 / 6.6e-4).  Criterion now: EACH variant within 1e-2 of max / 5e-3 relative L2 of the fp64 oracle
    (oracle/raster_torch.py::render_grads_tiled, evaluated in float64 on the GPU), and the variants within the same bounds of each
    other: >= 4.7x every distance measured on two boxes (worst: 2.1e-3 / 1.0e-3), all of it on ONE needle's scale / rotation rows."""
    from gaussctrl_amd import gsplat_ops as ops
    from gaussctrl_amd.camera import camera_to_gsplat
    from oracle import raster_torch as rt
    W, H = 320, 240
    K = dict(fx=300.0, fy=290.0, cx=161.3, cy=118.2)
    for seed, sm in ((0, 0.02), (1, 0.08), (2, 0.004)):
        P = syn.make_gaussians(60000, seed=seed, scale_mean=sm)
        g = np.random.default_rng(seed)
        P["scales"][::7, 0] += 2.0                              # needles
        P["scales"][::11] += 1.5                                # giants
        P["opacities"][::5] = g.normal(-5.0, 1.0, size=P["opacities"][::5].shape).astype(np.float32)    # around / below 1/255
        P["opacities"][::13] = 8.0                              # opaque: the alpha >= 1/255 ellipse exceeds gsplat's 3-sigma box
        c2w = syn.make_cameras(2, seed=seed + 3)[1]
        cam = camera_to_gsplat(c2w, K["fx"], K["fy"], K["cx"], K["cy"], W, H)
        v = torch.randn(H, W, 3, generator=torch.Generator().manual_seed(seed)).to(DEV)
        out = {}
        for tight in (True, False):
            tp = {k: _t(x).requires_grad_(True) for k, x in P.items()}
            aux = ops.RenderAux(); aux.tight_boxes = tight
            rgb, alpha, _ = ops.render_view(tp["means"], tp["scales"], tp["quats"], tp["opacities"], tp["features_dc"], tp["features_rest"],
                                            cam, _t(BG), False, 3, aux)
            ((rgb * v).sum() + alpha.sum()).backward()
            out[tight] = (rgb.detach(), alpha.detach(), {k: t.grad.double().cpu().numpy() for k, t in tp.items()}, aux.M, aux)
        assert torch.equal(out[True][0], out[False][0]) and torch.equal(out[True][1], out[False][1])
        assert out[True][3] < out[False][3]
        ref = rt.render_grads_tiled(P, c2w, K["fx"], K["fy"], K["cx"], K["cy"], W, H, BG, v.cpu(), out[False][4].gaussian_ids_sorted,
                                    out[False][4].tile_bins, device=DEV)
        scale = max(float(t.abs().max()) for t in ref.values())
        for k in ref:
            r = ref[k].numpy()
            for name, a, b in (("tight vs fp64", out[True][2][k], r), ("gsplat boxes vs fp64", out[False][2][k], r),
                               ("tight vs gsplat boxes", out[True][2][k], out[False][2][k])):
                d = a - b
                # the measured atomic-order noise sits on the needles' SCALE and ROTATION rows (ill-conditioned projection backward): only those
                # two tensors get the loose bars; means / opacities / colours keep the rasterizer's usual 1e-3 of max, so a 5e-3 regression in a
                # well-conditioned tensor still fails here (advisor finding, round 4)
                loose = k in ("scales", "quats")
                within(f"{k} {name} max-norm", np.abs(d).max(), (1e-2 if loose else 1e-3) * np.abs(r).max() + 1e-6 * scale)
                within(f"{k} {name} rel L2", np.linalg.norm(d), (5e-3 if loose else 1e-3) * np.linalg.norm(r) + 1e-6 * scale)
        print(f"seed {seed}: M tight {out[True][3]} / gsplat {out[False][3]} = {out[True][3] / out[False][3]:.3f}")


def test_sh_degree0_sigmoid_colour(oracle_c):
    """config.sh_degree == 0 branch of the reference (gc_model.py:169): rgbs = sigmoid(features_dc), no SH (K = 1)."""
    from gaussctrl_amd import gsplat_ops as ops
    from gaussctrl_amd.camera import camera_to_gsplat
    N, W, H = 30000, 160, 128
    P = syn.make_gaussians(N, seed=4, sh_degree=0, scale_mean=0.02)
    assert P["features_rest"].shape == (N, 0, 3)
    c2w = syn.make_cameras(1, seed=6)[0]
    g = np.random.default_rng(3)
    v_rgb = g.normal(size=(H, W, 3)).astype(np.float32)
    o = oracle_c.render(P, c2w, 150.0, 151.0, 80.3, 63.1, W, H, BG, training=True, sh_degree_to_use=-1, v_rgb=v_rgb)
    cam = camera_to_gsplat(c2w, 150.0, 151.0, 80.3, 63.1, W, H)
    tp = {k: _t(v).requires_grad_(True) for k, v in P.items()}
    rgb, alpha, _ = ops.render_view(tp["means"], tp["scales"], tp["quats"], tp["opacities"], tp["features_dc"], tp["features_rest"],
                                    cam, _t(BG), False, -1, None)
    # a 160 x 128 image has 61 k values: one knife-edge pixel (3 values) is already 5e-5 of them
    _img_close(rgb.detach().cpu().numpy(), o["rgb"], frac_max=2e-4)
    (rgb * _t(v_rgb)).sum().backward()
    scale = max(np.abs(o["grads"][k]).max() for k in ("means", "scales", "quats", "opacities", "features_dc"))
    for k in ("means", "scales", "quats", "opacities", "features_dc"):
        _grad_close(tp[k].grad.cpu().numpy(), o["grads"][k], scale)


def test_backward_accumulates_into_buffers(oracle_c):
    """RenderAux.grad_into: the fused backward writes (first view) / adds (further views) the six leaf gradients into caller buffers and
    autograd gets None for them -- equal to the sum of the per-view autograd gradients."""
    from gaussctrl_amd import gsplat_ops as ops
    from gaussctrl_amd.camera import camera_to_gsplat
    N, W, H = 30000, 160, 112
    P = syn.make_gaussians(N, seed=2, scale_mean=0.03)
    cams = syn.make_cameras(3, seed=4)
    tp = {k: _t(v).requires_grad_(True) for k, v in P.items()}
    g = torch.Generator(device="cpu").manual_seed(0)
    vs = [torch.randn(H, W, 3, generator=g).to(DEV) for _ in cams]

    def render(c2w, aux):
        cam = camera_to_gsplat(c2w, 150.0, 150.0, W / 2, H / 2, W, H)
        return ops.render_view(tp["means"], tp["scales"], tp["quats"], tp["opacities"], tp["features_dc"], tp["features_rest"], cam, _t(BG),
                               False, 3, aux)[0]
    ref = {k: torch.zeros_like(v) for k, v in tp.items()}
    for c2w, v in zip(cams, vs):
        for p in tp.values():
            p.grad = None
        (render(c2w, ops.RenderAux()) * v).sum().backward()
        for k in tp:
            ref[k] += tp[k].grad
    buf = {k: torch.full_like(v, float("nan")) for k, v in tp.items()}     # the first view must overwrite, not add
    for p in tp.values():
        p.grad = None
    for j, (c2w, v) in enumerate(zip(cams, vs)):
        aux = ops.RenderAux(); aux.grad_into, aux.grad_accumulate = buf, j > 0
        (render(c2w, aux) * v).sum().backward()
    assert all(p.grad is None for p in tp.values())
    scale = max(float(r.abs().max()) for r in ref.values())
    for k in tp:
        assert torch.isfinite(buf[k]).all(), k
        _grad_close(buf[k].cpu().numpy(), ref[k].cpu().numpy(), scale)
