"""Batched views (round 5, include/gaussctrl_hip.h "Batched views"): C cameras of one scene through ONE set of launches.  The reference
renders one camera per get_outputs call (/root/reference/gaussctrl/gc_pipeline.py:124-130, gc_trainer.py:186-201), so the contract tested
here is: view c of the batch == the single-view product path on camera c -- BIT for bit for every integer list and every image (same device
code, the view is a grid dimension / an inner loop), and == the CPU oracle at the rasterizer's usual bars; gradients = the sum over the views."""
import ctypes as C

import numpy as np
import pytest
import torch

from _margins import within
from gaussctrl_amd import synthetic as syn
from test_raster_gpu import BG, DEV, _grad_close, _img_close, _t

pytestmark = pytest.mark.gpu


def _cams(n, W, H, fx, seed=4):
    from gaussctrl_amd.camera import camera_to_gsplat
    return [camera_to_gsplat(c2w, fx, fx * 0.99, W / 2 + 1.3, H / 2 - 2.1, W, H) for c2w in syn.make_cameras(n, seed=seed)], syn.make_cameras(n, seed=seed)


@pytest.mark.parametrize("N,W,H,fx,sm,C", [(30000, 160, 112, 150.0, 0.03, 3), (200000, 512, 512, 540.0, 0.01, 5), (20000, 200, 136, 180.0, 0.03, 11),
                                             (7, 33, 17, 40.0, 0.3, 2)])
def test_render_views_bit_identical_to_single_view(N, W, H, fx, sm, C):
    """every per-view output of the batch -- images, alpha, depth, projection state, tile boxes, sorted id lists, tile bins, counts --
    equals the single-view render_view on that camera exactly (C = 11 crosses the 8-views-per-launch groups of the projection kernel)"""
    from gaussctrl_amd import gsplat_ops as ops
    P = syn.make_gaussians(N, seed=2, scale_mean=sm)
    cams, _ = _cams(C, W, H, fx)
    tp = {k: _t(v) for k, v in P.items()}
    bgs = torch.rand(C, 3, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    aux = ops.RenderAux()
    with torch.no_grad():
        rgb, alpha, depth = ops.render_views(tp["means"], tp["scales"], tp["quats"], tp["opacities"], tp["features_dc"], tp["features_rest"],
                                             cams, bgs, True, 3, aux)
    cnt, ovf = aux.M
    assert rgb.shape == (C, H, W, 3) and alpha.shape == (C, H, W) and depth.shape == (C, H, W) and int(ovf.max()) == 0
    for c, cam in enumerate(cams):
        a1 = ops.RenderAux()
        with torch.no_grad():
            r1, al1, d1 = ops.render_view(tp["means"], tp["scales"], tp["quats"], tp["opacities"], tp["features_dc"], tp["features_rest"],
                                          cam, bgs[c], True, 3, a1)
        assert torch.equal(rgb[c], r1) and torch.equal(alpha[c], al1) and torch.equal(depth[c], d1), c
        assert int(cnt[c]) == a1.M, (c, int(cnt[c]), a1.M)
        for name in ("xys", "radii", "num_tiles_hit", "depths", "tile_boxes"):
            assert torch.equal(getattr(aux, name)[c], getattr(a1, name)), (c, name)
        assert torch.equal(aux.gaussian_ids_sorted[c, :a1.M], a1.gaussian_ids_sorted), c
        assert torch.equal(aux.tile_bins[c], a1.tile_bins) and torch.equal(aux.final_index[c], a1.final_index), c


def test_render_views_vs_oracle(oracle_c):
    """two views of the batch against get_outputs restated on the CPU oracle (images 1e-4 rel, leaf gradients of the SUMMED loss 1e-3 of max)"""
    from gaussctrl_amd import gsplat_ops as ops
    N, W, H, fx = 20000, 200, 136, 180.0
    P = syn.make_gaussians(N, seed=3, scale_mean=0.03)
    cams, c2ws = _cams(2, W, H, fx, seed=7)
    g = np.random.default_rng(2)
    v_rgb = g.normal(size=(2, H, W, 3)).astype(np.float32); v_a = g.normal(size=(2, H, W)).astype(np.float32)
    tp = {k: _t(v).requires_grad_(True) for k, v in P.items()}
    rgb, alpha, _ = ops.render_views(tp["means"], tp["scales"], tp["quats"], tp["opacities"], tp["features_dc"], tp["features_rest"], cams,
                                     _t(BG), False, 3, ops.RenderAux())
    ((rgb * _t(v_rgb)).sum() + (alpha * _t(v_a)).sum()).backward()
    ref_g = {k: np.zeros_like(v, dtype=np.float64) for k, v in P.items()}
    for c in range(2):
        o = oracle_c.render(P, c2ws[c], fx, fx * 0.99, W / 2 + 1.3, H / 2 - 2.1, W, H, BG, training=True, v_rgb=v_rgb[c], v_alpha=v_a[c])
        _img_close(rgb[c].detach().cpu().numpy(), o["rgb"])
        _img_close(alpha[c].detach().cpu().numpy(), o["accumulation"][..., 0])
        for k in P:
            ref_g[k] += o["grads"][k]
    scale = max(np.abs(v).max() for v in ref_g.values())
    for k in P:
        _grad_close(tp[k].grad.cpu().numpy(), ref_g[k], scale)


def test_render_views_gradients_sum_over_views_and_accumulate():
    """leaf gradients of a batched training render = sum over the views of the single-view gradients (float atomics: 1e-3 of max); with
    RenderAux.grad_into the batch writes / adds into the caller's flat buffers exactly like the per-view path of bench.py"""
    from gaussctrl_amd import gsplat_ops as ops
    N, W, H, C = 30000, 160, 112, 4
    P = syn.make_gaussians(N, seed=2, scale_mean=0.03)
    cams, _ = _cams(C, W, H, 150.0)
    tp = {k: _t(v).requires_grad_(True) for k, v in P.items()}
    g = torch.Generator(device="cpu").manual_seed(0)
    vs = torch.randn(C, H, W, 3, generator=g).to(DEV)
    args = lambda: (tp["means"], tp["scales"], tp["quats"], tp["opacities"], tp["features_dc"], tp["features_rest"])
    ref = {k: torch.zeros_like(v) for k, v in tp.items()}
    for c in range(C):
        for p in tp.values():
            p.grad = None
        (ops.render_view(*args(), cams[c], _t(BG), False, 3, ops.RenderAux())[0] * vs[c]).sum().backward()
        for k in tp:
            ref[k] += tp[k].grad
    for p in tp.values():
        p.grad = None
    (ops.render_views(*args(), cams, _t(BG), False, 3, ops.RenderAux())[0] * vs).sum().backward()
    scale = max(float(r.abs().max()) for r in ref.values())
    for k in tp:
        _grad_close(tp[k].grad.cpu().numpy(), ref[k].cpu().numpy(), scale)
    # grad_into: first batch overwrites (NaN-filled buffers), the second adds
    buf = {k: torch.full_like(v, float("nan")) for k, v in tp.items()}
    for p in tp.values():
        p.grad = None
    for j in range(2):
        aux = ops.RenderAux(); aux.grad_into, aux.grad_accumulate = buf, j > 0
        (ops.render_views(*args(), cams[2 * j:2 * j + 2], _t(BG), False, 3, aux)[0] * vs[2 * j:2 * j + 2]).sum().backward()
    assert all(p.grad is None for p in tp.values())
    for k in tp:
        assert torch.isfinite(buf[k]).all(), k
        _grad_close(buf[k].cpu().numpy(), ref[k].cpu().numpy(), scale)


def test_project_sh_bwd_views_bit_identical_to_accumulating_single_view_launches():
    """the per-Gaussian backward has no atomics: with the SAME incoming gradients the batched kernel (sum in registers, one store) equals C
    accumulating single-view launches (write, then +=) bit for bit -- it adds the views in the same order"""
    from gaussctrl_amd import _lib as L, gsplat_ops as ops
    lib = L.lib()
    N, W, H, Cn = 50000, 256, 256, 11
    P = syn.make_gaussians(N, seed=5, scale_mean=0.03)
    cams, _ = _cams(Cn, W, H, 300.0)
    tp = {k: _t(v) for k, v in P.items()}
    aux = ops.RenderAux()
    with torch.no_grad():
        ops.render_views(tp["means"], tp["scales"], tp["quats"], tp["opacities"], tp["features_dc"], tp["features_rest"], cams, _t(BG), False, 3, aux)
    g = torch.Generator(device=DEV).manual_seed(3)
    v_xy = torch.randn(Cn, N, 2, device=DEV, generator=g); v_con = torch.randn(Cn, N, 3, device=DEV, generator=g)
    v_col = torch.randn(Cn, N, 3, device=DEV, generator=g); v_op = torch.randn(Cn, N, device=DEV, generator=g)
    # forward colours of each view (the clamp mask) and conics: recomputed through the single-view entry point
    outs = {}
    for mode in ("single", "views"):
        vm = torch.full((N, 3), float("nan"), device=DEV); vls = torch.full((N, 3), float("nan"), device=DEV)
        vq = torch.full((N, 4), float("nan"), device=DEV); vop = torch.full((N,), float("nan"), device=DEV)
        vdc = torch.full((N, 3), float("nan"), device=DEV); vrest = torch.full((N, 15, 3), float("nan"), device=DEV)
        m, ls, q, op = tp["means"], tp["scales"], tp["quats"], tp["opacities"].reshape(-1).contiguous()
        rg, con = [], []
        for c, cam in enumerate(cams):
            a1 = ops.RenderAux()
            xys = torch.empty(N, 2, device=DEV); dep = torch.empty(N, device=DEV); rad = torch.empty(N, dtype=torch.int32, device=DEV)
            cn = torch.empty(N, 3, device=DEV); nth = torch.empty(N, dtype=torch.int32, device=DEV); rgbs = torch.empty(N, 3, device=DEV)
            opac = torch.empty(N, device=DEV)
            V, Pm, O = L.host_floats(cam["viewmat"]), L.host_floats(cam["fullproj"]), L.host_floats(cam["origin"])
            L.check(lib.gc_project_sh_fwd(L.i64(N), L.ptr(m), L.ptr(ls), L.ptr(q), L.ptr(op), L.ptr(tp["features_dc"]), L.ptr(tp["features_rest"]),
                                          L.i32(3), L.i32(3), V, Pm, O, L.f32(cam["fx"]), L.f32(cam["fy"]), L.f32(cam["cx"]), L.f32(cam["cy"]),
                                          L.i32(H), L.i32(W), L.i32(16), L.i32(16), L.f32(0.01), L.ptr(xys), L.ptr(dep), L.ptr(rad), L.ptr(cn),
                                          L.ptr(nth), L.ptr(rgbs), L.ptr(opac), L.stream_ptr()), "fwd")
            assert torch.equal(rad, aux.radii[c])
            rg.append(rgbs); con.append(cn)
            if mode == "single":
                fn = lib.gc_project_sh_bwd_accumulate if c > 0 else lib.gc_project_sh_bwd
                L.check(fn(L.i64(N), L.ptr(m), L.ptr(ls), L.ptr(q), L.ptr(op), L.ptr(rgbs), L.i32(3), L.i32(3), V, Pm, O, L.f32(cam["fx"]),
                           L.f32(cam["fy"]), L.f32(cam["cx"]), L.f32(cam["cy"]), L.i32(H), L.i32(W), L.ptr(rad), L.ptr(cn), L.ptr(v_xy[c]),
                           L.ptr(v_con[c]), L.ptr(v_col[c]), L.ptr(v_op[c]), L.ptr(vm), L.ptr(vls), L.ptr(vq), L.ptr(vop), L.ptr(vdc),
                           L.ptr(vrest), L.stream_ptr()), "bwd")
        if mode == "views":
            rgbs_all = torch.stack(rg).contiguous(); con_all = torch.stack(con).contiguous()
            L.check(lib.gc_project_sh_bwd_views(L.i64(N), L.i32(Cn), L.i32(0), L.ptr(m), L.ptr(ls), L.ptr(q), L.ptr(op), L.ptr(rgbs_all), L.i32(3),
                                                L.i32(3), ops._cams_host(cams), L.i32(H), L.i32(W), L.ptr(aux.radii), L.ptr(con_all), L.ptr(v_xy),
                                                L.ptr(v_con), L.ptr(v_col), L.ptr(v_op), L.ptr(vm), L.ptr(vls), L.ptr(vq), L.ptr(vop), L.ptr(vdc),
                                                L.ptr(vrest), L.stream_ptr()), "bwd views")
        outs[mode] = (vm, vls, vq, vop, vdc, vrest)
    # views are summed in groups of 8 by the batched entry (group sums added to the buffer): bit-identity holds within the first group;
    # across groups the association differs ((g0..g7) + (g8..g10) vs a running sum) -- compare at 1-ulp-level tolerance there
    for a, b, name in zip(outs["single"], outs["views"], ("means", "scales", "quats", "opacities", "features_dc", "features_rest")):
        assert torch.isfinite(b).all(), name
        d = float((a - b).abs().max()); s = float(a.abs().max())
        within(f"{name}: |single - views| / max", d / max(s, 1e-30), 2e-6)
    # ... and exactly, for a batch that fits one group
    Cs = 8
    vm2 = [torch.full_like(t, float("nan")) for t in outs["single"]]
    vm1 = [torch.full_like(t, float("nan")) for t in outs["single"]]
    m, ls, q, op = tp["means"], tp["scales"], tp["quats"], tp["opacities"].reshape(-1).contiguous()
    rgbs_all = torch.stack(rg[:Cs]).contiguous(); con_all = torch.stack(con[:Cs]).contiguous()
    L.check(lib.gc_project_sh_bwd_views(L.i64(N), L.i32(Cs), L.i32(0), L.ptr(m), L.ptr(ls), L.ptr(q), L.ptr(op), L.ptr(rgbs_all), L.i32(3), L.i32(3),
                                        ops._cams_host(cams[:Cs]), L.i32(H), L.i32(W), L.ptr(aux.radii[:Cs].contiguous()), L.ptr(con_all),
                                        L.ptr(v_xy[:Cs].contiguous()), L.ptr(v_con[:Cs].contiguous()), L.ptr(v_col[:Cs].contiguous()),
                                        L.ptr(v_op[:Cs].contiguous()), *[L.ptr(t) for t in vm2], L.stream_ptr()), "bwd views 8")
    for c, cam in enumerate(cams[:Cs]):
        V, Pm, O = L.host_floats(cam["viewmat"]), L.host_floats(cam["fullproj"]), L.host_floats(cam["origin"])
        fn = lib.gc_project_sh_bwd_accumulate if c > 0 else lib.gc_project_sh_bwd
        L.check(fn(L.i64(N), L.ptr(m), L.ptr(ls), L.ptr(q), L.ptr(op), L.ptr(rg[c]), L.i32(3), L.i32(3), V, Pm, O, L.f32(cam["fx"]), L.f32(cam["fy"]),
                   L.f32(cam["cx"]), L.f32(cam["cy"]), L.i32(H), L.i32(W), L.ptr(aux.radii[c].contiguous()), L.ptr(con[c]), L.ptr(v_xy[c]),
                   L.ptr(v_con[c]), L.ptr(v_col[c]), L.ptr(v_op[c]), *[L.ptr(t) for t in vm1], L.stream_ptr()), "bwd")
    for a, b, name in zip(vm1, vm2, ("means", "scales", "quats", "opacities", "features_dc", "features_rest")):
        assert torch.equal(a, b), (name, float((a - b).abs().max()))


def test_render_views_sync_free_capacity_and_overflow():
    from gaussctrl_amd import gsplat_ops as ops
    N, W, H, C = 30000, 160, 112, 3
    P = syn.make_gaussians(N, seed=2, scale_mean=0.03)
    cams, _ = _cams(C, W, H, 150.0)
    tp = {k: _t(v) for k, v in P.items()}
    a0 = ops.RenderAux()
    with torch.no_grad():
        r0 = ops.render_views(tp["means"], tp["scales"], tp["quats"], tp["opacities"], tp["features_dc"], tp["features_rest"], cams, _t(BG), False, 3, a0)[0]
        mmax = int(a0.M[0].max())
        a1 = ops.RenderAux(); a1.m_cap = int(mmax * 1.3) + 64
        r1 = ops.render_views(tp["means"], tp["scales"], tp["quats"], tp["opacities"], tp["features_dc"], tp["features_rest"], cams, _t(BG), False, 3, a1)[0]
        assert torch.equal(r0, r1) and int(a1.M[1].max()) == 0 and torch.equal(a1.M[0], a0.M[0])
        a2 = ops.RenderAux(); a2.m_cap = mmax // 2                      # too small for at least the largest view: flagged, no crash
        ops.render_views(tp["means"], tp["scales"], tp["quats"], tp["opacities"], tp["features_dc"], tp["features_rest"], cams, _t(BG), False, 3, a2)
        ovf = a2.M[1].cpu().numpy(); cnt = a2.M[0].cpu().numpy()
        assert np.array_equal(ovf, (cnt > mmax // 2).astype(np.int32)) and ovf.max() == 1


def test_l1_ssim_loss_views_matches_per_image():
    from gaussctrl_amd.train_ops import l1_ssim_loss, l1_ssim_loss_views
    g = torch.Generator(device=DEV).manual_seed(0)
    B, H, W = 5, 136, 200
    x = torch.rand(B, H, W, 3, device=DEV, generator=g).requires_grad_(True); y = torch.rand(B, H, W, 3, device=DEV, generator=g)
    wts = torch.tensor([1.0, 0.5, 2.0, 1.0, 0.25], device=DEV)
    lv = l1_ssim_loss_views(x, y, 0.2)
    (lv * wts).sum().backward()
    gv = x.grad.clone(); x.grad = None
    for b in range(B):
        xb = x[b].detach().clone().requires_grad_(True)
        l1 = l1_ssim_loss(xb, y[b], 0.2)
        (l1 * wts[b]).backward()
        within("loss value", abs(float(lv[b]) - float(l1)), 2e-6)
        assert torch.equal(gv[b], xb.grad), b                           # same stencil kernels, per-image slot lines


def test_depth_order_views_with_and_without_pairs():
    """gc_raster_depth_order_views: the (depths, radii) form and the depth_pairs form give the same order / scan / counts as C single calls"""
    from gaussctrl_amd import _lib as L, gsplat_ops as ops
    lib = L.lib()
    N, Cn = 70001, 3
    g = torch.Generator(device=DEV).manual_seed(1)
    depths = torch.rand(Cn, N, device=DEV, generator=g) * 50 + 0.01
    depths[:, 0:N - 3:7] = depths[:, 3:N:7]                                          # ties: stable by Gaussian id
    radii = (torch.rand(Cn, N, device=DEV, generator=g) > 0.3).to(torch.int32) * 5
    nth = torch.randint(0, 9, (Cn, N), device=DEV, generator=g, dtype=torch.int32) * (radii > 0)
    st = L.stream_ptr()
    outs = []
    for use_pairs in (False, True):
        order = torch.empty(Cn, N, dtype=torch.int32, device=DEV); cum = torch.empty_like(order); cnt = torch.empty(Cn, dtype=torch.int32, device=DEV)
        wb = int(lib.gc_raster_depth_order_views_workspace_bytes(L.i64(N), L.i32(Cn)))
        ws = torch.empty(wb, dtype=torch.uint8, device=DEV)
        pairs = None
        if use_pairs:
            key = torch.where(radii > 0, depths.view(torch.int32), torch.full_like(radii, -1))
            pairs = torch.stack([key, torch.arange(N, device=DEV, dtype=torch.int32).expand(Cn, N)], -1).contiguous()
        L.check(lib.gc_raster_depth_order_views(L.i64(N), L.i32(Cn), L.ptr(None if use_pairs else depths), L.ptr(None if use_pairs else radii),
                                                L.ptr(pairs), L.ptr(nth), L.ptr(order), L.ptr(cum), L.ptr(cnt), L.ptr(ws), C.c_size_t(wb), st), "views")
        outs.append((order, cum, cnt))
    for c in range(Cn):
        o1 = torch.empty(N, dtype=torch.int32, device=DEV); c1 = torch.empty_like(o1); n1 = torch.empty(1, dtype=torch.int32, device=DEV)
        wb = int(lib.gc_raster_depth_order_workspace_bytes(L.i64(N)))
        ws = torch.empty(wb, dtype=torch.uint8, device=DEV)
        L.check(lib.gc_raster_depth_order(L.i64(N), L.ptr(depths[c].contiguous()), L.ptr(radii[c].contiguous()), L.ptr(nth[c].contiguous()), L.ptr(o1),
                                          L.ptr(c1), L.ptr(n1), L.ptr(ws), C.c_size_t(wb), st), "single")
        # reference order by torch: stable sort of (key, id)
        key = torch.where(radii[c] > 0, depths[c].view(torch.int32).to(torch.int64), torch.full((N,), 0xFFFFFFFF, dtype=torch.int64, device=DEV))
        ref = torch.sort(key, stable=True).indices.to(torch.int32)
        for order, cum, cnt in outs:
            assert torch.equal(order[c], o1) and torch.equal(cum[c], c1) and int(cnt[c]) == int(n1)
        assert torch.equal(o1, ref)
        assert torch.equal(c1.to(torch.int64), torch.cumsum(nth[c][ref.long()].to(torch.int64), 0))


# ---------------------------------------------------------------------------------------- the batched path AT THE SIZES bench.py times it
@pytest.mark.parametrize("N,kname,cams,views", [(1_000_000, "bear", (40, 1), (7, 19, 33)),                              # configs[1]: chunk_size 3
                                                  (4_000_000, "round", (256, 1), (17, 0, 100, 255, 3, 64, 128, 200))])    # configs[4]: 8 cameras per launch set
def test_render_views_full_size(oracle_c, N, kname, cams, views):
    """bench.py renders every view through gsplat_ops.render_views in sync-free capacity mode (RenderAux.m_cap: device-side counts, no host
    round trip) -- the eval renders (rgb + depth + alpha) and the training renders with the fused backward writing the batch's gradient SUM into
    one flat buffer (RenderAux.grad_into / dist.FlatGrads) -- at N = 1 M, C = 3 (BASELINE configs[1], the bear intrinsics) and, with
    `--workload raster`, N = 4 M, C = 8 (configs[4]).  Here exactly those calls, at exactly those sizes, against the C oracle:
      * the FIRST and the LAST view of the batch (per-view workspace offsets, blockIdx.y / z view indexing, the second group of camera
        arguments) vs oracle_c.render: rgb / alpha / depth at the 1e-4 bars of test_raster_gpu._img_close, PSNR >= 45 dB;
      * every OTHER view bit-identical to the single-view entry on its camera (which test_raster_gpu.py checks against the oracle at full size);
      * leaf gradients of the summed loss of the first and last view (the other views get a zero upstream gradient, so their launches run but
        add nothing) vs the sum of the two oracle gradients, 1e-3 of max, read from the flat buffer; no .grad tensors are created;
      * no capacity overflow with the margin bench.py uses (1.3 x the largest count + 1 024)."""
    import math
    from gaussctrl_amd import gsplat_ops as ops
    from gaussctrl_amd.camera import camera_to_gsplat
    from gaussctrl_amd.dist import FlatGrads
    from test_raster_gpu import INTRINSICS, full_cams, full_scene, full_scene_gpu, oracle_full, upstream
    K = INTRINSICS[kname]
    W, H = K["W"], K["H"]
    C = len(views)
    c2ws = full_cams(*cams)
    gcams = [camera_to_gsplat(c2ws[i], K["fx"], K["fy"], K["cx"], K["cy"], W, H) for i in views]
    P = full_scene(N)
    tp = {k: v.clone().requires_grad_(True) for k, v in full_scene_gpu(N).items()}
    args = (tp["means"], tp["scales"], tp["quats"], tp["opacities"], tp["features_dc"], tp["features_rest"])
    ends = (0, C - 1)
    # sizing pass (what bench.py's first frames do: counts read back once), then the sync-free calls proper
    a0 = ops.RenderAux()
    with torch.no_grad():
        ops.render_views(*args, gcams, _t(BG), True, 3, a0)
    cap = int(int(a0.M[0].max()) * 1.3) + 1024
    # ---- eval batch: rgb + depth + alpha, sync-free
    ae = ops.RenderAux(); ae.m_cap = cap
    with torch.no_grad():
        rgb, alpha, depth = ops.render_views(*args, gcams, _t(BG), True, 3, ae)
    assert int(ae.M[1].max()) == 0, "capacity overflow with bench.py's margin"
    assert torch.equal(ae.M[0], a0.M[0])
    for c in range(C):
        if c in ends:
            o = oracle_full(oracle_c, N, cams, views[c], kname, False)
            _img_close(rgb[c].cpu().numpy(), o["rgb"])
            _img_close(alpha[c].cpu().numpy(), o["accumulation"][..., 0])
            mse = float(((rgb[c].cpu().numpy().astype(np.float64) - o["rgb"]) ** 2).mean())
            assert 10 * math.log10(1.0 / max(mse, 1e-20)) >= 45.0
            d = depth[c].cpu().numpy(); od = o["depth"][..., 0]
            far = (od == 1000.0)
            assert np.array_equal(far, d == 1000.0)
            _img_close(np.where(far, 0, d), np.where(far, 0, od))
            assert int(ae.M[0][c]) <= o["M"] + 4
        else:
            a1 = ops.RenderAux()
            with torch.no_grad():
                r1, al1, d1 = ops.render_view(*args, gcams[c], _t(BG), True, 3, a1)
            assert torch.equal(rgb[c], r1) and torch.equal(alpha[c], al1) and torch.equal(depth[c], d1), c
            assert int(ae.M[0][c]) == a1.M
    del rgb, alpha, depth
    # ---- training batch: sync-free, fused backward into ONE flat buffer (pre-filled with NaN: the first batch must overwrite every element)
    fg = FlatGrads({k: v for k, v in tp.items()})
    fg.flat.fill_(float("nan"))
    at = ops.RenderAux(); at.m_cap = cap
    at.grad_into, at.grad_accumulate = fg.views, False
    rgb, alpha, _ = ops.render_views(*args, gcams, _t(BG), False, 3, at)
    v_rgb, v_a = upstream(H, W)
    up_rgb = torch.zeros(C, H, W, 3, device=DEV); up_a = torch.zeros(C, H, W, device=DEV)
    for c in ends:
        up_rgb[c] = _t(v_rgb); up_a[c] = _t(v_a)
    ((rgb * up_rgb).sum() + (alpha * up_a).sum()).backward()
    assert int(at.M[1].max()) == 0
    assert all(p.grad is None for p in tp.values()), "grad_into: autograd must not materialise .grad tensors"
    assert bool(torch.isfinite(fg.flat).all())
    ref_g = {k: np.zeros(v.shape, np.float64) for k, v in P.items()}
    for c in ends:
        o = oracle_full(oracle_c, N, cams, views[c], kname, True)
        _img_close(rgb[c].detach().cpu().numpy(), o["rgb"])
        _img_close(alpha[c].detach().cpu().numpy(), o["accumulation"][..., 0])
        for k in P:
            ref_g[k] += o["grads"][k]
    scale = max(np.abs(v).max() for v in ref_g.values())
    for k in P:
        _grad_close(fg.views[k].cpu().numpy(), ref_g[k], scale)


# ---------------------------------------------------------------------------------------- round 6: the gather-free depth order
@pytest.mark.parametrize("N,W,H,fx,sm,C,behind", [(30000, 160, 112, 150.0, 0.03, 3, False), (200000, 512, 512, 540.0, 0.01, 5, False),
                                                    (20000, 200, 136, 180.0, 0.03, 9, True), (7, 33, 17, 40.0, 0.3, 2, False),
                                                    (4097, 64, 64, 70.0, 0.05, 1, True)])
def test_sorted_boxes_chain_bit_identical_to_the_pair_chain(N, W, H, fx, sm, C, behind):
    """gc_raster_order_boxes_views + gc_raster_bin_sorted_views (the packed tight box rides through the depth sort as a third word of the item, the
    Gaussians the projection culled are dropped by the first radix pass, scan and emission read sorted arrays: RenderAux.sorted_boxes = True, the
    default) against the round-5 chain gc_raster_depth_order_views + gc_raster_bin_tiles_views (pairs through four passes, then
    num_tiles_hit[order[j]] and tile_boxes[order[j]] gathered): the same stable (depth bits, id) order over the same visible set, so every
    per-view output -- counts, sorted id lists, tile bins, images, depth, final indices -- must be BIT-identical.  `behind`: the scene is moved so
    that roughly half of the Gaussians are behind some of the cameras (culled: the items the first pass drops); N = 4097 crosses a radix block."""
    from gaussctrl_amd import gsplat_ops as ops
    P = syn.make_gaussians(N, seed=5, scale_mean=sm)
    if behind:
        P["means"][:, :] *= 3.0                                  # a cloud much larger than the camera orbit: many centres behind / outside
    cams, _ = _cams(C, W, H, fx, seed=9)
    tp = {k: _t(v) for k, v in P.items()}
    bgs = torch.rand(C, 3, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    outs = {}
    for mode in (True, False):
        aux = ops.RenderAux(); aux.sorted_boxes = mode
        with torch.no_grad():
            rgb, alpha, depth = ops.render_views(tp["means"], tp["scales"], tp["quats"], tp["opacities"], tp["features_dc"], tp["features_rest"],
                                                 cams, bgs, True, 3, aux)
        outs[mode] = (rgb, alpha, depth, aux)
    (r1, a1, d1, x1), (r0, a0, d0, x0) = outs[True], outs[False]
    assert torch.equal(x1.M[0], x0.M[0]) and int(x1.M[1].max()) == 0 and int(x0.M[1].max()) == 0
    if behind:
        vis = (x1.radii > 0).float().mean(dim=1)
        assert float(vis.min()) < 0.9, "the test scene should cull a good part of the Gaussians for some camera"
    for c in range(C):
        m = int(x1.M[0][c])
        assert torch.equal(x1.gaussian_ids_sorted[c, :m], x0.gaussian_ids_sorted[c, :m]), c
        assert torch.equal(x1.tile_bins[c], x0.tile_bins[c]) and torch.equal(x1.final_index[c], x0.final_index[c]), c
    assert torch.equal(r1, r0) and torch.equal(a1, a0) and torch.equal(d1, d0)


def test_sorted_boxes_chain_all_culled_and_capacity():
    """every Gaussian behind the camera: zero intersections, background image; and the sync-free capacity / overflow protocol on the new chain"""
    from gaussctrl_amd import gsplat_ops as ops
    from gaussctrl_amd.camera import camera_to_gsplat
    c2w = syn.look_at_c2w(np.array([0.0, -2.0, 0.0]), np.zeros(3))
    cams = [camera_to_gsplat(c2w, 100.0, 100.0, 32.0, 32.0, 64, 64)] * 2
    P = syn.make_gaussians(500, seed=1)
    P["means"][:] = [0, -5, 0]
    tp = {k: _t(v) for k, v in P.items()}
    aux = ops.RenderAux()
    with torch.no_grad():
        rgb, alpha, depth = ops.render_views(tp["means"], tp["scales"], tp["quats"], tp["opacities"], tp["features_dc"], tp["features_rest"], cams,
                                             _t(BG), True, 3, aux)
    assert int(aux.M[0].max()) == 0 and float(alpha.abs().max()) == 0.0 and torch.allclose(rgb, _t(BG).expand(2, 64, 64, 3))
    N, W, H, C = 30000, 160, 112, 3
    P = syn.make_gaussians(N, seed=2, scale_mean=0.03)
    cams, _ = _cams(C, W, H, 150.0)
    tp = {k: _t(v) for k, v in P.items()}
    a0 = ops.RenderAux()
    with torch.no_grad():
        ops.render_views(tp["means"], tp["scales"], tp["quats"], tp["opacities"], tp["features_dc"], tp["features_rest"], cams, _t(BG), False, 3, a0)
        mmax = int(a0.M[0].max())
        a2 = ops.RenderAux(); a2.m_cap = mmax // 2
        ops.render_views(tp["means"], tp["scales"], tp["quats"], tp["opacities"], tp["features_dc"], tp["features_rest"], cams, _t(BG), False, 3, a2)
    ovf = a2.M[1].cpu().numpy(); cnt = a2.M[0].cpu().numpy()
    assert np.array_equal(ovf, (cnt > mmax // 2).astype(np.int32)) and ovf.max() == 1
