"""Which kernel of a SECOND process disturbs the fused projection kernel of this one?  (scripts/project_repro_stress.py: lanes 48..63 of single
wavefronts read wrong inputs now and then when two pipelines share a GPU.)
    python scripts/corun_bisect.py victim SECONDS            projection calls on static inputs, every output compared with the first call's
    python scripts/corun_bisect.py load CLASS SECONDS        one kernel class in a loop (see LOADS)"""
import collections, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gaussctrl_amd.sd import ops
ops.configure(ops.options_from_env())
dev = "cuda:0"
role = sys.argv[1]
DT = torch.float16 if os.environ.get("GC_CORUN_DT", "f16") == "f16" else torch.bfloat16


def rnd(shape, seed, s=1.0):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    return (torch.randn(*shape, device=dev, generator=g) * s).to(DT)


if role in ("victim", "victim_unfused"):
    from gaussctrl_amd import _lib as L, gsplat_ops as G, synthetic as syn
    from gaussctrl_amd.camera import camera_to_gsplat
    secs = float(sys.argv[2])
    N, W, H = 20000, 128, 128
    P = {k: torch.as_tensor(v, dtype=torch.float32).to(dev).contiguous() for k, v in syn.make_gaussians(N, seed=0, scale_mean=0.03).items()}
    c2w = syn.make_cameras(6, seed=1)
    lib = L.lib()
    NAMES = ("xys", "depths", "radii", "conics", "num_tiles_hit", "rgbs", "opac", "boxes")

    def project(i):
        cam = camera_to_gsplat(c2w[i].detach().cpu().numpy() if hasattr(c2w[i], "detach") else c2w[i], 140.0, 140.0, 64.0, 64.0, W, H)
        V, Pm, O = L.host_floats(cam["viewmat"]), L.host_floats(cam["fullproj"]), L.host_floats(cam["origin"])
        xys = torch.empty(N, 2, device=dev); depths = torch.empty(N, device=dev)
        radii = torch.empty(N, dtype=torch.int32, device=dev); conics = torch.empty(N, 3, device=dev)
        nth = torch.empty(N, dtype=torch.int32, device=dev); rgbs = torch.empty(N, 3, device=dev); opac = torch.empty(N, device=dev)
        boxes = torch.empty(N, dtype=torch.int32, device=dev)
        L.check(lib.gc_project_sh_fwd_boxes(
            L.i64(N), L.ptr(P["means"]), L.ptr(P["scales"]), L.ptr(P["quats"]), L.ptr(P["opacities"].reshape(-1)), L.ptr(P["features_dc"]),
            L.ptr(P["features_rest"]), L.i32(3), L.i32(3), V, Pm, O, L.f32(140.0), L.f32(140.0), L.f32(64.0), L.f32(64.0), L.i32(H), L.i32(W),
            L.i32(8), L.i32(8), L.f32(0.01), L.ptr(xys), L.ptr(depths), L.ptr(radii), L.ptr(conics), L.ptr(nth), L.ptr(rgbs), L.ptr(opac),
            L.ptr(boxes), L.stream_ptr()), "gc_project_sh_fwd_boxes")
        return (xys, depths, radii, conics, nth, rgbs, opac, boxes)

    if role == "victim_unfused":          # the gsplat-shaped operators: projection (with cov3d) and SH as separate kernels
        NAMES = ("xys", "depths", "radii", "conics", "num_tiles_hit", "cov3d", "colors")
        scl = P["scales"].exp().contiguous(); qn = torch.nn.functional.normalize(P["quats"], dim=1).contiguous()
        coeffs = torch.cat([P["features_dc"][:, None, :], P["features_rest"]], 1).contiguous()

        def project(i):
            cam = camera_to_gsplat(c2w[i], 140.0, 140.0, 64.0, 64.0, W, H)
            vm = torch.tensor(cam["viewmat"], dtype=torch.float32).reshape(3, 4); pm = torch.tensor(cam["fullproj"], dtype=torch.float32).reshape(4, 4)
            xys, depths, radii, conics, nth, cov3d = G.project_gaussians(P["means"], scl, 1.0, qn, vm, pm, 140.0, 140.0, 64.0, 64.0, H, W, (8, 8, 1))
            vd = P["means"] - torch.tensor(cam["origin"], dtype=torch.float32, device=dev)
            vd = (vd / vd.norm(dim=1, keepdim=True)).contiguous()
            col = G.spherical_harmonics(3, vd, coeffs)
            return (xys, depths, radii, conics, nth, cov3d, col)

    names_bad = collections.Counter()
    ref = [project(i) for i in range(6)]
    torch.cuda.synchronize()
    lanes = collections.Counter(); bad = total = 0
    side = None
    if os.environ.get("GC_CORUN_SAME_PROCESS", "0") == "1":        # the disturbing matmuls on a second stream of THIS process
        side = torch.cuda.Stream()
        mx, mw = rnd((1536, 1280), 1), rnd((1280, 1280), 2)
    t0 = time.time()
    while time.time() - t0 < secs:
        if side is not None:
            with torch.cuda.stream(side):
                for _ in range(40):
                    mx @ mw
        for i in range(6):
            out = project(i)
            total += 1
            ds = [(a != b).reshape(N, -1).any(1) for a, b in zip(out, ref[i])]
            d = torch.stack(ds).any(0)
            if bool(d.any()):
                bad += 1
                names_bad[",".join(n for n, x in zip(NAMES, ds) if bool(x.any()))] += 1
                for j in d.nonzero().flatten().tolist():
                    lanes[j % 64 // 16] += 1
    print(f"{role}: {bad} deviating calls of {total}; wrong Gaussians by quarter of the wavefront {dict(sorted(lanes.items()))}; outputs {dict(names_bad.most_common(8))}", flush=True)
else:
    cls, secs = sys.argv[2], float(sys.argv[3])
    if cls == "inversion":
        import test_dist_gpu as T
        pipe, model = T._build(1, 0, -1)
        td = pipe.datamanager.train_data

        def step():
            for t in td:
                for k in ("z_0_image", "unedited_image", "depth_image"):
                    t.pop(k, None)
            pipe.render_reverse([1, 3, 5])
    elif cls == "linear":          # 8-wave LDS-DMA GEMM
        x, w, b = rnd((1536, 1280), 1), rnd((1280, 1280), 2, 0.03), rnd((1280,), 3)
        step = lambda: ops.linear(x, w, b)
    elif cls == "linear_small":    # few rows: the split-K planner
        x, w, b = rnd((128, 1280), 1), rnd((320, 1280), 2, 0.03), rnd((320,), 3)
        step = lambda: ops.linear(x, w, b)
    elif cls == "conv":
        x, w, b = rnd((6, 16, 16, 320), 1), rnd((320, 9 * 320), 2, 0.02), rnd((320,), 3)
        step = lambda: ops.conv3x3(x, w, b)
    elif cls == "conv8":
        x, w, b = rnd((6, 4, 4, 1280), 1), rnd((1280, 9 * 1280), 2, 0.01), rnd((1280,), 3)
        step = lambda: ops.conv3x3(x, w, b)
    elif cls.startswith("attn"):
        D = int(cls[4:]); heads = 8; L_ = {40: 256, 80: 64, 160: 16}[D]; f = 6
        q, k, v = rnd((2 * f, L_, heads * D), 1), rnd((2 * f, L_, heads * D), 2), rnd((2 * f, L_, heads * D), 3)
        Lp = (L_ + 7) // 8 * 8
        vt = torch.zeros(2 * f, heads * D, Lp, dtype=DT, device=dev); vt[:, :, :L_] = v.transpose(1, 2)
        sets = [(-1, 0.6)] + [(r, 0.1) for r in range(4)]
        step = lambda: ops.attention(q, k, vt, heads, sets, f, Lk=L_)
    elif cls == "gn":
        x, g_, b = rnd((6, 16, 16, 320), 1), rnd((320,), 2), rnd((320,), 3)
        step = lambda: ops.groupnorm(x, g_, b, 32, 1e-5, True)
    elif cls == "ln":
        x, g_, b = rnd((6, 256, 320), 1), rnd((320,), 2), rnd((320,), 3)
        step = lambda: ops.layernorm(x, g_, b)
    elif cls == "torch_mm":
        x, w = rnd((1536, 1280), 1), rnd((1280, 1280), 2)
        step = lambda: x @ w
    elif cls == "torch_ew":
        x = rnd((1 << 24,), 1)
        step = lambda: (x * 1.0001 + 0.5).sin()
    else:
        raise SystemExit(f"unknown load {cls}")
    n = 0
    t0 = time.time()
    while time.time() - t0 < secs:
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        n += 20
    print(f"load {cls}: {n} launches", flush=True)
