"""The fused projection kernel as a pure function: same parameters, same camera, called again and again between DDIM inversions
(scripts/reverse_repro_stress.py found its outputs to differ now and then when two processes share the GPU).  Every output of every call is
compared with the first call's ON THE GPU; for a deviating call: which outputs, which Gaussians, which lane (i % 64) of the wavefront.
Run two copies at once.  python scripts/project_repro_stress.py [iters] [calls per iteration]"""
import collections, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gaussctrl_amd.sd import ops
ops.configure(ops.options_from_env())
from gaussctrl_amd import _lib as L, gsplat_ops as G
from gaussctrl_amd.camera import camera_to_gsplat
import test_dist_gpu as T
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 8
pipe, model = T._build(1, 0, -1)
td = pipe.datamanager.train_data
cams = pipe.datamanager.cameras
views = [int(v) for v in os.environ.get("GC_STRESS_VIEWS", "1,3,5").split(",")]
dev = model.means.device
N = model.means.shape[0]
lib = L.lib()
NAMES = ("xys", "depths", "radii", "conics", "num_tiles_hit", "rgbs", "opac", "boxes")


def project(i, tight=True):
    c = cams[i:i + 1]
    cam = camera_to_gsplat(c.camera_to_worlds[0].detach().cpu().numpy(), float(c.fx.reshape(-1)[0]), float(c.fy.reshape(-1)[0]),
                           float(c.cx.reshape(-1)[0]), float(c.cy.reshape(-1)[0]), T.W, T.H)
    m, ls, q = (G._c(t) for t in (model.means, model.scales, model.quats))
    op, dc, rest = G._c(model.opacities).reshape(-1), G._c(model.features_dc), G._c(model.features_rest)
    V, P, O = L.host_floats(cam["viewmat"]), L.host_floats(cam["fullproj"]), L.host_floats(cam["origin"])
    xys = torch.empty(N, 2, device=dev); depths = torch.empty(N, device=dev)
    radii = torch.empty(N, dtype=torch.int32, device=dev); conics = torch.empty(N, 3, device=dev)
    nth = torch.empty(N, dtype=torch.int32, device=dev); rgbs = torch.empty(N, 3, device=dev); opac = torch.empty(N, device=dev)
    boxes = torch.empty(N, dtype=torch.int32, device=dev)
    tb = ((T.W + 15) // 16, (T.H + 15) // 16)
    L.check(lib.gc_project_sh_fwd_boxes(
        L.i64(N), L.ptr(m), L.ptr(ls), L.ptr(q), L.ptr(op), L.ptr(dc), L.ptr(rest), L.i32(3), L.i32(3), V, P, O,
        L.f32(cam["fx"]), L.f32(cam["fy"]), L.f32(cam["cx"]), L.f32(cam["cy"]), L.i32(T.H), L.i32(T.W), L.i32(tb[0]), L.i32(tb[1]),
        L.f32(0.01), L.ptr(xys), L.ptr(depths), L.ptr(radii), L.ptr(conics), L.ptr(nth), L.ptr(rgbs), L.ptr(opac), L.ptr(boxes),
        L.stream_ptr()), "gc_project_sh_fwd_boxes")
    return dict(zip(NAMES, (xys, depths, radii, conics, nth, rgbs, opac, boxes)))


ref = {i: project(i) for i in views}
torch.cuda.synchronize()
lanes = collections.Counter(); which = collections.Counter(); first_call = collections.Counter()
bad = total = 0
for it in range(iters):
    for t in td:
        for k in ("z_0_image", "unedited_image", "depth_image"):
            t.pop(k, None)
    if os.environ.get("GC_STRESS_NOINVERT", "0") == "0":
        pipe.render_reverse(views)
    for c in range(calls):
        for i in views:
            out = project(i)
            total += 1
            diff = {n: (out[n] != ref[i][n]).reshape(N, -1).any(1) for n in NAMES}
            anyd = torch.stack(list(diff.values())).any(0)
            if bool(anyd.any()):
                bad += 1
                idx = anyd.nonzero().flatten().tolist()
                names = [n for n in NAMES if bool(diff[n].any())]
                first_call[c] += 1
                for j in idx:
                    lanes[j % 64] += 1
                for n in names:
                    which[n] += 1
                print(f"iteration {it} call {c} view {i}: {len(idx)} Gaussians differ {idx[:12]} lanes {[j % 64 for j in idx[:12]]} in {names}", flush=True)
print(f"{bad} deviating calls of {total}; by call index after the inversion {dict(first_call)}; by output {dict(which)}; by lane {dict(sorted(lanes.items()))}", flush=True)
