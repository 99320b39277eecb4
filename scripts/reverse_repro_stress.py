"""Is render_reverse (eval render -> VAE encode -> DDIM inversion) bit-reproducible when two processes share the GPU?  Views of
tests/test_dist_gpu.py again and again; every hash (incl. the rasterizer's intermediates) must equal the first pass.  Run two copies at once.
python scripts/reverse_repro_stress.py [iters] (GC_* switches via ops.options_from_env; GC_STRESS_VIEWS=1,3,5)"""
import hashlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gaussctrl_amd.sd import ops
ops.configure(ops.options_from_env())
import test_dist_gpu as T
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
pipe, model = T._build(1, 0, -1)
td = pipe.datamanager.train_data
h = lambda t: hashlib.md5(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:8]
views = [int(v) for v in os.environ.get("GC_STRESS_VIEWS", "1,3,5").split(",")]
FIELDS = ("xys", "depths", "radii", "num_tiles_hit", "gaussian_ids_sorted", "tile_bins", "final_index")
ref, bad = None, 0
KEEP = {}
QREF = {n: getattr(model, n).detach().clone() for n in ("quats", "scales", "features_rest")} if os.environ.get("GC_STRESS_ECHO", "1") == "1" else None
cams = pipe.datamanager.cameras
for it in range(iters):
    cur = {}
    for i in views:
        if QREF is not None:                      # does a plain torch copy kernel see the parameters as they are?
            for name, r in QREF.items():
                if not torch.equal(getattr(model, name).detach().clone(), r):
                    print(f"   torch clone of {name} differs from its first copy (iteration {it}, before view {i})", flush=True)
        o = model.get_outputs_for_camera(cams[i:i + 1] if hasattr(cams, "__getitem__") else cams[i])
        a = model._aux
        cur[i] = {f: h(getattr(a, f)) for f in FIELDS}
        cur[i]["M"] = int(a.M); cur[i]["rgb"] = h(o["rgb"]); cur[i]["depth"] = h(o["depth"])
        cur[i]["inputs"] = h(torch.cat([p.detach().reshape(-1).float() for p in (model.means, model.scales, model.quats, model.opacities, model.features_dc, model.features_rest)]))
        tens = {f: getattr(a, f).detach().cpu().clone() for f in ("xys", "depths", "radii", "num_tiles_hit", "tile_boxes")}
        if i not in KEEP:
            KEEP[i] = tens
        elif ref is not None and cur[i] != ref[i]:
            for f, t in tens.items():
                r = KEEP[i][f]
                d = (t != r).reshape(t.shape[0], -1).any(1).nonzero().flatten()
                if d.numel():
                    j = d[:4].tolist()
                    print(f"   view {i} {f}: {d.numel()} Gaussians differ; first {j}: now {t[j].tolist()} was {r[j].tolist()}", flush=True)
            o2 = model.get_outputs_for_camera(cams[i:i + 1] if hasattr(cams, "__getitem__") else cams[i])
            print(f"   view {i} rendered again at once: radii {'same as reference' if h(model._aux.radii) == ref[i]['radii'] else 'DIFFERENT'}, rgb {'same' if h(o2['rgb']) == ref[i]['rgb'] else 'DIFFERENT'}", flush=True)
    for t in td:
        for k in ("z_0_image", "unedited_image", "depth_image"):
            t.pop(k, None)
    if os.environ.get("GC_STRESS_NOINVERT", "0") == "0":
        pipe.render_reverse(views)                      # (renders again, then VAE encode + DDIM inversion: the denoise-kernel context)
    if ref is None:
        ref = cur
        print("reference M", {i: cur[i]["M"] for i in views}, flush=True)
    elif cur != ref:
        bad += 1
        print(f"iteration {it}: " + " | ".join(f"view {i} differs in {[k for k in cur[i] if cur[i][k] != ref[i][k]]}" for i in views if cur[i] != ref[i]), flush=True)
print(f"{bad} deviating passes of {iters}", flush=True)
