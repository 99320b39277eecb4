"""Is render_reverse (eval render -> VAE encode -> DDIM inversion) bit-reproducible when two processes share the GPU?  The six views of
tests/test_dist_gpu.py again and again; every hash must equal the first pass.  Run two copies at once.
python scripts/reverse_repro_stress.py [iters] (GC_* switches via ops.options_from_env)"""
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
h = lambda t: hashlib.md5(t.detach().float().cpu().contiguous().numpy().tobytes()).hexdigest()[:8]
views = [int(v) for v in os.environ.get("GC_STRESS_VIEWS", "1,3,5").split(",")]
ref, bad = None, 0
for it in range(iters):
    for t in td:
        for k in ("z_0_image", "unedited_image", "depth_image"):
            t.pop(k, None)
    pipe.render_reverse(views)
    cur = {i: (h(td[i]["unedited_image"]), h(td[i]["depth_image"]), h(td[i]["z_0_image"])) for i in views}
    if ref is None:
        ref = cur
        print("reference", ref, flush=True)
    elif cur != ref:
        bad += 1
        print(f"iteration {it}: " + " ".join(f"view {i}: rgb {'=' if cur[i][0] == ref[i][0] else 'X'} depth {'=' if cur[i][1] == ref[i][1] else 'X'} z0 {'=' if cur[i][2] == ref[i][2] else 'X'}" for i in views), flush=True)
print(f"{bad} deviating passes of {iters}", flush=True)
