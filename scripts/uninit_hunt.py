"""Hunt for reads of memory nobody wrote: torch.empty / empty_like are patched to hand out NaN-filled buffers, then the small 1-rank
pipeline of tests/test_dist_gpu.py runs; every ops.* call is checked for NaN in its outputs (the first offender is printed).
python scripts/uninit_hunt.py [invariant]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
_empty, _empty_like = torch.empty, torch.empty_like


def _poison(t):
    if t.is_cuda and t.numel():
        if t.is_floating_point():
            t.fill_(float("nan"))
        elif t.dtype == torch.uint8:
            t.fill_(0xFF)
    return t


torch.empty = lambda *a, **k: _poison(_empty(*a, **k))
torch.empty_like = lambda *a, **k: _poison(_empty_like(*a, **k))
from gaussctrl_amd.sd import ops
if "invariant" in sys.argv:
    ops.configure(batch_invariant=True)
seen = set()


def wrap(name, fn):
    def f(*a, **k):
        out = fn(*a, **k)
        outs = out if isinstance(out, tuple) else (out,)
        for i, o in enumerate(outs):
            if isinstance(o, torch.Tensor) and o.is_floating_point() and o.numel() and name not in seen:
                bad = torch.isnan(o)
                if bool(bad.any()):
                    seen.add(name)
                    shapes = [tuple(x.shape) for x in a if isinstance(x, torch.Tensor)]
                    print(f"NaN in output {i} of ops.{name}: {int(bad.sum())} of {o.numel()} (shape {tuple(o.shape)}; inputs {shapes}; kwargs {sorted(k)})", flush=True)
        return out
    return f


for n in ("linear", "conv3x3", "groupnorm", "groupnorm_coef", "layernorm", "concat_add", "axpby", "attention", "transformer_tail", "transformer_head", "cast_f32",
          "depth_to_disparity", "mask_composite"):
    setattr(ops, n, wrap(n, getattr(ops, n)))
import test_dist_gpu as T
pipe, model = T._build(1, 0, -1)
imgs, losses, means = T._run(pipe, model)
print("images finite:", bool(torch.isfinite(imgs).all()), "nan count", int(torch.isnan(imgs).sum()))
