"""fused transformer head (csrc/dn_thead.hip) against the four per-op launches: timing at B = 6, 4096 tokens. python scripts/thead_check.py [f16]"""
import os
import sys
import torch
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, "tests"))
from test_ttail_gpu import _sd, P, T, C, H
from gaussctrl_amd.sd import ops, weights

dt = torch.float16 if "f16" in sys.argv else torch.bfloat16
dev = "cuda:0"
sd = _sd()
sd[P + ".norm.weight"] = torch.ones(C); sd[P + ".norm.bias"] = torch.zeros(C)
w = weights.prepare(sd, dt, dev, heads=H)
B, HW = 6, 4096
x = torch.randn(B, HW, C).to(dt).to(dev)


def per_op():
    xn = ops.groupnorm(x, w[P + ".norm.weight"], w[P + ".norm.bias"], 32, 1e-6, False)
    h = ops.linear(xn, w[P + ".proj_in.weight"], w[P + ".proj_in.bias"])
    n1 = ops.layernorm(h, w[T + ".norm1.weight"], w[T + ".norm1.bias"])
    vt = torch.empty(B, C, HW, dtype=dt, device=dev)
    return ops.linear(n1, w[T + ".attn1.to_qkv.weight"], None, rows_per_batch=HW, out_t=vt, ldt=HW, t_batch_stride=C * HW, t_col0=2 * C, out_cols=2 * C)


def fused():
    coef = ops.groupnorm_coef(x, w[P + ".norm.weight"], w[P + ".norm.bias"], 32, 1e-6)
    return ops.transformer_head(x, coef, w[P + ".head.w"], w[P + ".head.params"])


coef = ops.groupnorm_coef(x, w[P + ".norm.weight"], w[P + ".norm.bias"], 32, 1e-6)
for name, fn in (("per-op (GroupNorm 3 launches + proj_in + LayerNorm + QKV)", per_op), ("fused (GroupNorm statistics 2 launches + head)", fused),
                 ("head kernel alone", lambda: ops.transformer_head(x, coef, w[P + ".head.w"], w[P + ".head.params"]))):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): fn()
    b.record(); torch.cuda.synchronize()
    print(f"{name}: {a.elapsed_time(b) / 20 * 1e3:.1f} us")

import ctypes
from gaussctrl_amd import _lib
lib = _lib.lib()
if hasattr(lib, "gc_dn_transformer_head_stamps"):
    ops.transformer_head(x, coef, w[P + ".head.w"], w[P + ".head.params"]); torch.cuda.synchronize()
    buf = (ctypes.c_uint64 * 16)()
    lib.gc_dn_transformer_head_stamps(buf)
    st = list(buf)
    names = ["wait tables", "GroupNorm apply", "proj_in GEMM", "store h", "LayerNorm1", "Q GEMM", "store q", "K GEMM", "store k", "V GEMM", "last V^T stores"]
    for i, n in enumerate(names):
        print(f"{n:20s} {st[i + 1] - st[i]:8d} cycles")
