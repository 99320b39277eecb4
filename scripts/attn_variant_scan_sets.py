"""Attention kernel choices at the shapes of a 4-chunk launch set (24 CFG frames, cached reference bank): auto vs the forced alternatives.  python scripts/attn_variant_scan_sets.py"""
import sys
sys.path.insert(0, '.')
import torch
from gaussctrl_amd.sd import ops
from scripts.bench_kernels import timeit
DEV = 'cuda:0'
dt = torch.bfloat16
f = 12
B = 2 * f
g = torch.Generator(device=DEV).manual_seed(0)
for (L, heads, D, variants) in ((1024, 8, 80, (0, 1)), (256, 8, 160, (0, 128, 1)), (64, 8, 160, (0, 128, 1)), (4096, 8, 40, (0, 16))):
    C = heads * D
    Lp = (L + 63) // 64 * 64
    qk = (torch.randn(B, L, 2 * C, device=DEV, generator=g) * 0.5).to(dt)
    q, k = qk[..., :C], qk[..., C:]
    vt = torch.randn(B, C, Lp, device=DEV, generator=g).to(dt)
    kr = torch.randn(8, L, 2 * C, device=DEV, generator=g).to(dt)[..., C:]
    vtr = torch.randn(8, C, Lp, device=DEV, generator=g).to(dt)
    sets = [(-1, 0.6)] + [(r, 0.1) for r in range(4)]
    row = []
    for v in variants:
        ops.KERNEL_VARIANT["attn"] = v
        try:
            us = timeit(lambda: ops.attention(q, k, vt, heads, sets, f, Lk=L, kref=kr, vtref=vtr, ref_fph=4, q_prescaled=True))
            row.append(f"variant {v:3d}: {us:8.1f} us {4.0 * B * L * L * C * 5 / us / 1e6:6.0f} TF/s")
        except Exception as e:
            row.append(f"variant {v:3d}: {type(e).__name__}")
    ops.KERNEL_VARIANT["attn"] = 0
    print(f"L={L:5d} heads x D = {heads} x {D:3d}, B = {B}: " + " | ".join(row))
