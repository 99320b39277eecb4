"""Cost of the fused-normalisation epilogue options on the GEMM shapes that carry them (graph-captured timings, bf16, B = 6).
usage: python scripts/bench_epilogue.py"""
import sys, torch
sys.path.insert(0, '.')
from gaussctrl_amd.sd import ops
from gaussctrl_amd.sd.weights import conv3x3_weight, geglu_permute
from scripts.bench_kernels import timeit
DEV = 'cuda:0'; dt = torch.bfloat16; B = 6
rnd = lambda *s, scale=1.0: (torch.randn(*s, device=DEV) * scale).to(dt)
print("--- conv3x3: plain | +group stats")
for (H, Cin, Cout) in [(64, 320, 320), (64, 960, 320), (32, 640, 640), (16, 1280, 1280), (8, 1280, 1280)]:
    x = rnd(B, H, H, Cin); w = conv3x3_weight(rnd(Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5), dt); b = torch.randn(Cout, device=DEV)
    gs = torch.zeros(B, 32, 2, device=DEV)
    t0 = timeit(lambda: ops.conv3x3(x, w, b)); t1 = timeit(lambda: ops.conv3x3(x, w, b, group_stats=gs))
    t2 = timeit(lambda: ops.groupnorm(x, torch.ones(Cin, device=DEV), torch.zeros(Cin, device=DEV), 32, 1e-5, True))
    t3 = timeit(lambda: ops.groupnorm_apply(x, gs, torch.ones(Cin, device=DEV), torch.zeros(Cin, device=DEV), 32, 1e-5, True))
    print(f"  {H:3d}^2 {Cin:5d}->{Cout:5d}: conv {t0:7.1f} us  +stats {t1:7.1f} us ({t1 - t0:+.1f}) | groupnorm(input) 3-kernel {t2:6.1f} us, apply-from-stats {t3:6.1f} us")
print("--- linear M = B*L: plain | +row stats | +group stats | consumer with LN fold (vs layernorm + plain)")
for (L, K, N) in [(4096, 320, 320), (1024, 640, 640), (256, 1280, 1280), (64, 1280, 1280)]:
    x = rnd(B, L, K); w = rnd(N, K, scale=K ** -0.5); b = torch.randn(N, device=DEV); r = rnd(B, L, N)
    gs = torch.zeros(B, 32, 2, device=DEV); rs = ops.RowStats()
    t0 = timeit(lambda: ops.linear(x, w, b, residual=r))
    t1 = timeit(lambda: ops.linear(x, w, b, residual=r, row_stats=rs))
    t2 = timeit(lambda: ops.linear(x, w, b, residual=r, rows_per_batch=L, group_stats=gs))
    print(f"  M={B * L:6d} K={K:5d} N={N:5d}: plain {t0:6.1f} us  +row stats {t1:6.1f} ({t1 - t0:+.1f})  +group stats {t2:6.1f} ({t2 - t0:+.1f})   slots={rs.slots}")
    for (N2, geglu) in [(3 * K, False), (8 * K, True)]:
        w2 = rnd(N2, K, scale=K ** -0.5); b2 = torch.randn(N2, device=DEV)
        if geglu: w2, b2 = geglu_permute(w2, b2)
        cs = w2.float().sum(1).contiguous()
        xo = ops.linear(x, w, b, residual=r, row_stats=rs)
        g1 = torch.ones(N, device=DEV); b1 = torch.zeros(N, device=DEV)
        ta = timeit(lambda: ops.linear(xo, w2, b2, geglu=geglu))
        tb = timeit(lambda: ops.linear(xo, w2, b2, geglu=geglu, ln=(rs, cs, 1e-5)))
        tc = timeit(lambda: ops.layernorm(xo, g1, b1))
        print(f"       consumer N={N2:5d} geglu{int(geglu)}: plain {ta:6.1f} us  LN-folded {tb:6.1f} ({tb - ta:+.1f})   [layernorm alone {tc:5.1f} us]")
import os
if os.environ.get("GC_GEMM_DBG"):
    print("GC_GEMM_DBG =", os.environ["GC_GEMM_DBG"], "(1: no global group atomics, 2: no LDS atomics, 4: no DPP reduce)")
