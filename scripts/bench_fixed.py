"""Fixed (K-independent) cost of a GEMM launch: time vs K at fixed M, N."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussctrl_amd.sd import ops
from scripts.bench_kernels import timeit
dt = torch.bfloat16; DEV = "cuda:0"
for (M, N) in [(24576, 320), (24576, 640), (6144, 640), (1536, 1280)]:
    row = []
    for K in (64, 128, 320, 640, 1280):
        x = torch.randn(M, K, device=DEV).to(dt); w = (torch.randn(N, K, device=DEV) * K ** -0.5).to(dt); b = torch.randn(N, device=DEV)
        r = torch.randn(M, N, device=DEV).to(dt)
        row.append((K, timeit(lambda: ops.linear(x, w, b)), timeit(lambda: ops.linear(x, w, b, residual=r))))
    print(f"M={M} N={N}: " + "  ".join(f"K={k}: {a:.1f} / +res {c:.1f} us" for k, a, c in row))
