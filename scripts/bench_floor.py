import sys, torch
sys.path.insert(0, '.')
from gaussctrl_amd.sd import ops
import scripts.bench_kernels as bk
dt = torch.bfloat16; DEV = 'cuda:0'
for (M, K, N) in [(128, 64, 128), (128, 64, 160), (128, 320, 160), (128, 640, 160), (128, 1280, 160), (128, 2560, 160), (128, 64, 1280), (24576, 64, 320)]:
    x = (torch.randn(M, K, device=DEV)).to(dt); w = torch.randn(N, K, device=DEV).to(dt)
    us = bk.timeit(lambda: ops.linear(x, w))
    print(f"M={M} K={K} N={N}: {us:.1f} us")
x = torch.randn(1536, 1280, device=DEV).to(dt); g = torch.randn(1280, device=DEV); b = torch.randn(1280, device=DEV)
print("layernorm small:", bk.timeit(lambda: ops.layernorm(x, g, b)))
a = torch.randn(64, device=DEV).to(dt)
print("axpby tiny:", bk.timeit(lambda: ops.axpby(a)))
