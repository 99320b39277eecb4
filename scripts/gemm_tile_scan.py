"""Asymptotic efficiency of the 8-wave GEMM per workgroup tile height (MT = 2 / 3 / 4 -> 128 / 192 / 256 rows): 3x3 conv and linear at
growing batch, so that grid quantisation stops mattering.  If the taller tile is clearly faster per flop, operand (LDS) traffic per
MFMA is what bounds the kernel.   python scripts/gemm_tile_scan.py"""
import sys
sys.path.insert(0, '.')
import torch
from gaussctrl_amd.sd import ops
from gaussctrl_amd.sd.weights import conv3x3_weight
from scripts.bench_kernels import timeit
DEV = 'cuda:0'
dt = torch.bfloat16
rnd = lambda *s, scale=1.0: (torch.randn(*s, device=DEV) * scale).to(dt)
for (H, Cin, Cout) in ((64, 320, 320), (32, 640, 640)):
    for B in (6, 16, 32):
        x = rnd(B, H, H, Cin); w = conv3x3_weight(rnd(Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5), dt); b = torch.randn(Cout, device=DEV)
        row = []
        for mt in (0, 2, 3, 4):
            ops.KERNEL_VARIANT["gemm"] = mt
            us = timeit(lambda: ops.conv3x3(x, w, b))
            row.append(f"MT={mt or 'auto'}: {us:7.1f} us {2.0 * B * H * H * Cout * 9 * Cin / us / 1e6:6.0f} TF/s")
        ops.KERNEL_VARIANT["gemm"] = 0
        print(f"conv {H}x{H} {Cin}->{Cout} B={B:2d}  " + " | ".join(row))
for (L, K, N) in ((4096, 320, 320), (4096, 1280, 320), (1024, 640, 640)):
    for B in (6, 32):
        x = rnd(B, L, K); w = rnd(N, K, scale=K ** -0.5); b = torch.randn(N, device=DEV)
        row = []
        for mt in (0, 2, 3, 4):
            ops.KERNEL_VARIANT["gemm"] = mt
            us = timeit(lambda: ops.linear(x, w, b))
            row.append(f"MT={mt or 'auto'}: {us:7.1f} us {2.0 * B * L * N * K / us / 1e6:6.0f} TF/s")
        ops.KERNEL_VARIANT["gemm"] = 0
        print(f"linear M={B * L} K={K} N={N}  " + " | ".join(row))
