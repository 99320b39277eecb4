"""Does the 8-wave GEMM pick the best tile height at the shapes of a 4-chunk launch set (24 CFG frames; round 6)?  auto vs forced MT 2 / 3 / 4 (a forced MT also
disables k-slices) and, for the k-sliced shapes, the forced slice tile heights (kernel_variant bits 24..26).   python scripts/gemm_mt_scan_sets.py"""
import sys
sys.path.insert(0, '.')
import torch
from gaussctrl_amd.sd import ops
from gaussctrl_amd.sd.weights import conv3x3_weight
from scripts.bench_kernels import timeit
DEV = 'cuda:0'
dt = torch.bfloat16
rnd = lambda *s, scale=1.0: (torch.randn(*s, device=DEV) * scale).to(dt)
B = 24
for (H, Cin, Cout) in ((64, 320, 320), (64, 640, 320), (64, 960, 320), (32, 320, 640), (32, 640, 640), (32, 1280, 640), (32, 1920, 640), (32, 1280, 1280),
                       (16, 640, 1280), (16, 1280, 1280), (16, 2560, 1280), (8, 1280, 1280), (8, 2560, 1280)):
    x = rnd(B, H, H, Cin); w = conv3x3_weight(rnd(Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5), dt); b = torch.randn(Cout, device=DEV)
    row = []
    for name, kv in (("auto", 0), ("MT2", 2), ("MT3", 3), ("MT4", 4), ("sliceMT2", 2 << 24), ("sliceMT3", 3 << 24), ("sliceMT4", 4 << 24)):
        ops.KERNEL_VARIANT["gemm"] = kv
        us = timeit(lambda: ops.conv3x3(x, w, b, chan_parts=True))
        row.append(f"{name} {us:7.1f}")
    ops.KERNEL_VARIANT["gemm"] = 0
    fl = 2.0 * B * H * H * Cout * 9 * Cin
    best = min(row, key=lambda r: float(r.split()[1]))
    print(f"conv {H:2d}x{H:2d} {Cin:4d}->{Cout:4d}  " + " | ".join(row) + f"   [auto {fl / float(row[0].split()[1]) / 1e6:5.0f} TF/s; best {best.split()[0]}]")
for (L, K, N, geglu) in ((1024, 640, 640, False), (1024, 640, 5120, True), (1024, 3200, 640, False), (256, 1280, 1280, False), (256, 1280, 10240, True), (256, 6400, 1280, False),
                         (4096, 320, 320, False), (64, 1280, 1280, False)):
    x = rnd(B, L, K); w = rnd(N, K, scale=K ** -0.5); b = torch.randn(N, device=DEV)
    row = []
    for name, kv in (("auto", 0), ("MT2", 2), ("MT3", 3), ("MT4", 4)):
        ops.KERNEL_VARIANT["gemm"] = kv
        us = timeit(lambda: ops.linear(x, w, b, geglu=geglu))
        row.append(f"{name} {us:7.1f}")
    ops.KERNEL_VARIANT["gemm"] = 0
    best = min(row, key=lambda r: float(r.split()[1]))
    print(f"linear M={B * L:6d} K={K:5d} N={N:5d} geglu={int(geglu)}  " + " | ".join(row) + f"   [auto {2.0 * B * L * N * K / float(row[0].split()[1]) / 1e6:5.0f} TF/s; best {best.split()[0]}]")
