"""Achieved HBM rate of the one-launch GroupNorm (+ SiLU) apply kernel (k_gn_apply_parts) at the shapes of a 4-chunk launch set (24 CFG frames):
bytes = read x + write y.   python scripts/gn_apply_bw.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussctrl_amd.sd import ops
from gaussctrl_amd.sd.weights import conv3x3_weight

dev, dt = "cuda:0", torch.bfloat16


def t(fn, n=50):
    for _ in range(5):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1000 / n


for B, H, C in ((24, 64, 320), (24, 64, 640), (24, 64, 960), (24, 32, 640), (24, 32, 1280), (24, 32, 1920), (24, 16, 1280), (24, 16, 2560), (24, 8, 1280), (6, 64, 320)):
    Cc = min(C, 640)
    x = torch.randn(B, H, H, Cc, device=dev).to(dt)
    w = conv3x3_weight((torch.randn(C, Cc, 3, 3, device=dev) * (9 * Cc) ** -0.5).to(dt), dt)
    b = torch.randn(C, device=dev)
    gamma, beta = torch.randn(C, device=dev), torch.randn(C, device=dev)
    out, parts = ops.conv3x3(x, w, b, chan_parts=True)
    if parts is None:
        print(f"B{B} {H}x{H} C{C}: no parts"); continue
    us = t(lambda: ops.groupnorm(out, gamma, beta, 32, 1e-5, True, parts=parts))
    us3 = t(lambda: ops.groupnorm(out, gamma, beta, 32, 1e-5, True))
    by = 2.0 * out.numel() * 2
    print(f"B{B} {H}x{H} C{C:5d} slabs {parts.rows:4d} x{parts.nslab:3d}: gn(parts) {us:7.1f} us {by / us / 1e6:6.2f} TB/s | 3-kernel {us3:7.1f} us | {by / 1e6:7.1f} MB")
