"""GroupNorm statistics from the producer's epilogue (ops.ChanParts) vs the stand-alone three-kernel GroupNorm: per-launch times at the
benchmark's shapes (B = 6).  python scripts/gn_parts_bench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussctrl_amd.sd import ops
from gaussctrl_amd.sd.weights import conv3x3_weight

dev, dt = "cuda:0", torch.bfloat16


def t(fn, n=100):
    for _ in range(5):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1000 / n


print(f"{'shape':28s} {'conv':>8s} {'conv+parts':>10s} | {'gn 3-kernel':>11s} {'gn(parts)':>9s} | {'coef 2-kernel':>13s} {'coef(parts)':>11s}   [us]")
for B, H, C in ((6, 64, 320), (6, 32, 640), (6, 16, 1280), (6, 32, 320), (6, 64, 640)):
    x = torch.randn(B, H, H, C, device=dev).to(dt)
    w = conv3x3_weight((torch.randn(C, C, 3, 3, device=dev) * (9 * C) ** -0.5).to(dt), dt)
    b = torch.randn(C, device=dev)
    gamma, beta = torch.randn(C, device=dev), torch.randn(C, device=dev)
    out, parts = ops.conv3x3(x, w, b, chan_parts=True)
    x3 = out.view(B, H * H, C)
    r = [t(lambda: ops.conv3x3(x, w, b)), t(lambda: ops.conv3x3(x, w, b, chan_parts=True)),
         t(lambda: ops.groupnorm(out, gamma, beta, 32, 1e-5, True)), t(lambda: ops.groupnorm(out, gamma, beta, 32, 1e-5, True, parts=parts)) if parts else float("nan"),
         t(lambda: ops.groupnorm_coef(x3, gamma, beta, 32, 1e-5)), t(lambda: ops.groupnorm_coef(x3, gamma, beta, 32, 1e-5, parts=parts)) if parts else float("nan")]
    print(f"B{B} {H}x{H} C{C} rows/slab {parts.rows if parts else 0:4d} x{parts.nslab if parts else 0:2d}  {r[0]:8.1f} {r[1]:10.1f} | {r[2]:11.1f} {r[3]:9.1f} | {r[4]:13.1f} {r[5]:11.1f}")
for B, H, C1, C2 in ((6, 64, 320, 320), (6, 32, 640, 640), (6, 16, 1280, 1280)):
    a = torch.randn(B, H, H, C1, device=dev).to(dt); bb = torch.randn(B, H, H, C2, device=dev).to(dt); cc = torch.randn(B, H, H, C2, device=dev).to(dt)
    print(f"concat B{B} {H}x{H} {C1}+{C2}: plain {t(lambda: ops.concat_add(a, bb, cc)):.1f}  with parts {t(lambda: ops.concat_add(a, bb, cc, chan_parts=True)):.1f}")
