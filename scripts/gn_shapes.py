"""GroupNorm(+SiLU) launch times at the UNet's shapes (B = 6): python scripts/gn_shapes.py"""
import torch
from gaussctrl_amd.sd import ops
dev = "cuda:0"
for HW, Cs in ((4096, (320, 640, 960)), (1024, (320, 640, 960, 1280, 1920)), (256, (640, 1280, 1920, 2560)), (64, (1280, 2560))):
    for C in Cs:
        x = torch.randn(6, HW, C, device=dev).to(torch.bfloat16)
        g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
        for _ in range(3): ops.groupnorm(x, g, b, 32, 1e-5, True)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50): ops.groupnorm(x, g, b, 32, 1e-5, True)
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / 50 * 1e3
        mb = x.numel() * 2 / 1e6
        print(f"GroupNorm+SiLU [6, {HW:4d}, {C:4d}] {mb:6.1f} MB: {us:7.1f} us   ({3 * mb / us:5.2f} TB/s for read + read + write)")
