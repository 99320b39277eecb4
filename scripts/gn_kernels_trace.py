"""Per-kernel GPU time of the GroupNorm paths at one shape, to be run under rocprofv3 --kernel-trace --stats (host-side timing of
4-20 us kernels measures Python): python scripts/gn_kernels_trace.py B H C"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussctrl_amd.sd import ops
from gaussctrl_amd.sd.weights import conv3x3_weight
ops.configure(ops.options_from_env())
B, H, C = (int(v) for v in sys.argv[1:4])
dev, dt = "cuda:0", torch.bfloat16
x = torch.randn(B, H, H, C, device=dev).to(dt)
w = conv3x3_weight((torch.randn(C, C, 3, 3, device=dev) * (9 * C) ** -0.5).to(dt), dt)
b = torch.randn(C, device=dev); gamma, beta = torch.randn(C, device=dev), torch.randn(C, device=dev)
a2 = torch.randn(B, H, H, C, device=dev).to(dt)
for _ in range(30):
    out, parts = ops.conv3x3(x, w, b, chan_parts=True)
    ops.conv3x3(x, w, b)
    ops.groupnorm(out, gamma, beta, 32, 1e-5, True)
    if parts is not None:
        ops.groupnorm(out, gamma, beta, 32, 1e-5, True, parts=parts)
    cat, p2 = ops.concat_add(out, a2, a2, chan_parts=True)
    ops.concat_add(out, a2, a2)
    if p2 is not None:
        ops.groupnorm(cat, torch.cat([gamma, gamma]), torch.cat([beta, beta]), 32, 1e-5, True, parts=p2)
    ops.groupnorm(cat, torch.cat([gamma, gamma]), torch.cat([beta, beta]), 32, 1e-5, True)
torch.cuda.synchronize()
