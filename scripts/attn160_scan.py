"""k_attn (online-softmax kernel) at D = 160: time vs number of 64-key tiles, one K/V set, 192 workgroups.  python scripts/attn160_scan.py"""
import torch
from gaussctrl_amd.sd import ops
dev, dt = "cuda:0", torch.bfloat16
B, L, C, heads = 6, 256, 1280, 8
q = torch.randn(B, L, C, device=dev).to(dt)
for Lk in (64, 128, 256, 512, 1024, 2048):
    k = torch.randn(B, Lk, C, device=dev).to(dt); vt = torch.randn(B, C, Lk, device=dev).to(dt)
    fn = lambda: ops.attention(q, k, vt, heads, [(-1, 1.0)], 3, Lk=Lk, q_prescaled=True)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50): fn()
    e.record(); torch.cuda.synchronize()
    print(f"Lk = {Lk:5d} ({Lk // 64:3d} tiles): {s.elapsed_time(e) / 50 * 1e3:7.1f} us")
