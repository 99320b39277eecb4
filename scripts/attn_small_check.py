"""launch times of the online-softmax attention kernel (attn_safe_body) at the UNet's D = 160 / D = 80-text shapes, B = 6 frames (2 halves x 3).
python scripts/attn_small_check.py"""
import torch
from gaussctrl_amd.sd import ops
dev, dt = "cuda:0", torch.bfloat16
def t(fn, n=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
B, f = 6, 3
for name, L, C, heads in (("level 2 (16x16)", 256, 1280, 8), ("mid (8x8)", 64, 1280, 8), ("level 1 (32x32) text", 1024, 640, 8)):
    q = torch.randn(B, L, C, device=dev).to(dt); k = torch.randn(B, L, C, device=dev).to(dt); vt = torch.randn(B, C, L, device=dev).to(dt)
    kr = torch.randn(8, L, C, device=dev).to(dt); vr = torch.randn(8, C, L, device=dev).to(dt)
    sets = [(-1, 0.6)] + [(r, 0.1) for r in range(4)]
    if "text" not in name:
        us = t(lambda: ops.attention(q, k, vt, heads, sets, f, Lk=L, kref=kr, vtref=vr, ref_fph=4, q_prescaled=True))
        fl = 4.0 * B * L * L * C * 5
        print(f"{name}: cross-view self-attention, 5 sets x {L} keys, D = {C // heads}: {us:6.1f} us  ({fl / us * 1e-6:6.1f} TF/s)")
    kt = torch.randn(2, 77, C, device=dev).to(dt); vtt = torch.zeros(2, C, 80, device=dev, dtype=dt); vtt[..., :77] = torch.randn(2, C, 77, device=dev).to(dt)
    us = t(lambda: ops.attention(q, kt, vtt, heads, [(-2, 1.0)], f, Lk=77, q_prescaled=True))
    print(f"{name}: text cross-attention, 77 keys, D = {C // heads}: {us:6.1f} us")
