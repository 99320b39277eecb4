import sys, torch
sys.path.insert(0, '.')
from gaussctrl_amd.sd import ops
DEV='cuda:0'
for dt in (torch.bfloat16, torch.float16):
    torch.manual_seed(0)
    B, L, H, D = 2, 64, 1, 40
    C = H * D
    q = torch.randn(B, L, C, device=DEV).to(dt); k = torch.randn(B, L, C, device=DEV).to(dt); v = torch.randn(B, L, C, device=DEV).to(dt)
    vt = v.transpose(1, 2).contiguous()
    o = ops.attention(q, k, vt, H, [(-1, 1.0)], 1, Lk=L).float()
    ref = torch.softmax((q.float() @ k.float().transpose(1, 2)) * D ** -0.5, -1) @ v.float()
    r = (o / ref)
    print(dt, "max err", float((o - ref).abs().max()))
    print(" ratio per d (row 0):", [round(float(x), 3) for x in r[0, 0]])
    print(" ratio per q (d 0):", [round(float(x), 3) for x in r[0, :20, 0]])
