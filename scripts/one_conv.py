import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussctrl_amd.sd import ops
from gaussctrl_amd.sd.weights import conv3x3_weight
dt = torch.bfloat16; DEV='cuda:0'; B=6
which = sys.argv[1] if len(sys.argv) > 1 else 'conv'
if which == 'conv':
    H, Cin, Cout = 64, 320, 320
    x = (torch.randn(B, H, H, Cin, device=DEV)).to(dt); w = conv3x3_weight((torch.randn(Cout, Cin, 3, 3, device=DEV) * (9*Cin) ** -0.5).to(dt), dt); b = torch.randn(Cout, device=DEV)
    for _ in range(5): ops.conv3x3(x, w, b)
elif which.startswith('lin'):
    M, K, N = {'lin': (24576, 320, 320), 'lin2': (6144, 640, 640), 'lin3': (24576, 320, 2560), 'lin4': (1536, 1280, 1280)}[which]
    x = torch.randn(M, K, device=DEV).to(dt); w = (torch.randn(N, K, device=DEV) * K ** -0.5).to(dt); b = torch.randn(N, device=DEV)
    for _ in range(5): ops.linear(x, w, b)
elif which.startswith('conv'):
    H, Cin, Cout = {'conv2': (32, 640, 640), 'conv3': (16, 1280, 1280), 'conv4': (64, 640, 320)}[which]
    x = (torch.randn(B, H, H, Cin, device=DEV)).to(dt); w = conv3x3_weight((torch.randn(Cout, Cin, 3, 3, device=DEV) * (9*Cin) ** -0.5).to(dt), dt); b = torch.randn(Cout, device=DEV)
    for _ in range(5): ops.conv3x3(x, w, b)
else:
    L, C, heads = 4096, 320, 8
    q = torch.randn(B, L, C, device=DEV).to(dt); k = torch.randn(B, L, C, device=DEV).to(dt); vt = torch.randn(B, C, L, device=DEV).to(dt)
    kr = torch.randn(8, L, C, device=DEV).to(dt); vtr = torch.randn(8, C, L, device=DEV).to(dt)
    for _ in range(3): ops.attention(q, k, vt, heads, [(-1, 0.6)] + [(r, 0.1) for r in range(4)], B // 2, Lk=L, kref=kr, vtref=vtr, ref_fph=4, q_prescaled=True)
torch.cuda.synchronize()
