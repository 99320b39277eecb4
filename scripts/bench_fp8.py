"""fp8 (e4m3, block-scaled MFMA) vs bf16 3x3 convolutions at the SD1.5 shapes (B = 6): graph-captured timings.
usage: python scripts/bench_fp8.py"""
import sys, torch
sys.path.insert(0, '.')
from gaussctrl_amd.sd import ops
from gaussctrl_amd.sd.weights import conv3x3_weight, conv3x3_weight_fp8
from scripts.bench_kernels import timeit
DEV = 'cuda:0'; dt = torch.bfloat16; B = 6
rnd = lambda *s, scale=1.0: (torch.randn(*s, device=DEV) * scale).to(dt)
print("--- conv3x3 B=6: bf16 | fp8 (MX-scaled 16x16x128) | GroupNorm+SiLU -> bf16 (3 kernels) | -> e4m3 (stats + quantising apply)")
for (H, Cin, Cout) in [(64, 320, 320), (64, 640, 320), (64, 960, 320), (32, 640, 640), (32, 1280, 640), (16, 1280, 1280), (16, 2560, 1280), (8, 1280, 1280)]:
    x = rnd(B, H, H, Cin); w32 = torch.randn(Cout, Cin, 3, 3, device=DEV) * (9 * Cin) ** -0.5
    w = conv3x3_weight(w32, dt); b = torch.randn(Cout, device=DEV)
    w8, wsc = conv3x3_weight_fp8(w32)
    Cp = ops.pad128(Cin)
    x8 = torch.zeros(B, H, H, Cp, dtype=torch.uint8, device=DEV)
    x8[..., :Cin] = x.float().to(torch.float8_e4m3fn).view(torch.uint8)
    gam = torch.ones(Cin, device=DEV); bet = torch.zeros(Cin, device=DEV)
    t0 = timeit(lambda: ops.conv3x3(x, w, b)); t1 = timeit(lambda: ops.conv3x3_fp8(x8, w8, wsc, dt, b))
    t2 = timeit(lambda: ops.groupnorm(x, gam, bet, 32, 1e-5, True)); t3 = timeit(lambda: ops.groupnorm_fp8(x, gam, bet, 32, 1e-5, True))
    fl = 2.0 * B * H * H * Cout * 9 * Cin
    print(f"  {H:3d}^2 {Cin:5d}->{Cout:5d}: bf16 {t0:7.1f} us ({fl / t0 / 1e6:6.0f} TF/s)  fp8 {t1:7.1f} us ({fl / t1 / 1e6:6.0f} TF/s useful, x{t0 / t1:.2f})   GN bf16 {t2:6.1f} us  GN->e4m3 {t3:6.1f} us")
