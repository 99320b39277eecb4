"""Seeded sweep over irregular GEMM / conv / attention shapes: the dispatch in dn_gemm.hip (4-wave / 8-wave, MT, split-K, MODE 0..4)
and dn_attn.hip (static-offset vs online softmax, ragged tiles, 1..5 K/V sets) must give the plain PyTorch answer for every grid."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.test_denoise_kernels_gpu import DEV, _close, _rand, _ref_attn

pytestmark = pytest.mark.gpu


def test_linear_shape_sweep():
    from gaussctrl_amd.sd import ops
    rng = np.random.default_rng(7)
    dt = torch.bfloat16
    Ms = [1, 6, 77, 128, 200, 384, 1000, 1536, 3000, 6144, 12288 + 40, 24576]
    Ks = [64, 72, 320, 640, 768, 1280, 2560, 5120]
    Ns = [8, 40, 320, 328, 640, 960, 1280, 2560]
    for it in range(36):
        M, K, N = int(rng.choice(Ms)), int(rng.choice(Ks)), int(rng.choice(Ns))
        if M * N * K > 3e11:
            continue
        x = _rand((M, K), dt, 1.0, 100 + it); w = _rand((N, K), dt, K ** -0.5, 200 + it)
        b = torch.randn(N, device=DEV) if it % 3 else None
        r = _rand((M, N), dt, 1.0, 300 + it) if it % 2 else None
        ref = x.double() @ w.double().T + (0 if b is None else b.double()) + (0 if r is None else r.double())
        try:
            _close(ops.linear(x, w, b, residual=r), ref, dt, extra=2.0)
        except AssertionError as e:
            raise AssertionError(f"linear M={M} K={K} N={N}: {e}")


def test_conv_shape_sweep():
    from gaussctrl_amd.sd import ops
    from gaussctrl_amd.sd.weights import conv3x3_weight
    rng = np.random.default_rng(8)
    dt = torch.bfloat16
    for it in range(24):
        B = int(rng.choice([1, 2, 3, 6]))
        H = int(rng.choice([5, 8, 16, 24, 32, 64])); W = int(rng.choice([6, 8, 16, 32, 64]))
        Cin = int(rng.choice([8, 64, 96, 128, 320, 640])); Cout = int(rng.choice([32, 64, 160, 320, 640]))
        stride = int(rng.choice([1, 1, 1, 2])); ups = bool(stride == 1 and rng.random() < 0.25)
        if B * H * W * (4 if ups else 1) * Cin * Cout * 18 > 2.5e11:
            continue
        x = _rand((B, H, W, Cin), dt, 1.0, 400 + it); w = _rand((Cout, Cin, 3, 3), dt, (9 * Cin) ** -0.5, 500 + it)
        b = torch.randn(Cout, device=DEV)
        xin = x.double().permute(0, 3, 1, 2)
        if ups:
            xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
        ref = F.conv2d(xin, w.double(), b.double(), stride=stride, padding=1).permute(0, 2, 3, 1)
        try:
            _close(ops.conv3x3(x, conv3x3_weight(w, dt), b, stride=stride, upsample=ups), ref, dt, extra=2.0)
        except AssertionError as e:
            raise AssertionError(f"conv B={B} {H}x{W} {Cin}->{Cout} s{stride} ups{int(ups)}: {e}")


@pytest.mark.parametrize("D", [40, 80, 160])
def test_attention_shape_sweep(D):
    from gaussctrl_amd.sd import ops
    rng = np.random.default_rng(9 + D)
    dt = torch.bfloat16
    for it in range(10):
        f = int(rng.choice([4, 5, 7])); heads = int(rng.choice([1, 2, 3])); L = int(rng.choice([17, 64, 100, 256, 320, 777]))
        nref = int(rng.choice([0, 1, 4])); coeff = float(rng.choice([0.0, 0.6])) if nref else 1.0
        B, Cc = 2 * f, heads * D
        q = _rand((B, L, Cc), dt, 1.0, 600 + it); k = _rand((B, L, Cc), dt, 1.0, 700 + it); v = _rand((B, L, Cc), dt, 1.0, 800 + it)
        Lp = (L + 7) // 8 * 8
        vt = torch.zeros(B, Cc, Lp, dtype=dt, device=DEV); vt[:, :, :L] = v.transpose(1, 2)
        scale = D ** -0.5
        sets = ([(-1, coeff)] if coeff != 0 else []) + [(r, (1 - coeff) / max(nref, 1)) for r in range(nref)]
        ref = coeff * _ref_attn(q, k, v, heads, scale) if coeff != 0 else 0
        for r in range(nref):
            idx = torch.arange(B, device=DEV) // f * f + r
            ref = ref + (1 - coeff) / nref * _ref_attn(q, k[idx], v[idx], heads, scale)
        try:
            _close(ops.attention(q, k, vt, heads, sets, f, Lk=L), ref, dt, extra=12.0)     # P is rounded to bf16 before P V; 4 equal-weight sets
        except AssertionError as e:
            raise AssertionError(f"attention D={D} f={f} heads={heads} L={L} sets={sets}: {e}")
