"""CPU tests pinning the denoise oracle: the cross-view attention restatement against golden vectors the
REFERENCE's own utils.py produced (tests/golden/make_xview_golden.py), plus structural checks."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import sd15_torch as sd

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "xview_attn_*.npz")))


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_xview_attention_matches_reference_golden(path):
    z = np.load(path)
    f, L, H, D, Lt, Ct = [int(v) for v in z["meta"]]
    coeff = float(z["coeff"])
    for kind in ("self", "text"):
        w = {"a.to_q.weight": torch.tensor(z[f"{kind}_wq"]), "a.to_k.weight": torch.tensor(z[f"{kind}_wk"]),
             "a.to_v.weight": torch.tensor(z[f"{kind}_wv"]), "a.to_out.0.weight": torch.tensor(z[f"{kind}_wo"]),
             "a.to_out.0.bias": torch.tensor(z[f"{kind}_bo"])}
        x = torch.tensor(z[f"{kind}_x"])
        ctx = torch.tensor(z[f"{kind}_ctx"]) if kind == "text" else None
        y = sd.attention_layer(w, "a", x, ctx, H, "xview", coeff)
        np.testing.assert_allclose(y.numpy(), z[f"{kind}_y"], rtol=0, atol=2e-6)


BIG = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "xview_big_*.npz")))


@pytest.mark.parametrize("path", BIG, ids=[os.path.basename(p) for p in BIG])
def test_xview_attention_matches_reference_golden_production_geometry(path):
    """The oracle's cross-view attention (sdpa form: what the full-geometry fixtures are generated with) against the REFERENCE's
    output at SD1.5's own level-0 / level-1 geometry (L = 4096, D = 40 and L = 1024, D = 80; f = 5, CFG batch 10), on the
    strided token rows the fixture keeps."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from xview_big_inputs import big_inputs
    z = np.load(path)
    f, L, H, D, Lt, Ct, seed, stride = [int(v) for v in z["meta"]]
    inp = big_inputs(seed, f, L, H, D, Lt, Ct)
    keep, sd.ATTN_IMPL = sd.ATTN_IMPL, "sdpa"
    try:
        for kind in ("self", "text"):
            d = inp[kind]
            w = {"a.to_q.weight": d["wq"], "a.to_k.weight": d["wk"], "a.to_v.weight": d["wv"], "a.to_out.0.weight": d["wo"],
                 "a.to_out.0.bias": d["bo"]}
            with torch.no_grad():
                y = sd.attention_layer(w, "a", d["x"], d["ctx"], H, "xview", float(z["coeff"]))
            ref = z[f"{kind}_y_rows"]
            got = y[:, ::stride].numpy()
            assert abs(float(y.double().norm()) / float(z[f"{kind}_y_norm"]) - 1) < 1e-5
            np.testing.assert_allclose(got, ref, rtol=0, atol=1e-5 * max(1.0, float(np.abs(ref).max())))
    finally:
        sd.ATTN_IMPL = keep


def test_golden_present():
    assert len(GOLD) >= 4 and len(BIG) == 2


def test_tiny_unet_controlnet_shapes_and_ref_independence():
    """refs attend only to refs (utils.py:94-117): the first 4 frames' outputs do not depend on the chunk frames."""
    cfg = sd.TINY
    torch.manual_seed(0)
    uw, cw = sd.make_unet_weights(cfg, 1), sd.make_controlnet_weights(cfg, 2)
    f, h = 6, 8
    lat = torch.randn(f, 4, h, h); disp = torch.rand(f, 3, 8 * h, 8 * h)
    cn, cp = torch.randn(1, cfg["text_len"], cfg["cross_dim"]), torch.randn(1, cfg["text_len"], cfg["cross_dim"])
    a = sd.denoise_chunk(uw, cw, lat, disp, cn, cp, 5.0, 2, cfg, 20)
    assert a.shape == lat.shape and torch.isfinite(a).all()
    lat2 = lat.clone(); lat2[4:] = torch.randn(2, 4, h, h)
    b = sd.denoise_chunk(uw, cw, lat2, disp, cn, cp, 5.0, 2, cfg, 20)
    assert torch.allclose(a[:4], b[:4], atol=1e-5)
    assert not torch.allclose(a[4:], b[4:], atol=1e-3)


def test_ddim_schedule():
    s = sd.DDIM()
    assert s.timesteps(20).tolist() == list(range(951, 0, -50))
    assert s.timesteps(20, inverse=True).tolist() == list(range(1, 1000, 50))
    x = torch.randn(2, 4, 8, 8); e = torch.randn(2, 4, 8, 8)
    # one inverse step followed by the matching forward step with the same eps is the identity
    y = s.inverse_step(e, 51, x, 20)
    np.testing.assert_allclose(s.step(e, 51, y, 20).numpy(), x.numpy(), atol=1e-5)


def test_vae_decode_shape():
    w = sd.make_vae_decoder_weights(sd.VAE_TINY, 3)
    img = sd.vae_decode(w, torch.randn(1, 4, 4, 4), sd.VAE_TINY)
    assert img.shape == (1, 3, 32, 32)


def test_layernorm_fold_algebra():
    """weights.prepare(fold_ln=True): LN(x) W^T + b == rstd (x W'^T - mean colsum) + b' with W' = W diag(gamma), b' = b + W beta --
    checked in fp32 on the CPU for the three folded GEMMs of a transformer block (Q|K|V, attn2.to_q, GEGLU projection)."""
    import torch.nn.functional as F
    from gaussctrl_amd.sd import weights as W
    cfg = sd.TINY
    uw = sd.make_unet_weights(cfg, 1)
    out = W.prepare(uw, torch.float32, "cpu", heads=cfg["heads"], fold_ln=True)
    t = "down_blocks.0.attentions.0.transformer_blocks.0."
    C = cfg["block_out_channels"][0]
    x = torch.randn(10, C, generator=torch.Generator().manual_seed(0)) * 2 + 0.7
    mean = x.mean(1, keepdim=True); rstd = ((x * x).mean(1, keepdim=True) - mean ** 2 + 1e-5).rsqrt()
    fold = lambda name: rstd * (x @ out[name + ".weight"].T - mean * out[name + ".colsum"][None]) + out[name + ".bias"][None]
    sc = (C // cfg["heads"]) ** -0.5 * W.LOG2E
    ln1 = F.layer_norm(x, (C,), uw[t + "norm1.weight"], uw[t + "norm1.bias"], 1e-5)
    ref = ln1 @ torch.cat([uw[t + "attn1.to_q.weight"] * sc, uw[t + "attn1.to_k.weight"], uw[t + "attn1.to_v.weight"]], 0).T
    assert float((fold(t + "attn1.to_qkv") - ref).abs().max()) < 5e-6
    ln2 = F.layer_norm(x, (C,), uw[t + "norm2.weight"], uw[t + "norm2.bias"], 1e-5)
    assert float((fold(t + "attn2.to_q") - ln2 @ (uw[t + "attn2.to_q.weight"] * sc).T).abs().max()) < 5e-6
    ln3 = F.layer_norm(x, (C,), uw[t + "norm3.weight"], uw[t + "norm3.bias"], 1e-5)
    pr = ln3 @ uw[t + "ff.net.0.proj.weight"].T + uw[t + "ff.net.0.proj.bias"]
    n = pr.shape[1] // 2
    idx = torch.arange(n).reshape(n // 16, 16); perm = torch.cat([idx, idx + n], 1).reshape(-1)
    assert float((fold(t + "ff.net.0.proj") - pr[:, perm]).abs().max()) < 5e-6


def test_e4m3_emulation_hooks():
    """sd.FP8_EMU (the arithmetic of the product's fp8 path restated on the oracle): inert when no site qualifies (bit-identical to the fp32
    oracle), every site class moves the result when on, the weight rounding is the product's (row maximum in e4m3's top binade, at most half
    an e4m3 step per element), and the whole-network distance at a tiny geometry is of e4m3's order (3 mantissa bits), not a blow-up."""
    cfg = sd.TINY
    torch.manual_seed(0)
    uw, cw = sd.make_unet_weights(cfg, 1), sd.make_controlnet_weights(cfg, 2)
    f, h = 5, 8
    lat = torch.randn(f, 4, h, h); disp = torch.rand(f, 3, 8 * h, 8 * h)
    cn, cp = torch.randn(1, cfg["text_len"], cfg["cross_dim"]), torch.randn(1, cfg["text_len"], cfg["cross_dim"])
    run = lambda: sd.denoise_chunk(uw, cw, lat, disp, cn, cp, 5.0, 2, cfg, 20)
    plain = run()
    outs = {}
    try:
        sd.FP8_EMU = {"min_hw": 1 << 30, "min_rows": 1 << 30, "linears": 7, "cache": {}}
        assert torch.equal(run(), plain)
        for name, emu in (("convs", dict(min_hw=4, min_rows=1 << 30, linears=0)), ("ff", dict(min_hw=1 << 30, min_rows=1, linears=1)),
                          ("to_q", dict(min_hw=1 << 30, min_rows=1, linears=2)), ("qkv", dict(min_hw=1 << 30, min_rows=1, linears=4)),
                          ("all", dict(min_hw=4, min_rows=1, linears=7))):
            sd.FP8_EMU = dict(emu, cache={})
            outs[name] = float((run() - plain).norm() / plain.norm())
        w = torch.randn(64, 3, 3, 3) * torch.exp2(torch.randint(-8, 8, (64, 1, 1, 1)).float())
        sd.FP8_EMU = {"cache": {}}
        q = sd._q8_rows(w, "w")
    finally:
        sd.FP8_EMU = None
    assert all(v > 1e-5 for v in outs.values()), outs                    # every site class is live (TINY has C = 128 levels)
    assert 1e-3 < outs["all"] < 1e-1, outs
    amax = w.abs().amax(dim=(1, 2, 3), keepdim=True)
    assert bool(((q - w).abs() <= 0.0626 * w.abs() + amax * 2.0 ** -9).all())
    assert torch.equal(run(), plain)                                     # and off again


def test_activation_storage_emulation():
    """sd.ACT_ROUND (every tensor the product stores between kernels rounded to bf16 / f16, arithmetic fp32): off = the fp32 oracle bit for
    bit; on, the latents move by the storage type's order -- and bf16 costs ~8x f16 (3 fewer mantissa bits), the ratio the parity bars of
    tests/test_fullgeom_gpu.py assume (1e-3 for f16 = north_star's, 8e-3 for bf16)."""
    cfg = sd.TINY
    torch.manual_seed(0)
    uw, cw = sd.make_unet_weights(cfg, 1), sd.make_controlnet_weights(cfg, 2)
    f, h = 5, 8
    lat = torch.randn(f, 4, h, h); disp = torch.rand(f, 3, 8 * h, 8 * h)
    cn, cp = torch.randn(1, cfg["text_len"], cfg["cross_dim"]), torch.randn(1, cfg["text_len"], cfg["cross_dim"])
    run = lambda: sd.denoise_chunk(uw, cw, lat, disp, cn, cp, 5.0, 3, cfg, 20)
    plain = run()
    d = {}
    try:
        for dt in (torch.bfloat16, torch.float16):
            sd.ACT_ROUND = dt
            d[dt] = float((run() - plain).norm() / plain.norm())
    finally:
        sd.ACT_ROUND = None
    assert torch.equal(run(), plain)
    assert 1e-3 < d[torch.bfloat16] < 2e-2 and 1e-4 < d[torch.float16] < 2.5e-3, d
    assert 5.0 < d[torch.bfloat16] / d[torch.float16] < 12.0, d
