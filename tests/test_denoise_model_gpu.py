"""GPU parity of the whole denoise step (ControlNet + UNet + CFG + DDIM, cross-view attention) against the CPU
fp32 oracle (oracle/sd15_torch.py) on the same seeded inputs and the same (2-byte-rounded) random weights of the
exact SD1.5 / sd-controlnet-depth shapes, at a latent size the oracle finishes in seconds.

Bars (BASELINE.json north_star: "UNet latents within 1e-3 rel fp16"): relative L2 error of the latents after the
step(s): f16 <= 1e-3, bf16 <= 8e-3 (bf16 carries 3 fewer mantissa bits than the reference's fp16)."""
import numpy as np
import pytest
from _margins import within
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = {torch.float16: 1e-3, torch.bfloat16: 8e-3}


def _rel(a, b):
    a = a.double().cpu(); b = b.double().cpu()
    return float((a - b).norm() / b.norm())


@pytest.fixture(scope="module")
def sd15():
    from oracle import sd15_torch as sd
    import _nets
    torch.manual_seed(0)
    uw, cw = _nets.raw_sd15(rounded=False)          # (the seeded generators' tensors, shared with the full-geometry tests)
    return sd, uw, cw


def _inputs(f, h, seed=2):
    g = torch.Generator().manual_seed(seed)
    lat = torch.randn(f, 4, h, h, generator=g)
    disp = torch.rand(f, 3, 8 * h, 8 * h, generator=g)
    cn = torch.randn(1, 77, 768, generator=g); cp = torch.randn(1, 77, 768, generator=g)
    return lat, disp, cn, cp


def _round(w, dt):
    return {k: v.to(dt).float() for k, v in w.items()}


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_edit_chunk_matches_oracle(sd15, dt):
    sd, uw, cw = sd15
    from gaussctrl_amd.sd.pipeline import DenoisePipeline
    from gaussctrl_amd.sd.weights import prepare
    f, h, steps = 5, 16, 2                      # 4 references + 1 chunk frame, CFG batch 10
    lat, disp, cn, cp = _inputs(f, h)
    r = lambda x: x.to(dt).float()
    uwr, cwr = _round(uw, dt), _round(cw, dt)
    torch.set_num_threads(torch.get_num_threads())
    ref = sd.denoise_chunk(uwr, cwr, lat, r(disp), r(cn), r(cp), 5.0, steps, sd.SD15, 20)
    pipe = DenoisePipeline(prepare(uw, dt, DEV, heads=8), prepare(cw, dt, DEV, heads=8), None, 20, 5.0)
    got = pipe.edit_chunk(lat.to(DEV), disp.to(DEV), cn.to(DEV), cp.to(DEV), steps=steps)
    e = _rel(got, ref)
    print(f"edit_chunk {dt}: latent rel L2 err after {steps} steps = {e:.3e}")
    within("e", e, TOL[dt])
    # reference K/V cache: the chunk frame alone against cached reference K / V^T gives the same latents
    bank = pipe.build_ref_bank(lat[:4].to(DEV), disp[:4].to(DEV), cn.to(DEV), cp.to(DEV), steps=steps)
    got_c = pipe.edit_chunk_cached(lat[4:].to(DEV), disp[4:].to(DEV), cn.to(DEV), cp.to(DEV), bank, steps=steps)
    ec = _rel(got_c, got[4:])
    print(f"cached-reference path vs in-batch references: rel L2 diff {ec:.3e}")
    within("ec", ec, TOL[dt])
    within("_rel(got_c, ref[4:])", _rel(got_c, ref[4:]), TOL[dt] * 1.5)


@pytest.mark.parametrize("dt", [torch.float16])
def test_unet_eps_and_inversion_match_oracle(sd15, dt):
    sd, uw, cw = sd15
    from gaussctrl_amd.sd.pipeline import DenoisePipeline
    from gaussctrl_amd.sd.weights import prepare
    f, h = 2, 16
    lat, disp, cn, cp = _inputs(f, h, seed=5)
    r = lambda x: x.to(dt).float()
    uwr, cwr = _round(uw, dt), _round(cw, dt)
    # DDIM inversion: plain attention, no CFG, batched views (gc_pipeline.py:136-145)
    sch = sd.DDIM()
    x = lat.clone()
    ctx = r(cp).expand(f, -1, -1)
    for t in sch.timesteps(20, inverse=True)[:2]:
        down, mid = sd.controlnet_forward(cwr, x, t, ctx, r(disp), sd.SD15, 1.0, "plain", 0.0)
        eps = sd.unet_forward(uwr, x, t, ctx, down, mid, sd.SD15, "plain", 0.0)
        x = sch.inverse_step(eps, t, x, 20)
    pipe = DenoisePipeline(prepare(uw, dt, DEV, heads=8), prepare(cw, dt, DEV, heads=8), None, 20, 5.0)
    got = pipe.invert(lat.to(DEV), disp.to(DEV), cp.to(DEV), steps=2)
    e = _rel(got, x)
    print(f"inversion {dt}: rel L2 err {e:.3e}")
    within("e", e, TOL[dt])


def test_vae_decode_matches_oracle():
    from oracle import sd15_torch as sd
    from gaussctrl_amd.sd.pipeline import DenoisePipeline
    from gaussctrl_amd.sd.vae import VAEDecoder, prepare_vae_weights
    from gaussctrl_amd.sd.pipeline import to_nhwc8
    dt = torch.float16
    vw = sd.make_vae_decoder_weights(sd.VAE_SD, 300)
    z = torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(1))
    ref = sd.postprocess_image(sd.vae_decode({k: v.to(dt).float() for k, v in vw.items()}, (z / 0.18215).to(dt).float(), sd.VAE_SD))
    dec = VAEDecoder(prepare_vae_weights(vw, dt, DEV))
    got = dec.decode(to_nhwc8((z / 0.18215).to(DEV), dt), postprocess=True)[..., :3].permute(0, 3, 1, 2)
    err = float((got.cpu() - ref).abs().max())
    print(f"vae decode max abs err {err:.3e}")
    within("err", err, 1.0 / 255.0)  # images in [0,1]; f16 activations through 30 convs


def test_pipeline_decode_groups_large_batches(monkeypatch):
    """DenoisePipeline.decode splits a batch whose widest full-resolution map would pass 2^31 elements (the kernels' 32-bit element offsets; 15 frames at
    512 x 512) into groups of frames: the grouped path (forced here at a small size) equals one pass -- frames are independent (GroupNorm is per sample)."""
    from oracle import sd15_torch as sd
    from gaussctrl_amd.sd.pipeline import DenoisePipeline
    from gaussctrl_amd.sd.vae import prepare_vae_weights
    assert DenoisePipeline.decode_group(64, 64) == 15 and DenoisePipeline.decode_group(8, 8) >= 960
    dt = torch.float16          # (kernel plans follow the row count: the two paths differ by accumulation order -- one 8-bit level in f16, several in bf16)
    pipe = DenoisePipeline.__new__(DenoisePipeline)          # decode() needs the VAE and the dtype only
    from gaussctrl_amd.sd.vae import VAEDecoder
    pipe.vae = VAEDecoder(prepare_vae_weights(sd.make_vae_decoder_weights(sd.VAE_SD, 300), dt, DEV)); pipe.dtype = dt
    z = torch.randn(5, 4, 8, 8, generator=torch.Generator().manual_seed(2)).to(DEV)
    one = pipe.decode(z)
    monkeypatch.setattr(DenoisePipeline, "decode_group", staticmethod(lambda h, w: 2))
    grouped = pipe.decode(z)                                  # 2 + 2 + 1 frames
    assert grouped.shape == one.shape == (5, 3, 64, 64)
    within("grouped vs one-pass decode (max abs, [0,1] image)", float((grouped - one).abs().max()), 2.0 / 255.0)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_cfg_shared_prefix_equals_duplicated_computation(sd15, dt, monkeypatch):
    """KernelOptions.cfg_share (sd.unet.AttnCtx.share, default on): conv_in .. the first transformer block's cross-view self-attention run for ONE
    of the two identical CFG halves ([uncond ; cond] copies of the same latents: cat([latents] * 2) in the diffusers pipeline the reference
    calls, /root/reference/gaussctrl/gc_pipeline.py:209-219).  Against the duplicated computation (cfg_share = False): BIT-identical latents and
    BIT-identical reference bank in batch-invariant mode (full SD1.5 widths, 16 x 16 latents, 2 steps, in-batch references AND bank + chunk);
    in the default planning the two differ by accumulation-order noise only (inside the dtype's bar)."""
    import dataclasses
    sd, uw, cw = sd15
    from gaussctrl_amd.sd import ops
    from gaussctrl_amd.sd.pipeline import DenoisePipeline
    from gaussctrl_amd.sd.weights import prepare
    lat, disp, cn, cp = _inputs(6, 16)
    to = lambda t: t.to(DEV)
    pipe = DenoisePipeline(prepare(uw, dt, DEV, heads=8, fold_ln=2), prepare(cw, dt, DEV, heads=8, fold_ln=2), None, 20, 5.0)
    keep = ops.OPTIONS

    def run(share, invariant):
        monkeypatch.setattr(ops, "OPTIONS", dataclasses.replace(keep, cfg_share=share))
        monkeypatch.setattr(ops, "BATCH_INVARIANT", invariant)
        a = pipe.edit_chunk(to(lat), to(disp), to(cn), to(cp), steps=2)
        bank = pipe.build_ref_bank(to(lat[:4]), to(disp[:4]), to(cn), to(cp), steps=2)
        b = pipe.edit_chunk_cached(to(lat[4:]), to(disp[4:]), to(cn), to(cp), bank, steps=2)
        return a, b, bank
    a1, b1, k1 = run(True, True)
    a0, b0, k0 = run(False, True)
    assert torch.equal(a1, a0) and torch.equal(b1, b0)
    assert set(k1.store) == set(k0.store)
    for key in k0.store:
        assert k1.store[key][0].shape == k0.store[key][0].shape and k1.store[key][0].stride(1) == k0.store[key][0].stride(1), key
        assert torch.equal(k1.store[key][0], k0.store[key][0]) and torch.equal(k1.store[key][1], k0.store[key][1]), key
    a1, b1, _ = run(True, False)
    a0, b0, _ = run(False, False)
    within("shared vs duplicated, in-batch (default planning)", _rel(a1, a0), TOL[dt])
    within("shared vs duplicated, bank + chunk (default planning)", _rel(b1, b0), TOL[dt])
