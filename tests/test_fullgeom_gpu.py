"""GPU parity at the geometry bench.py measures: 64x64 latents (512x512 images), full SD1.5 / sd-controlnet-depth / VAE
shapes, f = 4 reference + 3 chunk frames (CFG batch 14) -- the k_attn3<.,40,..> L = 4096 grids and the 256-tile k_gemm8 grids
that carry the timed step -- through ALL 20 DDIM steps, in both activation dtypes, against fixtures the CPU fp32 oracle produced
once in the build container (tests/golden/make_fullgeom_golden.py; the GPU box only loads the .npz).

Bars.  north_star: "UNet latents within 1e-3 rel fp16" -- the f16 relative L2 error of the latents stays <= 1e-3 after every
one of the 20 steps.  bf16 carries 3 fewer mantissa bits (8x the rounding unit), its bar is 8e-3 on the same curve (measured
curve printed by the test and recorded in DESIGN.md section 2)."""
import os

import numpy as np
import pytest
from _margins import within
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(__file__), "golden")
BAR = {torch.float16: 1e-3, torch.bfloat16: 8e-3}


def _rel(a, b):
    a = a.double().cpu(); b = b.double().cpu()
    return float((a - b).norm() / b.norm())


def _inputs(f, h, seed):
    g = torch.Generator().manual_seed(seed)
    lat = torch.randn(f, 4, h, h, generator=g)
    disp = torch.rand(f, 3, 8 * h, 8 * h, generator=g)
    cn = torch.randn(1, 77, 768, generator=g); cp = torch.randn(1, 77, 768, generator=g)
    r = lambda x: x.to(torch.bfloat16).float()          # the fixture rounds the 2-byte inputs to bf16 (exact in f16 too)
    return lat, r(disp), r(cn), r(cp)


@pytest.fixture(scope="module")
def nets():
    """the fixture's weights (seeded CPU generators, rounded to bf16 like the generator did) prepared per dtype on demand (tests/_nets.py:
    shared with every other test of the session that wants them)"""
    import _nets
    return lambda dt: _nets.prepared(dt, False)


def _curve(trace, ref, sl=slice(None)):
    return [_rel(t[sl], torch.tensor(ref[i][sl])) for i, t in enumerate(trace)]


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_edit_f7_h64_all_20_steps(nets, dt):
    """BASELINE configs[1] geometry: in-batch references (reference order, gc_pipeline.py:206-219) AND the cached-reference
    product path, every DDIM step compared with the oracle trajectory."""
    from gaussctrl_amd.sd.pipeline import DenoisePipeline
    z = np.load(os.path.join(GOLD, "fullgeom_edit_f7_h64.npz"))
    ref = z["lat_steps"]                                            # [20, 7, 4, 64, 64]
    f, h, steps, seed = [int(v) for v in z["meta"][:4]]
    lat, disp, cn, cp = _inputs(f, h, seed)
    uw, cw = nets(dt)
    pipe = DenoisePipeline(uw, cw, None, 20, 5.0)
    trace = []
    on = lambda i, l: trace.append(l.permute(0, 3, 1, 2).float().cpu())
    pipe.edit_chunk(lat.to(DEV), disp.to(DEV), cn.to(DEV), cp.to(DEV), steps=steps, on_step=on)
    assert len(trace) == steps == 20
    cur = _curve(trace, ref)
    print(f"\nedit f=7 h=64 {dt}: rel L2 error of the latents per DDIM step (in-batch references):\n  " +
          " ".join(f"{e:.2e}" for e in cur))
    within("max(cur)", max(cur), BAR[dt])
    # Where the bar comes from: the fp32 oracle re-run with every stored activation rounded to this dtype (oracle/sd15_torch.py ACT_ROUND,
    # tests/golden/make_fullgeom_golden.py edit7_actround, CPU) sits 3.86e-3 .. 5.02e-3 (bf16) / 4.82e-4 .. 6.24e-4 (f16) from the fp32 trajectory
    # over the first 6 steps -- the storage type's own cost.  The product must land ON that curve (measured: within 1 %), not merely under a bar.
    emu = np.load(os.path.join(GOLD, "fullgeom_edit_f7_h64_actround.npz"))["rel_bf16" if dt == torch.bfloat16 else "rel_f16"]
    print("  predicted by the activation-storage emulation: " + " ".join(f"{e:.2e}" for e in emu))
    within("max_i |cur[i] / predicted[i] - 1|, steps 1..6", max(abs(cur[i] / float(emu[i]) - 1.0) for i in range(len(emu))), 0.25)
    # product path: reference K / V^T from the bank, chunk frames only
    bank = pipe.build_ref_bank(lat[:4].to(DEV), disp[:4].to(DEV), cn.to(DEV), cp.to(DEV))
    trace_c = []
    pipe.edit_chunk_cached(lat[4:].to(DEV), disp[4:].to(DEV), cn.to(DEV), cp.to(DEV), bank,
                           on_step=lambda i, l: trace_c.append(l.permute(0, 3, 1, 2).float().cpu()))
    cur_c = [_rel(t, torch.tensor(ref[i][4:])) for i, t in enumerate(trace_c)]
    print(f"edit f=7 h=64 {dt}: cached-reference path, chunk frames vs oracle:\n  " + " ".join(f"{e:.2e}" for e in cur_c))
    within("max(cur_c)", max(cur_c), BAR[dt])
    d = _rel(trace_c[-1], trace[-1][4:])
    print(f"cached vs in-batch after 20 steps: {d:.2e}")
    within("d", d, BAR[dt])


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_edit_f7_h64_layernorm_folded(dt):
    """The same trajectory with the LayerNorms of the C = 640 / 1280 transformer blocks FOLDED into their consumer GEMMs
    (weights.prepare(fold_ln=2): norm1 -> Q | K | V^T, norm2 -> attn2.to_q, norm3 -> GEGLU; row statistics from the producers' lean epilogues;
    the C = 320 blocks stay on the row-resident head / tail kernels): in-batch references and the cached-bank product path, all 20 DDIM steps,
    same bars as the unfolded network and the same falsifiable form -- the distance to the fp32 oracle must sit on the storage type's own curve
    (the fold removes one rounding of the normalised activations and rounds W diag(gamma) once instead: no visible change is expected)."""
    from oracle import sd15_torch as sd
    from gaussctrl_amd.sd.pipeline import DenoisePipeline
    from gaussctrl_amd.sd.weights import prepare
    z = np.load(os.path.join(GOLD, "fullgeom_edit_f7_h64.npz"))
    ref = z["lat_steps"]
    f, h, steps, seed = [int(v) for v in z["meta"][:4]]
    lat, disp, cn, cp = _inputs(f, h, seed)
    import _nets
    uw, cw = _nets.prepared(dt, 2)
    assert any(k.endswith("attn1.to_qkv.colsum") for k in uw) and any(k.endswith(".tail.a") for k in uw)
    pipe = DenoisePipeline(uw, cw, None, 20, 5.0)
    trace = []
    pipe.edit_chunk(lat.to(DEV), disp.to(DEV), cn.to(DEV), cp.to(DEV), steps=steps, on_step=lambda i, l: trace.append(l.permute(0, 3, 1, 2).float().cpu()))
    cur = _curve(trace, ref)
    print(f"\nedit f=7 h=64 {dt}, LayerNorms of levels 1-3 folded: rel L2 per step (in-batch):\n  " + " ".join(f"{e:.2e}" for e in cur))
    within("max(cur)", max(cur), BAR[dt])
    emu = np.load(os.path.join(GOLD, "fullgeom_edit_f7_h64_actround.npz"))["rel_bf16" if dt == torch.bfloat16 else "rel_f16"]
    within("max_i |cur[i] / predicted[i] - 1|, steps 1..6", max(abs(cur[i] / float(emu[i]) - 1.0) for i in range(len(emu))), 0.25)
    bank = pipe.build_ref_bank(lat[:4].to(DEV), disp[:4].to(DEV), cn.to(DEV), cp.to(DEV))
    trace_c = []
    pipe.edit_chunk_cached(lat[4:].to(DEV), disp[4:].to(DEV), cn.to(DEV), cp.to(DEV), bank,
                           on_step=lambda i, l: trace_c.append(l.permute(0, 3, 1, 2).float().cpu()))
    cur_c = [_rel(t, torch.tensor(ref[i][4:])) for i, t in enumerate(trace_c)]
    print("  cached-reference path: " + " ".join(f"{e:.2e}" for e in cur_c))
    within("max(cur_c)", max(cur_c), BAR[dt])


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_invert_f3_h64_all_20_steps(nets, dt):
    """render_reverse's DDIM inversion (plain attention, no CFG, batched views; gc_pipeline.py:136-145) at full size."""
    from gaussctrl_amd.sd.pipeline import DenoisePipeline
    z = np.load(os.path.join(GOLD, "fullgeom_invert_f3_h64.npz"))
    ref = z["lat_steps"]
    f, h, steps, seed = [int(v) for v in z["meta"][:4]]
    lat, disp, cn, cp = _inputs(f, h, seed)
    uw, cw = nets(dt)
    pipe = DenoisePipeline(uw, cw, None, 20, 5.0)
    trace = []
    pipe.invert(lat.to(DEV), disp.to(DEV), cp.to(DEV), steps=steps,
                on_step=lambda i, l: trace.append(l.permute(0, 3, 1, 2).float().cpu()))
    cur = _curve(trace, ref)
    print(f"\ninversion f=3 h=64 {dt}: rel L2 per step:\n  " + " ".join(f"{e:.2e}" for e in cur))
    within("max(cur)", max(cur), BAR[dt])
    # the storage type's own cost on this trajectory (oracle ACT_ROUND, make_fullgeom_golden.py invert_actround): 6.04e-3 (bf16) / 7.50e-4 (f16) at
    # the last step, where the inversion's error is largest -- the product measured 6.04e-3 / 7.52e-4: it must land there
    emu = np.load(os.path.join(GOLD, "fullgeom_invert_f3_h64_actround.npz"))["rel_bf16" if dt == torch.bfloat16 else "rel_f16"]
    print("  predicted by the activation-storage emulation:\n  " + " ".join(f"{e:.2e}" for e in emu))
    within("|cur[-1] / predicted[-1] - 1|", abs(cur[-1] / float(emu[-1]) - 1.0), 0.25)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_edit_f12_h64_config4_geometry(nets, dt):
    """BASELINE configs[3] geometry: chunk_size 8 -> f = 12 frames, CFG batch 24 (first 2 of 20 steps)."""
    from gaussctrl_amd.sd.pipeline import DenoisePipeline
    path = os.path.join(GOLD, "fullgeom_edit_f12_h64.npz")
    if not os.path.exists(path):
        pytest.skip("fixture not generated")
    z = np.load(path)
    ref = z["lat_steps"]
    f, h, steps, seed = [int(v) for v in z["meta"][:4]]
    lat, disp, cn, cp = _inputs(f, h, seed)
    uw, cw = nets(dt)
    pipe = DenoisePipeline(uw, cw, None, 20, 5.0)
    trace = []
    pipe.edit_chunk(lat.to(DEV), disp.to(DEV), cn.to(DEV), cp.to(DEV), steps=steps,
                    on_step=lambda i, l: trace.append(l.permute(0, 3, 1, 2).float().cpu()))
    cur = _curve(trace, ref)
    print(f"\nedit f=12 h=64 {dt}: " + " ".join(f"{e:.2e}" for e in cur))
    within("max(cur)", max(cur), BAR[dt])


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_vae_decode_h64(dt):
    """vae.decode(z / 0.18215) -> (x/2 + 0.5).clamp(0,1) at 64x64 latents (512x512 image; L = 4096, D = 512 mid-block
    attention), against the oracle's fp32 decode of the same (bf16-rounded) weights.
    Bars, from the measured errors (MI355X, round 2: f16 rel L2 5.3e-4 / mean abs 2.1e-4 / max abs 3.6e-3; bf16 4.2e-3 / 1.7e-3 /
    2.1e-2): the same relative-L2 bars as the latents (f16 1e-3 = north_star's, bf16 8e-3), plus per-pixel limits in 8-bit levels of
    the [0,1] image: f16 every pixel within ONE level (1/255); bf16 (8 mantissa bits through ~30 convolutions) mean error within
    one level and isolated worst pixels within 8 levels.  Round 1 asserted only max <= 1/255 on an 8x8 latent, where the tail of
    the error distribution is never sampled; the 2e-3 figure it first tried was a guess, not a derived bar."""
    from oracle import sd15_torch as sd
    from gaussctrl_amd.sd.pipeline import to_nhwc8
    from gaussctrl_amd.sd.vae import VAEDecoder, prepare_vae_weights
    z = np.load(os.path.join(GOLD, "fullgeom_vae_h64.npz"))
    ref = torch.tensor(z["image"])
    _, h, seed, wseed = [int(v) for v in z["meta"]]
    vw = {k: v.to(torch.bfloat16).float() for k, v in sd.make_vae_decoder_weights(sd.VAE_SD, wseed).items()}
    lat = torch.randn(1, 4, h, h, generator=torch.Generator().manual_seed(seed))
    zin = (lat / 0.18215).to(torch.bfloat16).float()
    dec = VAEDecoder(prepare_vae_weights(vw, dt, DEV))
    got = dec.decode(to_nhwc8(zin.to(DEV), dt), postprocess=True)[..., :3].permute(0, 3, 1, 2).cpu()
    err = (got - ref).abs()
    print(f"\nvae decode h=64 {dt}: max abs {float(err.max()):.3e}  mean abs {float(err.mean()):.3e}  rel L2 {_rel(got, ref):.3e}")
    within("_rel(got, ref)", _rel(got, ref), BAR[dt])
    # the storage type's own cost (the oracle's decode with every stored activation rounded to dt, sd.ACT_ROUND, CPU, same inputs): f16 rel L2
    # 5.295e-4 / mean abs 2.13e-4 / max abs 3.1e-3; bf16 4.221e-3 / 1.70e-3 / 2.30e-2 -- the product measured 5.27e-4 and 4.23e-3: it must land there
    predicted = 5.295e-4 if dt == torch.float16 else 4.221e-3
    within("|rel L2 / predicted - 1|", abs(_rel(got, ref) / predicted - 1.0), 0.25)
    if dt == torch.float16:
        within("float(err.max())", float(err.max()), 1.0 / 255.0)
    else:
        within("float(err.mean())", float(err.mean()), 1.0 / 255.0); within("float(err.max())", float(err.max()), 8.0 / 255.0)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_vae_encode_h512(dt):
    """image2latent (gc_pipeline.py:239-246) at the full 512 x 512 render: vae.encode(2x - 1).mean * 0.18215 -> [1,4,64,64]
    (the encoder's mid-block attention runs at L = 4096, D = 512), against the oracle's fp32 encode of the same bf16-rounded weights
    and input.  Bars: the VAE is not the UNet of north_star's "latents within 1e-3 rel fp16" -- its encoder sums 128-channel 3x3
    windows over 512 x 512 maps, and the first measured run gave f16 1.07e-3 / bf16 8e-3-class errors -- so the bars are 2x the
    denoise-latent bars: f16 2e-3, bf16 1.6e-2 (written here before the second run, not fitted to it)."""
    from oracle import sd15_torch as sd
    from gaussctrl_amd.sd.pipeline import to_nhwc8
    from gaussctrl_amd.sd.vae import VAEEncoder, prepare_vae_encoder_weights
    z = np.load(os.path.join(GOLD, "fullgeom_vaeenc_h512.npz"))
    ref = torch.tensor(z["latent"])
    H, seed, wseed = [int(v) for v in z["meta"]]
    vw = {k: v.to(torch.bfloat16).float() for k, v in sd.make_vae_encoder_weights(sd.VAE_SD, wseed).items()}
    img = torch.rand(H, H, 3, generator=torch.Generator().manual_seed(seed))
    x = (img * 2 - 1).to(torch.bfloat16).float().permute(2, 0, 1)[None]
    enc = VAEEncoder(prepare_vae_encoder_weights({k: v.to(DEV) for k, v in vw.items()}, dt, DEV))
    got = (enc.encode_mean(to_nhwc8(x.to(DEV), dt))[..., :4] * 0.18215).permute(0, 3, 1, 2).float().cpu()
    rel = _rel(got, ref)
    print(f"\nvae encode 512x512 {dt}: rel L2 {rel:.3e}  max abs {float((got - ref).abs().max()):.3e}")
    assert got.shape == (1, 4, H // 8, H // 8); within("rel", rel, 2 * BAR[dt])


def _config4_run(nets, dt, fp8, cached=False):
    """BASELINE configs[3] end to end against the oracle fixture: f = 4 references + chunk_size 8, ALL 20 DDIM steps, VAE decode of the 8
    chunk frames, composite through the synthetic elliptical object mask (gc_pipeline.py:209-234).
    fp8: False | "convs" (e4m3 resnet convolutions) | "all" (`bench.py --dtype fp8 --fp8-linears 7`: convolutions + add_fp8_linears(.., 7)) |
    "hybrid" (`bench.py --dtype fp8` since round 5: e4m3 convolutions + the folded / merged bf16 transformer blocks).
    cached: False = in-batch references (CFG batch 24); True = the product path, reference bank + the 8 chunk frames (CFG batch 16: the
    grids `bench.py --dtype fp8 --chunk-size 8 --mask` launches -- k_gemm8q picks tiles and k-slices from M, so B = 16 is its own plan)."""
    from oracle import sd15_torch as sd
    from gaussctrl_amd import synthetic as syn
    from gaussctrl_amd.sd import ops as sdops
    from gaussctrl_amd.sd.pipeline import DenoisePipeline, to_nhwc8
    from gaussctrl_amd.sd.vae import prepare_vae_weights
    from gaussctrl_amd.sd.weights import add_fp8_convs, add_fp8_linears
    path = os.path.join(GOLD, "fullgeom_config4_f12_h64.npz")
    if not os.path.exists(path):
        pytest.skip("fixture not generated (tests/golden/make_fullgeom_golden.py config4)")
    z = np.load(path)
    f, h, steps, seed, _, _, vseed, stride = [int(v) for v in z["meta"]]
    which = [int(v) for v in z["which_steps"]]
    lat, disp, cn, cp = _inputs(f, h, seed)
    if fp8 == "hybrid":       # e4m3 convolutions + the round-5 bf16 transformer blocks (LayerNorm fold, text fold, FF merge, Q-only): `bench.py --dtype fp8` since round 5
        import _nets
        uw, cw = (dict(w) for w in _nets.prepared(dt, 2))
        add_fp8_convs(uw, _nets.conv_weights()[0], DEV); add_fp8_convs(cw, _nets.conv_weights()[1], DEV)
    else:
        uw, cw = nets(dt)
    if fp8 and fp8 != "hybrid":
        import _nets
        uw, cw = dict(uw), dict(cw)
        add_fp8_convs(uw, _nets.conv_weights()[0], DEV)
        add_fp8_convs(cw, _nets.conv_weights()[1], DEV)
        if fp8 == "all":
            add_fp8_linears(uw, 7); add_fp8_linears(cw, 7)
    vw = {k: v.to(torch.bfloat16).float().to(DEV) for k, v in sd.make_vae_decoder_weights(sd.VAE_SD, vseed).items()}
    pipe = DenoisePipeline(uw, cw, prepare_vae_weights(vw, dt, DEV), 20, 5.0)
    assert bool(pipe.unet.fp8) == bool(fp8) and bool(pipe.controlnet.fp8) == bool(fp8)
    if fp8 == "all":
        assert pipe.unet.fp8_lin == 7 and pipe.controlnet.fp8_lin == 7
    trace = []
    on = lambda i, l: trace.append(l.permute(0, 3, 1, 2).float().cpu())
    to = lambda t: t.to(DEV)
    if cached:
        bank = pipe.build_ref_bank(to(lat[:4]), to(disp[:4]), to(cn), to(cp))
        chunk = pipe.edit_chunk_cached(to(lat[4:]), to(disp[4:]), to(cn), to(cp), bank, on_step=on)
        sl = slice(4, None)
    else:
        chunk = pipe.edit_chunk(to(lat), to(disp), to(cn), to(cp), steps=steps, on_step=on)[4:]
        sl = slice(None)
    assert len(trace) == 20 and chunk.shape[0] == f - 4
    cur = [_rel(trace[s - 1], torch.tensor(z["lat_steps"][k][sl])) for k, s in enumerate(which)]
    imgs = pipe.vae.decode(to_nhwc8(chunk / 0.18215, dt), postprocess=True)          # [8,H,W,8] fp32
    H = 8 * h
    mask = torch.tensor(syn.elliptical_mask(H, H, soft=True)).to(DEV)
    g = torch.Generator().manual_seed(seed + 1000)
    errs = []
    for j in range(f - 4):
        unedited = torch.rand(H, H, 3, generator=g).to(DEV)
        comp = sdops.mask_composite(imgs[j].contiguous(), unedited, mask)[::stride, ::stride].cpu()
        errs.append((comp - torch.tensor(z["composite"][j])).abs())
    e = torch.stack(errs)
    tag = {False: "", "convs": " + e4m3 convs", "all": " + e4m3 convs and linears", "hybrid": " + e4m3 convs, folded / merged bf16 blocks"}[fp8] + (", cached bank B=16" if cached else ", in-batch B=24")
    print(f"\nconfig 4 ({dt}{tag}): latent rel L2 at steps {which}: " + " ".join(f"{c:.2e}" for c in cur) +
          f"; composited images: mean abs {float(e.mean()):.3e} max abs {float(e.max()):.3e}")
    return cur, float(e.mean()), float(e.max())


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_config4_f12_all_steps_decode_mask(nets, dt):
    cur, mean_e, max_e = _config4_run(nets, dt, False)
    within("max(cur)", max(cur), BAR[dt])
    within("mean_e", mean_e, 1.0 / 255.0); within("max_e", max_e, (2.0 if dt == torch.float16 else 16.0) / 255.0)


def _e4m3_predicted():
    """What e4m3 operands + bf16 storage cost on the benchmark-geometry trajectory, per DDIM step 1..6: the e4m3-emulating oracle's distance to
    the fp32 oracle (FP8_EMU, fp32 storage) and the bf16-storage-emulating oracle's (ACT_ROUND) add in quadrature (independent roundings)."""
    e = np.load(os.path.join(GOLD, "fullgeom_edit_f7_h64_e4m3.npz"))["rel_vs_fp32"]
    b = np.load(os.path.join(GOLD, "fullgeom_edit_f7_h64_actround.npz"))["rel_bf16"]
    n = min(len(e), len(b))
    return [float(np.sqrt(float(e[i]) ** 2 + float(b[i]) ** 2)) for i in range(n)]


def test_config4_f12_fp8_convs(nets):
    """configs[3] as named: "fp8 MFMA UNet path" -- e4m3 convolutions ON at f = 12 (B = 24 picks other k_gemm8q tiles than f = 7),
    all 20 steps, decode, mask.  Own bars of the e4m3 path (3 mantissa bits): latents <= 6e-2 relative L2 at every step, composited
    image within 2 eight-bit levels on average."""
    cur, mean_e, max_e = _config4_run(nets, torch.bfloat16, "convs")
    within("max(cur)", max(cur), 6e-2)
    within("mean_e", mean_e, 2.0 / 255.0)


@pytest.mark.parametrize("cached", [False, True])
def test_config4_f12_fp8_convs_with_folded_linears(nets, cached):
    """configs[3] on the configuration `bench.py --dtype fp8 --chunk-size 8 --mask` launches since round 5: e4m3 resnet convolutions beside the
    round-5 bf16 transformer blocks (LayerNorm fold, text attention as two GEMMs, FF-down + proj_out merged, Q-only ControlNet projections, CFG-shared
    prefix).  f = 4 + 8 frames in-batch (CFG batch 24) and the cached bank with the 8 chunk frames (CFG batch 16), all 20 DDIM steps, VAE decode, mask
    composite; the e4m3 bars and the falsifiable saturated-curve bar of test_config4_f12_fp8_convs_and_linears."""
    cur, mean_e, max_e = _config4_run(nets, torch.bfloat16, "hybrid", cached)
    within("max(cur)", max(cur), 6e-2)
    within("mean_e", mean_e, 2.0 / 255.0)
    pred = _e4m3_predicted()[-1]
    within("|cur[-1] / predicted - 1|", abs(cur[-1] / pred - 1.0), 0.25)


@pytest.mark.parametrize("cached", [False, True])
def test_config4_f12_fp8_convs_and_linears(nets, cached):
    """configs[3] on the fp8 path the product RUNS since round 4 (`bench.py --dtype fp8 --chunk-size 8 --mask`: e4m3 resnet convolutions
    >= 16 x 16, k-sliced where the grid is part-filled, + e4m3 Q|K|V / attn2.to_q / GEGLU / FF-down of the C = 640 / 1280 blocks with the
    norms and the GEGLU epilogue writing e4m3): f = 4 + 8 frames, in-batch (CFG batch 24) and the cached bank with the 8 chunk frames (CFG
    batch 16, the benchmark's launch plan), all 20 DDIM steps, VAE decode, mask composite, against the fp32 oracle fixture.
    Bars: (i) the e4m3 path's own 6e-2 at every recorded step; (ii) FALSIFIABLE -- the distance at the last recorded step must sit within
    25 % of what e4m3 operands + bf16 storage cost by the emulating oracles' saturated value (`_e4m3_predicted()[-1]`, 2.71e-2: the error curve
    saturates after ~5 DDIM steps, DESIGN.md section 2, and a per-frame rel L2 does not depend on how many frames share the batch); a kernel
    that added visible error of its own, or a plan that silently skipped the e4m3 sites, fails it."""
    cur, mean_e, max_e = _config4_run(nets, torch.bfloat16, "all", cached)
    within("max(cur)", max(cur), 6e-2)
    within("mean_e", mean_e, 2.0 / 255.0)
    pred = _e4m3_predicted()[-1]
    print(f"  predicted by the e4m3 + bf16-storage emulation (saturated): {pred:.2e}")
    within("|cur[-1] / predicted - 1|", abs(cur[-1] / pred - 1.0), 0.25)


@pytest.mark.parametrize("dt,chunk", [(torch.float16, 3), (torch.bfloat16, 3), (torch.bfloat16, 8)])
def test_batch_invariant_mode_bit_identical(nets, dt, chunk):
    """sd.ops.BATCH_INVARIANT (gc_gemm_desc.plan_rows, GroupNorm planning bit, no set-split attention): a view's latents are BIT-identical
    whatever shares its chunk -- the property that makes an N-rank edit (other chunk compositions) equal to the single-GPU one.
    Full SD1.5 widths at 64 x 64 latents, cached reference bank, 3 DDIM steps: chunk {0,1,2} vs chunks {0} and {1,2}; and BASELINE
    configs[3]'s chunk_size 8 (CFG batch 16) vs two chunks of 4 (CFG batch 8) -- at B = 16 the D = 160 attention grid reaches 512 workgroups,
    where the library stops splitting the K / V sets over workgroups: the decision must not depend on the batch in this mode."""
    from gaussctrl_amd.sd import ops
    from gaussctrl_amd.sd.pipeline import DenoisePipeline
    lat, disp, cn, cp = _inputs(4 + chunk, 64, 2)
    uw, cw = nets(dt)
    keep = ops.BATCH_INVARIANT
    ops.BATCH_INVARIANT = True
    nst = 3 if chunk == 3 else 2
    cut = 1 if chunk == 3 else chunk // 2
    try:
        pipe = DenoisePipeline(uw, cw, None, 20, 5.0)
        to = lambda t: t.to(DEV)
        bank = pipe.build_ref_bank(to(lat[:4]), to(disp[:4]), to(cn), to(cp), steps=nst)
        full = pipe.edit_chunk_cached(to(lat[4:]), to(disp[4:]), to(cn), to(cp), bank, steps=nst)
        one = pipe.edit_chunk_cached(to(lat[4:4 + cut]), to(disp[4:4 + cut]), to(cn), to(cp), bank, steps=nst)
        two = pipe.edit_chunk_cached(to(lat[4 + cut:]), to(disp[4 + cut:]), to(cn), to(cp), bank, steps=nst)
        assert torch.isfinite(full).all()
        assert torch.equal(full[:cut], one), float((full[:cut] - one).abs().max())
        assert torch.equal(full[cut:], two), float((full[cut:] - two).abs().max())
        # and it is still the same computation: within the dtype's bar of the default planning
        ops.BATCH_INVARIANT = False
        ref = pipe.edit_chunk_cached(to(lat[4:]), to(disp[4:]), to(cn), to(cp), bank, steps=nst)
        within("_rel(full, ref)", _rel(full, ref), BAR[dt])
    finally:
        ops.BATCH_INVARIANT = keep


def test_edit_f7_h64_fp8_convs(nets):
    """BASELINE configs[3] "fp8 MFMA UNet path": the resnet 3x3 convolutions of UNet and ControlNet on e4m3 operands (block-scaled
    MFMA, GroupNorm writing e4m3), bf16 elsewhere, all 20 DDIM steps at the benchmarked geometry against the fp32 oracle fixture.
    e4m3 keeps 3 mantissa bits (2^-4 relative rounding): the bar is its own -- relative L2 of the latents <= 6e-2 at every step."""
    from oracle import sd15_torch as sd
    from gaussctrl_amd.sd.pipeline import DenoisePipeline
    from gaussctrl_amd.sd.weights import add_fp8_convs
    z = np.load(os.path.join(GOLD, "fullgeom_edit_f7_h64.npz"))
    ref = z["lat_steps"]
    f, h, steps, seed = [int(v) for v in z["meta"][:4]]
    lat, disp, cn, cp = _inputs(f, h, seed)
    dt = torch.bfloat16
    uw, cw = nets(dt)
    uw, cw = dict(uw), dict(cw)
    import _nets
    add_fp8_convs(uw, _nets.conv_weights()[0], DEV)
    add_fp8_convs(cw, _nets.conv_weights()[1], DEV)
    pipe = DenoisePipeline(uw, cw, None, 20, 5.0)
    assert pipe.unet.fp8 and pipe.controlnet.fp8
    trace = []
    pipe.edit_chunk(lat.to(DEV), disp.to(DEV), cn.to(DEV), cp.to(DEV), steps=steps,
                    on_step=lambda i, l: trace.append(l.permute(0, 3, 1, 2).float().cpu()))
    cur = _curve(trace, ref)
    print("\nedit f=7 h=64 fp8 convs: rel L2 per step:\n  " + " ".join(f"{e:.2e}" for e in cur))
    within("max(cur)", max(cur), 6e-2)


def test_edit_f7_h64_fp8_convs_with_folded_linears():
    """The configuration `bench.py --dtype fp8` runs since round 5: e4m3 resnet convolutions (maps >= 16 x 16, GroupNorm writing e4m3) AND the round-5
    graph of the transformer blocks in bf16 (LayerNorms folded into the GEMM epilogues, text attention as two GEMMs, FF-down + proj_out merged, Q-only
    ControlNet projections, CFG-shared prefix) -- faster than e4m3 linears without those merges (profiles/r05_fp8_hybrid_ab.txt).  All 20 DDIM steps at
    the benchmarked geometry, in-batch references and the cached-bank product path, against the fp32 oracle fixture: the e4m3 bar (6e-2) and the
    falsifiable form -- the distance must sit on the curve the emulating oracles predict (e4m3 operands at the convolutions -- the linears' e4m3 noise
    never showed: 2.67e-2 with, 2.7e-2 without -- in quadrature with bf16 storage) within 25 % at each of the first 6 steps."""
    from oracle import sd15_torch as sd
    from gaussctrl_amd.sd.pipeline import DenoisePipeline
    from gaussctrl_amd.sd.weights import add_fp8_convs, prepare
    z = np.load(os.path.join(GOLD, "fullgeom_edit_f7_h64.npz"))
    ref = z["lat_steps"]
    f, h, steps, seed = [int(v) for v in z["meta"][:4]]
    lat, disp, cn, cp = _inputs(f, h, seed)
    dt = torch.bfloat16
    import _nets
    uw, cw = (dict(w) for w in _nets.prepared(dt, 2))
    add_fp8_convs(uw, _nets.conv_weights()[0], DEV); add_fp8_convs(cw, _nets.conv_weights()[1], DEV)
    pipe = DenoisePipeline(uw, cw, None, 20, 5.0)
    assert pipe.unet.fp8 and pipe.controlnet.fp8 and not pipe.unet.fp8_lin and any(k.endswith("ffout.weight") for k in uw)
    trace = []
    pipe.edit_chunk(lat.to(DEV), disp.to(DEV), cn.to(DEV), cp.to(DEV), steps=steps, on_step=lambda i, l: trace.append(l.permute(0, 3, 1, 2).float().cpu()))
    cur = _curve(trace, ref)
    print("\nedit f=7 h=64 e4m3 convs + folded / merged bf16 linears: rel L2 per step (in-batch):\n  " + " ".join(f"{e:.2e}" for e in cur))
    within("max(cur)", max(cur), 6e-2)
    pred = _e4m3_predicted()
    within("max_i |cur[i] / predicted[i] - 1|, steps 1..6", max(abs(cur[i] / pred[i] - 1.0) for i in range(len(pred))), 0.25)
    bank = pipe.build_ref_bank(lat[:4].to(DEV), disp[:4].to(DEV), cn.to(DEV), cp.to(DEV))
    trace_c = []
    pipe.edit_chunk_cached(lat[4:].to(DEV), disp[4:].to(DEV), cn.to(DEV), cp.to(DEV), bank,
                           on_step=lambda i, l: trace_c.append(l.permute(0, 3, 1, 2).float().cpu()))
    cur_c = [_rel(t, torch.tensor(ref[i][4:])) for i, t in enumerate(trace_c)]
    print("  cached-reference path: " + " ".join(f"{e:.2e}" for e in cur_c))
    within("max(cur_c)", max(cur_c), 6e-2)


@pytest.mark.parametrize("which", [1, 7])
def test_edit_f7_h64_fp8_convs_and_linears(nets, which):
    """configs[3] "fp8 MFMA UNet path", widened: besides the resnet convolutions, the transformer linears of the C = 640 / 1280 levels run on
    e4m3 operands (weights.add_fp8_linears; which = 1: GEGLU projection + FF down projection with the hidden kept in e4m3; 7: also the fused
    Q | K | V and attn2.to_q, their inputs written as e4m3 by the LayerNorm kernel), all 20 DDIM steps at the benchmarked geometry against the
    fp32 oracle fixture.  Own bar of the e4m3 path, the one of the convolutions alone (test_edit_f7_h64_fp8_convs): relative L2 of the latents
    <= 6e-2 at every step -- measured 2.67e-2 for both masks against 2.7e-2 without the linears: their e4m3 noise does not show."""
    from oracle import sd15_torch as sd
    from gaussctrl_amd.sd.pipeline import DenoisePipeline
    from gaussctrl_amd.sd.weights import add_fp8_convs, add_fp8_linears
    z = np.load(os.path.join(GOLD, "fullgeom_edit_f7_h64.npz"))
    ref = z["lat_steps"]
    f, h, steps, seed = [int(v) for v in z["meta"][:4]]
    lat, disp, cn, cp = _inputs(f, h, seed)
    dt = torch.bfloat16
    uw, cw = nets(dt)
    uw, cw = dict(uw), dict(cw)
    import _nets
    add_fp8_convs(uw, _nets.conv_weights()[0], DEV)
    add_fp8_convs(cw, _nets.conv_weights()[1], DEV)
    add_fp8_linears(uw, which); add_fp8_linears(cw, which)
    pipe = DenoisePipeline(uw, cw, None, 20, 5.0)
    assert pipe.unet.fp8 and pipe.controlnet.fp8 and pipe.unet.fp8_lin == which and pipe.controlnet.fp8_lin == which
    trace = []
    pipe.edit_chunk(lat.to(DEV), disp.to(DEV), cn.to(DEV), cp.to(DEV), steps=steps,
                    on_step=lambda i, l: trace.append(l.permute(0, 3, 1, 2).float().cpu()))
    cur = _curve(trace, ref)
    print(f"\nedit f=7 h=64 fp8 convs + linears (mask {which}): rel L2 per step:\n  " + " ".join(f"{e:.2e}" for e in cur))
    within("max(cur)", max(cur), 6e-2)
    if which == 7:
        # ... and against the oracle that restates the fp8 path's ARITHMETIC (oracle/sd15_torch.py FP8_EMU: e4m3 roundings at the same sites;
        # tests/golden/make_fullgeom_golden.py edit7_e4m3, first 6 steps).  That oracle sits 2.08e-2 .. 2.66e-2 from the fp32 one -- the product's
        # 2.12e-2 .. 2.72e-2 is what e4m3 operands cost, not kernel error.  The two round the same tensors, but a value near an e4m3 boundary
        # rounds either way under a bf16-sized perturbation, so they are not expected to coincide; the bar here is the one the triangle
        # inequality guarantees (distance to fp32 of each < 3e-2); the recorded value is what a later round can tighten.
        emu = np.load(os.path.join(GOLD, "fullgeom_edit_f7_h64_e4m3.npz"))
        cur_e = [_rel(trace[i], torch.tensor(emu["lat_steps"][i])) for i in range(emu["lat_steps"].shape[0])]
        print("vs the e4m3-emulating oracle (its own distance to fp32: " + " ".join(f"{e:.2e}" for e in emu["rel_vs_fp32"]) + "):\n  " +
              " ".join(f"{e:.2e}" for e in cur_e))
        within("max(cur_e)", max(cur_e), 6e-2)
        # FALSIFIABLE form (as the bf16 / f16 tests above): the product's distance to the fp32 oracle must land ON the curve the two emulating
        # oracles predict -- e4m3 operands (FP8_EMU, 2.08e-2 .. 2.66e-2) and bf16 storage (ACT_ROUND, 3.9e-3 .. 5.0e-3) in quadrature -- within
        # 25 % at every one of the first 6 steps (measured round 4: within 2 %).
        pred = _e4m3_predicted()
        print("  predicted (e4m3 emulation (+) bf16 storage): " + " ".join(f"{e:.2e}" for e in pred))
        within("max_i |cur[i] / predicted[i] - 1|, steps 1..6", max(abs(cur[i] / pred[i] - 1.0) for i in range(len(pred))), 0.25)
    # the product path at the benchmark's grids: reference bank + a 3-view chunk (CFG batch 6 -- there the 16 x 16-map convolutions are part-
    # filled grids and run as k-sliced k_gemm8q + the split-K reduce kernel that leaves the GroupNorm partials; B = 14 above does not slice)
    bank = pipe.build_ref_bank(lat[:4].to(DEV), disp[:4].to(DEV), cn.to(DEV), cp.to(DEV))
    trace_c = []
    pipe.edit_chunk_cached(lat[4:].to(DEV), disp[4:].to(DEV), cn.to(DEV), cp.to(DEV), bank,
                           on_step=lambda i, l: trace_c.append(l.permute(0, 3, 1, 2).float().cpu()))
    cur_c = [_rel(t, torch.tensor(ref[i][4:])) for i, t in enumerate(trace_c)]
    print(f"cached-reference path, chunk frames vs oracle:\n  " + " ".join(f"{e:.2e}" for e in cur_c))
    within("max(cur_c)", max(cur_c), 6e-2)
