"""Generates the full-geometry denoise fixtures tests/golden/fullgeom_*.npz with the CPU fp32 oracle
(oracle/sd15_torch.py) at the geometry bench.py measures: 64x64 latents (512x512 images), SD1.5 /
sd-controlnet-depth / VAE shapes, f = 4 reference + 3 chunk frames (CFG batch 14), all 20 DDIM steps.

The oracle needs ~20 min of 8 host cores for the f=7 trajectory, so it is run ONCE here and the GPU box only
loads the .npz (data: per-step latents / decoded image; inputs and weights are re-created from the seeds below with
CPU torch generators, which are deterministic across machines).

Weights and the 2-byte inputs (disparity, text embeddings) are rounded to bf16 before the fp32 run.  bf16 values are
exactly representable in f16 inside f16's normal range (weights below 6e-5 in magnitude lose at most 3e-8 absolute),
so ONE fixture serves both the bf16 and the f16 GPU runs.

Attention uses oracle.sd15_torch.ATTN_IMPL = "sdpa" (torch's fused CPU kernel of softmax(q k^T s) v; the explicit
form would materialise 5 x [14*8, 4096, 4096] fp32 tensors per layer); tests/test_oracle_sd.py pins both forms
against the reference-generated goldens.

`config4` is BASELINE configs[3] end to end on the oracle: f = 4 references + chunk_size 8 = 12 frames (CFG batch 24), ALL 20 DDIM
steps, the 8 chunk frames decoded by the VAE and composited through the synthetic elliptical object mask
(gc_pipeline.py:209-234).  Stored: latents after steps 1, 2, 5, 10, 20 and the composited images on a stride-4 pixel lattice
(full-resolution decode parity has its own fixture, fullgeom_vae_h64.npz).

usage: python tests/golden/make_fullgeom_golden.py [edit7|vae|invert|edit12|config4|vaeenc|edit7_e4m3|edit7_actround|invert_actround ...]
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import sd15_torch as sd  # noqa: E402

SEED_UNET, SEED_CN, SEED_VAE = 100, 200, 300


def bf16r(x):
    return x.to(torch.bfloat16).float()


def inputs(f, h, seed):
    """same recipe as tests/test_denoise_model_gpu.py::_inputs"""
    g = torch.Generator().manual_seed(seed)
    lat = torch.randn(f, 4, h, h, generator=g)
    disp = torch.rand(f, 3, 8 * h, 8 * h, generator=g)
    cn = torch.randn(1, 77, 768, generator=g); cp = torch.randn(1, 77, 768, generator=g)
    return lat, disp, cn, cp


def weights():
    uw = {k: bf16r(v) for k, v in sd.make_unet_weights(sd.SD15, SEED_UNET).items()}
    cw = {k: bf16r(v) for k, v in sd.make_controlnet_weights(sd.SD15, SEED_CN).items()}
    return uw, cw


def edit(f, h, steps, seed, name):
    uw, cw = weights()
    lat, disp, cn, cp = inputs(f, h, seed)
    trace = []
    t0 = time.time()
    with torch.no_grad():
        sd.denoise_chunk(uw, cw, lat, bf16r(disp), bf16r(cn), bf16r(cp), 5.0, steps, sd.SD15, 20, trace=trace)
    np.savez_compressed(os.path.join(HERE, name), lat_steps=torch.stack(trace).numpy(),
                        meta=np.array([f, h, steps, seed, SEED_UNET, SEED_CN], np.int64))
    print(f"{name}: {time.time() - t0:.0f}s", flush=True)


def invert(f, h, steps, seed, name):
    """DDIM inversion: plain attention, guidance 0 -> no CFG batch (gc_pipeline.py:136-145)."""
    uw, cw = weights()
    lat, disp, cn, cp = inputs(f, h, seed)
    sch = sd.DDIM()
    x = lat.clone()
    ctx = bf16r(cp).expand(f, -1, -1)
    trace = []
    t0 = time.time()
    with torch.no_grad():
        for t in sch.timesteps(20, inverse=True)[:steps]:
            down, mid = sd.controlnet_forward(cw, x, t, ctx, bf16r(disp), sd.SD15, 1.0, "plain", 0.0)
            eps = sd.unet_forward(uw, x, t, ctx, down, mid, sd.SD15, "plain", 0.0)
            x = sch.inverse_step(eps, t, x, 20)
            trace.append(x.clone())
    np.savez_compressed(os.path.join(HERE, name), lat_steps=torch.stack(trace).numpy(),
                        meta=np.array([f, h, steps, seed, SEED_UNET, SEED_CN], np.int64))
    print(f"{name}: {time.time() - t0:.0f}s", flush=True)


def invert_actround(f, h, steps, seed, name, ref_name):
    """the inversion trajectory with bf16 / f16 activation storage restated on the oracle (sd.ACT_ROUND), all `steps` steps: curves only"""
    uw, cw = weights()
    lat, disp, cn, cp = inputs(f, h, seed)
    ref = np.load(os.path.join(HERE, ref_name))["lat_steps"]
    sch = sd.DDIM()
    ctx = bf16r(cp).expand(f, -1, -1)
    out = {}
    for dname, dt in (("bf16", torch.bfloat16), ("f16", torch.float16)):
        x = lat.clone()
        rel = []
        t0 = time.time()
        sd.ACT_ROUND = dt
        try:
            with torch.no_grad():
                for i, t in enumerate(sch.timesteps(20, inverse=True)[:steps]):
                    down, mid = sd.controlnet_forward(cw, x, t, ctx, bf16r(disp), sd.SD15, 1.0, "plain", 0.0)
                    eps = sd.unet_forward(uw, x, t, ctx, down, mid, sd.SD15, "plain", 0.0)
                    x = sch.inverse_step(eps, t, x, 20)
                    rel.append(float((x - torch.tensor(ref[i])).norm() / torch.tensor(ref[i]).norm()))
        finally:
            sd.ACT_ROUND = None
        out[dname] = np.array(rel)
        print(f"{name} [{dname} activations]: {time.time() - t0:.0f}s; relative L2 vs the fp32 oracle per step: " + " ".join(f"{e:.3e}" for e in rel), flush=True)
    np.savez_compressed(os.path.join(HERE, name), rel_bf16=out["bf16"], rel_f16=out["f16"], meta=np.array([f, h, steps, seed, SEED_UNET, SEED_CN], np.int64))


def vae(h, seed, name):
    vw = {k: bf16r(v) for k, v in sd.make_vae_decoder_weights(sd.VAE_SD, SEED_VAE).items()}
    z = torch.randn(1, 4, h, h, generator=torch.Generator().manual_seed(seed))
    t0 = time.time()
    with torch.no_grad():
        img = sd.postprocess_image(sd.vae_decode(vw, bf16r(z / 0.18215), sd.VAE_SD))
    np.savez_compressed(os.path.join(HERE, name), image=img.numpy(), meta=np.array([1, h, seed, SEED_VAE], np.int64))
    print(f"{name}: {time.time() - t0:.0f}s", flush=True)


def vaeenc(H, seed, name):
    """image2latent (gc_pipeline.py:239-246) at the full 512 x 512 image: vae.encode(2x-1).mean * 0.18215"""
    vw = {k: bf16r(v) for k, v in sd.make_vae_encoder_weights(sd.VAE_SD, SEED_VAE + 100).items()}
    img = torch.rand(H, H, 3, generator=torch.Generator().manual_seed(seed))
    t0 = time.time()
    with torch.no_grad():
        lat = sd.vae_encode_mean(vw, bf16r(img * 2 - 1).permute(2, 0, 1)[None], sd.VAE_SD) * 0.18215
    np.savez_compressed(os.path.join(HERE, name), latent=lat.numpy(), meta=np.array([H, seed, SEED_VAE + 100], np.int64))
    print(f"{name}: {time.time() - t0:.0f}s", flush=True)


CONFIG4_STEPS = (1, 2, 5, 10, 20)
CONFIG4_STRIDE = 4


def config4(h, seed, name):
    """BASELINE configs[3]: chunk_size 8 -> f = 12, all 20 steps, VAE decode of the chunk frames, mask composite."""
    from gaussctrl_amd import synthetic as syn          # numpy-only helper (the mask recipe bench.py --mask uses)
    f, steps = 12, 20
    uw, cw = weights()
    vw = {k: bf16r(v) for k, v in sd.make_vae_decoder_weights(sd.VAE_SD, SEED_VAE).items()}
    lat, disp, cn, cp = inputs(f, h, seed)
    trace = []
    t0 = time.time()
    with torch.no_grad():
        out = sd.denoise_chunk(uw, cw, lat, bf16r(disp), bf16r(cn), bf16r(cp), 5.0, steps, sd.SD15, 20, trace=trace)
        print(f"config4 denoise: {time.time() - t0:.0f}s", flush=True)
        H = 8 * h
        mask = torch.tensor(syn.elliptical_mask(H, H, soft=True))
        g = torch.Generator().manual_seed(seed + 1000)
        comps = []
        for j in range(4, f):                      # the references are dropped (gc_pipeline.py:219)
            img = sd.postprocess_image(sd.vae_decode(vw, bf16r(out[j:j + 1] / 0.18215), sd.VAE_SD))[0]      # [3,H,W]
            unedited = torch.rand(H, H, 3, generator=g)
            comps.append(sd.mask_composite(img, unedited, mask)[::CONFIG4_STRIDE, ::CONFIG4_STRIDE].clone())
    np.savez_compressed(os.path.join(HERE, name), lat_steps=torch.stack([trace[s - 1] for s in CONFIG4_STEPS]).numpy(),
                        which_steps=np.array(CONFIG4_STEPS, np.int64), composite=torch.stack(comps).numpy(),
                        meta=np.array([f, h, steps, seed, SEED_UNET, SEED_CN, SEED_VAE, CONFIG4_STRIDE], np.int64))
    print(f"{name}: {time.time() - t0:.0f}s", flush=True)


def edit_e4m3(f, h, steps, seed, name, ref_name):
    """The same trajectory with the ARITHMETIC of the product's fp8 path restated on the oracle (sd.FP8_EMU: e4m3 roundings at the sites
    gaussctrl_amd/sd/unet.py quantises -- resnet conv inputs on maps >= 16 x 16, LayerNorm outputs / GEGLU hidden / weights of the C = 640 /
    1280 transformer linears), first `steps` DDIM steps: what distance from the fp32 trajectory e4m3 operands by themselves produce."""
    uw, cw = weights()
    lat, disp, cn, cp = inputs(f, h, seed)
    trace = []
    t0 = time.time()
    sd.FP8_EMU = {"min_hw": 256, "min_rows": 1024, "linears": 7, "cache": {}}
    try:
        with torch.no_grad():
            sd.denoise_chunk(uw, cw, lat, bf16r(disp), bf16r(cn), bf16r(cp), 5.0, steps, sd.SD15, 20, trace=trace)
    finally:
        sd.FP8_EMU = None
    ref = np.load(os.path.join(HERE, ref_name))["lat_steps"]
    rel = [float((t - torch.tensor(ref[i])).norm() / torch.tensor(ref[i]).norm()) for i, t in enumerate(trace)]
    np.savez_compressed(os.path.join(HERE, name), lat_steps=torch.stack(trace).numpy().astype(np.float32), rel_vs_fp32=np.array(rel),
                        meta=np.array([f, h, steps, seed, SEED_UNET, SEED_CN], np.int64))
    print(f"{name}: {time.time() - t0:.0f}s; relative L2 of the e4m3-emulating oracle vs the fp32 oracle per step: " + " ".join(f"{e:.3e}" for e in rel), flush=True)


def edit_actround(f, h, steps, seed, name, ref_name):
    """What the 2-byte STORAGE of activations costs by itself: the same trajectory with sd.ACT_ROUND = bfloat16 / float16 (every tensor the
    product stores between kernels rounded to that type, arithmetic fp32), first `steps` DDIM steps, relative L2 vs the fp32 trajectory.
    The derivation of the parity bars of tests/test_fullgeom_gpu.py.  Stores the curves only."""
    uw, cw = weights()
    lat, disp, cn, cp = inputs(f, h, seed)
    ref = np.load(os.path.join(HERE, ref_name))["lat_steps"]
    out = {}
    for dname, dt in (("bf16", torch.bfloat16), ("f16", torch.float16)):
        trace = []
        t0 = time.time()
        sd.ACT_ROUND = dt
        try:
            with torch.no_grad():
                sd.denoise_chunk(uw, cw, lat, bf16r(disp), bf16r(cn), bf16r(cp), 5.0, steps, sd.SD15, 20, trace=trace)
        finally:
            sd.ACT_ROUND = None
        out[dname] = np.array([float((t - torch.tensor(ref[i])).norm() / torch.tensor(ref[i]).norm()) for i, t in enumerate(trace)])
        print(f"{name} [{dname} activations]: {time.time() - t0:.0f}s; relative L2 vs the fp32 oracle per step: " + " ".join(f"{e:.3e}" for e in out[dname]), flush=True)
    np.savez_compressed(os.path.join(HERE, name), rel_bf16=out["bf16"], rel_f16=out["f16"], meta=np.array([f, h, steps, seed, SEED_UNET, SEED_CN], np.int64))


if __name__ == "__main__":
    torch.set_num_threads(int(os.environ.get("GC_GOLDEN_THREADS", "7")))
    sd.ATTN_IMPL = "sdpa"
    what = sys.argv[1:] or ["vae", "edit7", "invert", "edit12"]
    for w in what:
        if w == "vae":
            vae(64, 1, "fullgeom_vae_h64.npz")
        elif w == "edit7":       # BASELINE configs[1]: chunk_size 3 -> f = 7, CFG batch 14, all 20 steps
            edit(7, 64, 20, 2, "fullgeom_edit_f7_h64.npz")
        elif w == "edit7_e4m3":  # the fp8 path's arithmetic restated on the oracle, first 6 steps of the edit7 trajectory
            edit_e4m3(7, 64, int(os.environ.get("GC_E4M3_STEPS", "6")), 2, "fullgeom_edit_f7_h64_e4m3.npz", "fullgeom_edit_f7_h64.npz")
        elif w == "edit7_e4m3_study":  # design study (prints only): e4m3 at MORE sites than the product has -- every map size, every channel count
            # (the C = 320 blocks too), attention outputs + out-projections -- how far would a wider fp8 path sit from fp32?
            uw, cw = weights()
            lat, disp, cn, cp = inputs(7, 64, 2)
            ref = np.load(os.path.join(HERE, "fullgeom_edit_f7_h64.npz"))["lat_steps"]
            for label, emu in (("convs on every map + linears at every C (bits 0-2)", dict(min_hw=1, min_rows=1, linears=7, any_c=True)),
                               ("... + attention outputs / out-projections (bit 3)", dict(min_hw=1, min_rows=1, linears=15, any_c=True))):
                trace = []
                sd.FP8_EMU = dict(emu, cache={})
                try:
                    with torch.no_grad():
                        sd.denoise_chunk(uw, cw, lat, bf16r(disp), bf16r(cn), bf16r(cp), 5.0, int(os.environ.get("GC_E4M3_STEPS", "3")), sd.SD15, 20, trace=trace)
                finally:
                    sd.FP8_EMU = None
                print(label + ": " + " ".join(f"{float((t - torch.tensor(ref[i])).norm() / torch.tensor(ref[i]).norm()):.3e}" for i, t in enumerate(trace)), flush=True)
        elif w == "invert_actround":  # the inversion trajectory with bf16 / f16 activation storage, all 20 steps
            invert_actround(3, 64, 20, 5, "fullgeom_invert_f3_h64_actround.npz", "fullgeom_invert_f3_h64.npz")
        elif w == "edit7_actround":  # the oracle with bf16 / f16 activation storage, first 6 steps of the edit7 trajectory
            edit_actround(7, 64, int(os.environ.get("GC_E4M3_STEPS", "6")), 2, "fullgeom_edit_f7_h64_actround.npz", "fullgeom_edit_f7_h64.npz")
        elif w == "invert":      # render_reverse's inversion, 3 views batched, all 20 steps
            invert(3, 64, 20, 5, "fullgeom_invert_f3_h64.npz")
        elif w == "edit12":      # BASELINE configs[3]: chunk_size 8 -> f = 12, CFG batch 24 (2 of 20 steps)
            edit(12, 64, 2, 7, "fullgeom_edit_f12_h64.npz")
        elif w == "vaeenc":      # image2latent at 512 x 512 (render_reverse's encoder call)
            vaeenc(512, 9, "fullgeom_vaeenc_h512.npz")
        elif w == "config4":     # BASELINE configs[3] end to end (same inputs as edit12: its 2 steps are this trajectory's first 2)
            config4(64, 7, "fullgeom_config4_f12_h64.npz")
