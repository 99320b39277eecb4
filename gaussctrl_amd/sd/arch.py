"""Parameter inventory (diffusers key names + shapes) of the three networks on the denoise path:
SD1.x UNet2DConditionModel, lllyasviel/sd-controlnet-depth ControlNetModel and the SD VAE decoder
(loaded by the reference at /root/reference/gaussctrl/gc_pipeline.py:100-102).

Used to (i) validate a real checkpoint's state dict and (ii) create seeded random weights of the exact
shapes directly on the GPU when no checkpoint is available (no network in the build environment)."""
from __future__ import annotations

import math

import torch

UNET_CFG = dict(block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, cross_dim=768, in_channels=4,
                out_channels=4, attn_levels=(True, True, True, False))
CONTROLNET_COND = (16, 32, 96, 256)
VAE_CFG = dict(block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=4)


def _conv(s, n, cin, cout, k): s[n + ".weight"] = (cout, cin, k, k); s[n + ".bias"] = (cout,)


def _lin(s, n, cin, cout, bias=True):
    s[n + ".weight"] = (cout, cin)
    if bias:
        s[n + ".bias"] = (cout,)


def _norm(s, n, c): s[n + ".weight"] = (c,); s[n + ".bias"] = (c,)


def _resnet(s, n, cin, cout, temb):
    _norm(s, n + ".norm1", cin); _conv(s, n + ".conv1", cin, cout, 3)
    if temb:
        _lin(s, n + ".time_emb_proj", temb, cout)
    _norm(s, n + ".norm2", cout); _conv(s, n + ".conv2", cout, cout, 3)
    if cin != cout:
        _conv(s, n + ".conv_shortcut", cin, cout, 1)


def _transformer(s, n, c, cross):
    _norm(s, n + ".norm", c); _conv(s, n + ".proj_in", c, c, 1)
    t = n + ".transformer_blocks.0"
    for k in ("norm1", "norm2", "norm3"):
        _norm(s, f"{t}.{k}", c)
    for a, kd in (("attn1", c), ("attn2", cross)):
        _lin(s, f"{t}.{a}.to_q", c, c, False); _lin(s, f"{t}.{a}.to_k", kd, c, False)
        _lin(s, f"{t}.{a}.to_v", kd, c, False); _lin(s, f"{t}.{a}.to_out.0", c, c)
    _lin(s, f"{t}.ff.net.0.proj", c, 8 * c); _lin(s, f"{t}.ff.net.2", 4 * c, c)
    _conv(s, n + ".proj_out", c, c, 1)


def _encoder(s, cfg):
    boc = cfg["block_out_channels"]; temb = 4 * boc[0]
    _conv(s, "conv_in", cfg["in_channels"], boc[0], 3)
    _lin(s, "time_embedding.linear_1", boc[0], temb); _lin(s, "time_embedding.linear_2", temb, temb)
    cin = boc[0]
    for i, cout in enumerate(boc):
        for j in range(cfg["layers_per_block"]):
            _resnet(s, f"down_blocks.{i}.resnets.{j}", cin, cout, temb)
            if cfg["attn_levels"][i]:
                _transformer(s, f"down_blocks.{i}.attentions.{j}", cout, cfg["cross_dim"])
            cin = cout
        if i < len(boc) - 1:
            _conv(s, f"down_blocks.{i}.downsamplers.0.conv", cout, cout, 3)
    _resnet(s, "mid_block.resnets.0", boc[-1], boc[-1], temb)
    _transformer(s, "mid_block.attentions.0", boc[-1], cfg["cross_dim"])
    _resnet(s, "mid_block.resnets.1", boc[-1], boc[-1], temb)


def skip_channels(cfg=UNET_CFG):
    boc = cfg["block_out_channels"]
    ch = [boc[0]]
    for i, c in enumerate(boc):
        ch += [c] * cfg["layers_per_block"]
        if i < len(boc) - 1:
            ch.append(c)
    return ch


def unet_shapes(cfg=UNET_CFG) -> dict:
    s = {}
    _encoder(s, cfg)
    boc = cfg["block_out_channels"]; temb = 4 * boc[0]
    skips = skip_channels(cfg)
    rev = list(reversed(boc)); rev_attn = list(reversed(cfg["attn_levels"]))
    prev = rev[0]
    for i, cout in enumerate(rev):
        for j in range(cfg["layers_per_block"] + 1):
            _resnet(s, f"up_blocks.{i}.resnets.{j}", prev + skips.pop(), cout, temb)
            if rev_attn[i]:
                _transformer(s, f"up_blocks.{i}.attentions.{j}", cout, cfg["cross_dim"])
            prev = cout
        if i < len(rev) - 1:
            _conv(s, f"up_blocks.{i}.upsamplers.0.conv", cout, cout, 3)
    _norm(s, "conv_norm_out", boc[0]); _conv(s, "conv_out", boc[0], cfg["out_channels"], 3)
    return s


def controlnet_shapes(cfg=UNET_CFG, cond=CONTROLNET_COND) -> dict:
    s = {}
    _encoder(s, cfg)
    boc = cfg["block_out_channels"]
    _conv(s, "controlnet_cond_embedding.conv_in", 3, cond[0], 3)
    k = 0
    for i in range(len(cond) - 1):
        _conv(s, f"controlnet_cond_embedding.blocks.{k}", cond[i], cond[i], 3); k += 1
        _conv(s, f"controlnet_cond_embedding.blocks.{k}", cond[i], cond[i + 1], 3); k += 1
    _conv(s, "controlnet_cond_embedding.conv_out", cond[-1], boc[0], 3)
    for n, c in enumerate(skip_channels(cfg)):
        _conv(s, f"controlnet_down_blocks.{n}", c, c, 1)
    _conv(s, "controlnet_mid_block", boc[-1], boc[-1], 1)
    return s


def vae_decoder_shapes(cfg=VAE_CFG) -> dict:
    s = {}
    boc = cfg["block_out_channels"]; lc = cfg["latent_channels"]
    _conv(s, "post_quant_conv", lc, lc, 1)
    _conv(s, "decoder.conv_in", lc, boc[-1], 3)
    _resnet(s, "decoder.mid_block.resnets.0", boc[-1], boc[-1], 0)
    a = "decoder.mid_block.attentions.0"
    _norm(s, a + ".group_norm", boc[-1])
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        _lin(s, f"{a}.{n}", boc[-1], boc[-1])
    _resnet(s, "decoder.mid_block.resnets.1", boc[-1], boc[-1], 0)
    rev = list(reversed(boc)); prev = rev[0]
    for i, cout in enumerate(rev):
        for j in range(cfg["layers_per_block"] + 1):
            _resnet(s, f"decoder.up_blocks.{i}.resnets.{j}", prev, cout, 0); prev = cout
        if i < len(rev) - 1:
            _conv(s, f"decoder.up_blocks.{i}.upsamplers.0.conv", cout, cout, 3)
    _norm(s, "decoder.conv_norm_out", boc[0]); _conv(s, "decoder.conv_out", boc[0], 3, 3)
    return s


def vae_encoder_shapes(cfg=VAE_CFG) -> dict:
    s = {}
    boc = cfg["block_out_channels"]; lc = cfg["latent_channels"]
    _conv(s, "encoder.conv_in", 3, boc[0], 3)
    prev = boc[0]
    for i, cout in enumerate(boc):
        for j in range(cfg["layers_per_block"]):
            _resnet(s, f"encoder.down_blocks.{i}.resnets.{j}", prev, cout, 0); prev = cout
        if i < len(boc) - 1:
            _conv(s, f"encoder.down_blocks.{i}.downsamplers.0.conv", cout, cout, 3)
    _resnet(s, "encoder.mid_block.resnets.0", boc[-1], boc[-1], 0)
    a = "encoder.mid_block.attentions.0"
    _norm(s, a + ".group_norm", boc[-1])
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        _lin(s, f"{a}.{n}", boc[-1], boc[-1])
    _resnet(s, "encoder.mid_block.resnets.1", boc[-1], boc[-1], 0)
    _norm(s, "encoder.conv_norm_out", boc[-1]); _conv(s, "encoder.conv_out", boc[-1], 2 * lc, 3)
    _conv(s, "quant_conv", 2 * lc, 2 * lc, 1)
    return s


def random_state_dict(shapes: dict, seed: int, device, zero_conv_std: float = 0.02) -> dict:
    """PyTorch-default-like init (uniform +-1/sqrt(fan_in)); norm affine ~ (1, 0) + noise; ControlNet zero-convs
    N(0, 0.02^2) so residuals are non-trivial.  fp32 tensors on `device`."""
    g = torch.Generator(device=device).manual_seed(seed)
    out = {}
    for k, shp in shapes.items():
        zero = k.startswith("controlnet_down_blocks") or k.startswith("controlnet_mid_block") or k.startswith("controlnet_cond_embedding.conv_out")
        if k.endswith(".weight") and len(shp) == 1:
            out[k] = 1.0 + 0.1 * torch.randn(shp, generator=g, device=device)
        elif k.endswith(".bias") and (k[:-5] + ".weight") in shapes and len(shapes[k[:-5] + ".weight"]) == 1:
            out[k] = 0.1 * torch.randn(shp, generator=g, device=device)
        elif zero:
            out[k] = zero_conv_std * torch.randn(shp, generator=g, device=device)
        else:
            wshape = shapes[k] if k.endswith(".weight") else shapes[k[:-5] + ".weight"]
            fan = math.prod(wshape[1:])
            out[k] = (torch.rand(shp, generator=g, device=device) * 2 - 1) / math.sqrt(fan)
    return out


def check_state_dict(sd: dict, shapes: dict) -> None:
    missing = [k for k in shapes if k not in sd]
    bad = [k for k in shapes if k in sd and tuple(sd[k].shape) != tuple(shapes[k])]
    if missing or bad:
        raise ValueError(f"state dict mismatch: missing {missing[:5]} wrong-shape {bad[:5]}")
