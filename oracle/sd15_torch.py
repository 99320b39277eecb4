"""oracle/sd15_torch.py -- TEST INFRASTRUCTURE ONLY (CPU oracle of the denoise half of the hot path).

Plain PyTorch (fp32 by default) restatement of what the reference executes through
diffusers==0.26.0 (third-party, pinned at /root/reference/requirements.txt:2, NOT under
/root/reference and not installable here) when GaussCtrlPipeline.edit_images / render_reverse call
`self.pipe(...)` (/root/reference/gaussctrl/gc_pipeline.py:142-145,209-219):

  * UNet2DConditionModel (SD1.x config) and ControlNetModel (sd-controlnet-depth config) forward,
    SURVEY.md Appendix B;
  * the attention layers as the reference's CrossViewAttnProcessor computes them
    (/root/reference/gaussctrl/utils.py:25-133) -- THIS part is pinned: tests/golden/xview_attn_*.npz
    were produced by importing utils.py itself (tests/golden/make_xview_golden.py);
  * DDIM / inverse-DDIM step and the pipeline's CFG arithmetic, SURVEY.md Appendix C;
  * AutoencoderKL decoder (vae.decode inside pipe(output_type='pt')).

PARITY UNPINNED for everything except the attention processor: diffusers and the SD1.5/ControlNet
weights are unavailable, so architecture semantics follow the published configs from memory
([recall] in SURVEY.md) with seeded random weights of the exact shapes.  State-dict keys follow the
diffusers naming so real checkpoints can be dropped in.

Two emulation switches (both None = off = the plain fp32 oracle, bit for bit) restate the PRODUCT's roundings on top of this arithmetic,
to tell what a storage / operand type costs by itself: ACT_ROUND (bf16 / f16 storage of every tensor the product keeps between kernels) and
FP8_EMU (OCP e4m3 operands at the sites of the product's fp8 path).  tests/golden/make_fullgeom_golden.py edit7_actround / invert_actround /
edit7_e4m3 ran them at the benchmark geometry; DESIGN.md 2 sets the results beside the product's measured distances.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------- e4m3 emulation (BASELINE configs[3])
# Restatement of the ARITHMETIC of the product's fp8 path (gaussctrl_amd/sd/unet.py with weights.add_fp8_convs / add_fp8_linears; DESIGN.md
# 3.3) on top of this fp32 oracle: the same tensors are rounded to OCP e4m3 at the same sites -- resnet conv1 / conv2 inputs (GroupNorm + SiLU
# outputs, tensor-wide scale 2^0, saturation at +-448) on maps of >= min_hw pixels; in the transformer blocks with C % 128 == 0 and >=
# min_rows token rows the LayerNorm outputs that feed Q | K | V (bit 2), attn2.to_q (bit 1) and the GEGLU projection (bit 0) and the GEGLU
# hidden that feeds the down projection; weights per OUTPUT ROW with a power-of-two scale that puts the row maximum in e4m3's top binade --
# and multiplied exactly (fp32 here; the block-scaled MFMA accumulates in fp32).  It answers one question: how far from the fp32 oracle
# SHOULD latents computed with these roundings be?  (tests/golden/make_fullgeom_golden.py edit7_e4m3; the product's measured distance is compared
# with it in DESIGN.md 3.3.)  FP8_EMU = None: off (every other use of this file).
FP8_EMU = None          # or {"min_hw": 256, "min_rows": 1024, "linears": 7, "cache": {}}


def _q8(x):
    return x.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(x.dtype)


# --------------------------------------------------------------------------------------- 2-byte activation emulation
# ACT_ROUND = torch.bfloat16 / torch.float16: every tensor the product STORES in its activation type between kernels is rounded to that type
# here (conv / linear outputs after their fused bias / time-embedding / residual epilogue, GroupNorm and LayerNorm outputs, Q / K / V,
# attention outputs, the GEGLU hidden, the ControlNet residual sums, the 2-byte network input); weights are whatever the caller passes (the
# fixtures round them to bf16), arithmetic inside an op stays fp32 as on the MFMA.  Like FP8_EMU it answers "what does the storage type cost by
# itself" -- the derivation of the bf16 / f16 parity bars.  Not modelled: the softmax probabilities rounded for the P V MFMA, and
# the level-0 blocks' row-resident kernels, which keep MORE in fp32 registers than this does.  None: off.
ACT_ROUND = None


def _r(x):
    return x if ACT_ROUND is None else x.to(ACT_ROUND).to(torch.float32)


def _q8_rows(w, key):
    """weight [N, ...] -> its e4m3 rounding with one power-of-two scale per output row (gaussctrl_amd/sd/weights.py::quantize_rows_e4m3)"""
    c = FP8_EMU["cache"]
    if key not in c:
        w2 = w.reshape(w.shape[0], -1).float()
        e = torch.floor(torch.log2(448.0 / w2.abs().amax(dim=1).clamp_min(1e-30))).clamp(-100, 100)
        sc = torch.exp2(e)[:, None]
        c[key] = ((w2 * sc).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float() / sc).reshape(w.shape).to(w.dtype)
    return c[key]

SD15 = dict(block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, heads=8, cross_dim=768,
            in_channels=4, out_channels=4, groups=32, attn_levels=(True, True, True, False),
            cond_channels=(16, 32, 96, 256), text_len=77)
# narrow variant with the same topology for fast CPU tests
TINY = dict(block_out_channels=(32, 64, 128, 128), layers_per_block=2, heads=2, cross_dim=64,
            in_channels=4, out_channels=4, groups=8, attn_levels=(True, True, True, False),
            cond_channels=(8, 8, 16, 32), text_len=7)
VAE_SD = dict(block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=4, groups=32)
VAE_TINY = dict(block_out_channels=(16, 32, 64, 64), layers_per_block=2, latent_channels=4, groups=8)


# =========================================================================================== weights
class _Init:
    def __init__(self, seed):
        self.g = torch.Generator().manual_seed(seed)
        self.w = {}

    def conv(self, name, cin, cout, k, zero=False, std=None):
        fan = cin * k * k
        bound = 1.0 / math.sqrt(fan)
        if zero:   # "zero convs" of ControlNet, re-initialised N(0, 0.02^2) so residuals are non-trivial
            self.w[name + ".weight"] = torch.randn(cout, cin, k, k, generator=self.g) * 0.02
            self.w[name + ".bias"] = torch.randn(cout, generator=self.g) * 0.02
        else:
            self.w[name + ".weight"] = (torch.rand(cout, cin, k, k, generator=self.g) * 2 - 1) * bound
            self.w[name + ".bias"] = (torch.rand(cout, generator=self.g) * 2 - 1) * bound

    def linear(self, name, cin, cout, bias=True):
        bound = 1.0 / math.sqrt(cin)
        self.w[name + ".weight"] = (torch.rand(cout, cin, generator=self.g) * 2 - 1) * bound
        if bias:
            self.w[name + ".bias"] = (torch.rand(cout, generator=self.g) * 2 - 1) * bound

    def norm(self, name, c):
        self.w[name + ".weight"] = 1.0 + 0.1 * torch.randn(c, generator=self.g)
        self.w[name + ".bias"] = 0.1 * torch.randn(c, generator=self.g)

    def resnet(self, name, cin, cout, temb):
        self.norm(name + ".norm1", cin); self.conv(name + ".conv1", cin, cout, 3)
        if temb:
            self.linear(name + ".time_emb_proj", temb, cout)
        self.norm(name + ".norm2", cout); self.conv(name + ".conv2", cout, cout, 3)
        if cin != cout:
            self.conv(name + ".conv_shortcut", cin, cout, 1)

    def transformer(self, name, c, cross):
        self.norm(name + ".norm", c); self.conv(name + ".proj_in", c, c, 1)
        t = name + ".transformer_blocks.0"
        for n in ("norm1", "norm2", "norm3"):
            self.norm(f"{t}.{n}", c)
        for a, kd in (("attn1", c), ("attn2", cross)):
            self.linear(f"{t}.{a}.to_q", c, c, bias=False); self.linear(f"{t}.{a}.to_k", kd, c, bias=False)
            self.linear(f"{t}.{a}.to_v", kd, c, bias=False); self.linear(f"{t}.{a}.to_out.0", c, c)
        self.linear(f"{t}.ff.net.0.proj", c, 8 * c); self.linear(f"{t}.ff.net.2", 4 * c, c)
        self.conv(name + ".proj_out", c, c, 1)


def _encoder_weights(I: _Init, cfg):
    boc = cfg["block_out_channels"]; temb = 4 * boc[0]
    I.conv("conv_in", cfg["in_channels"], boc[0], 3)
    I.linear("time_embedding.linear_1", boc[0], temb); I.linear("time_embedding.linear_2", temb, temb)
    cin = boc[0]
    for i, cout in enumerate(boc):
        for j in range(cfg["layers_per_block"]):
            I.resnet(f"down_blocks.{i}.resnets.{j}", cin, cout, temb)
            if cfg["attn_levels"][i]:
                I.transformer(f"down_blocks.{i}.attentions.{j}", cout, cfg["cross_dim"])
            cin = cout
        if i < len(boc) - 1:
            I.conv(f"down_blocks.{i}.downsamplers.0.conv", cout, cout, 3)
    I.resnet("mid_block.resnets.0", boc[-1], boc[-1], temb)
    I.transformer("mid_block.attentions.0", boc[-1], cfg["cross_dim"])
    I.resnet("mid_block.resnets.1", boc[-1], boc[-1], temb)


def skip_channels(cfg):
    boc = cfg["block_out_channels"]
    ch = [boc[0]]
    for i, c in enumerate(boc):
        ch += [c] * cfg["layers_per_block"]
        if i < len(boc) - 1:
            ch.append(c)
    return ch


def make_unet_weights(cfg=SD15, seed=100):
    I = _Init(seed)
    _encoder_weights(I, cfg)
    boc = cfg["block_out_channels"]; temb = 4 * boc[0]
    skips = skip_channels(cfg)
    rev = list(reversed(boc))
    prev = rev[0]
    for i, cout in enumerate(rev):
        for j in range(cfg["layers_per_block"] + 1):
            sk = skips.pop()
            I.resnet(f"up_blocks.{i}.resnets.{j}", prev + sk, cout, temb)
            if list(reversed(cfg["attn_levels"]))[i]:
                I.transformer(f"up_blocks.{i}.attentions.{j}", cout, cfg["cross_dim"])
            prev = cout
        if i < len(rev) - 1:
            I.conv(f"up_blocks.{i}.upsamplers.0.conv", cout, cout, 3)
    I.norm("conv_norm_out", boc[0]); I.conv("conv_out", boc[0], cfg["out_channels"], 3)
    return I.w


def make_controlnet_weights(cfg=SD15, seed=200):
    I = _Init(seed)
    _encoder_weights(I, cfg)
    cc = cfg["cond_channels"]; boc = cfg["block_out_channels"]
    I.conv("controlnet_cond_embedding.conv_in", 3, cc[0], 3)
    k = 0
    for i in range(len(cc) - 1):
        I.conv(f"controlnet_cond_embedding.blocks.{k}", cc[i], cc[i], 3); k += 1
        I.conv(f"controlnet_cond_embedding.blocks.{k}", cc[i], cc[i + 1], 3); k += 1
    I.conv("controlnet_cond_embedding.conv_out", cc[-1], boc[0], 3, zero=True)
    for n, c in enumerate(skip_channels(cfg)):
        I.conv(f"controlnet_down_blocks.{n}", c, c, 1, zero=True)
    I.conv("controlnet_mid_block", boc[-1], boc[-1], 1, zero=True)
    return I.w


def make_vae_decoder_weights(cfg=VAE_SD, seed=300):
    I = _Init(seed)
    boc = cfg["block_out_channels"]; lc = cfg["latent_channels"]
    I.conv("post_quant_conv", lc, lc, 1)
    I.conv("decoder.conv_in", lc, boc[-1], 3)
    I.resnet("decoder.mid_block.resnets.0", boc[-1], boc[-1], 0)
    a = "decoder.mid_block.attentions.0"
    I.norm(a + ".group_norm", boc[-1])
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        I.linear(f"{a}.{n}", boc[-1], boc[-1])
    I.resnet("decoder.mid_block.resnets.1", boc[-1], boc[-1], 0)
    rev = list(reversed(boc)); prev = rev[0]
    for i, cout in enumerate(rev):
        for j in range(cfg["layers_per_block"] + 1):
            I.resnet(f"decoder.up_blocks.{i}.resnets.{j}", prev, cout, 0); prev = cout
        if i < len(rev) - 1:
            I.conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", cout, cout, 3)
    I.norm("decoder.conv_norm_out", boc[0]); I.conv("decoder.conv_out", boc[0], 3, 3)
    return I.w


# =========================================================================================== attention
def _heads(t, h):
    b, l, c = t.shape
    return t.reshape(b, l, h, c // h).permute(0, 2, 1, 3)          # [B,H,L,D]


ATTN_IMPL = "explicit"      # "explicit": materialised softmax(q k^T s) v, the form the reference executes (utils.py:25-37);
                            # "sdpa": torch's fused CPU kernel of the same function, used only by the full-geometry fixture
                            # generator (tests/golden/make_fullgeom_golden.py) where L = 4096 makes the explicit form take
                            # hours; tests/test_oracle_sd.py checks the two agree on the reference-generated goldens.


def plain_attention(q, k, v, heads):
    """softmax(q k^T / sqrt(d)) v -- diffusers AttnProcessor / get_attention_scores."""
    qh, kh, vh = _heads(q, heads), _heads(k, heads), _heads(v, heads)
    if ATTN_IMPL == "sdpa":
        o = F.scaled_dot_product_attention(qh, kh, vh)
    else:
        s = (qh @ kh.transpose(-1, -2)) * (qh.shape[-1] ** -0.5)
        o = s.softmax(-1) @ vh
    b, h, l, d = o.shape
    return o.permute(0, 2, 1, 3).reshape(b, l, h * d)


def cross_view_attention(q, k, v, heads, self_attn_coeff, unet_chunk_size=2, num_refs=4):
    """utils.py:86-117: a*self + (1-a)*mean_r attn(Q, K_ref_r, V_ref_r); the r-th reference is frame r of the
    SAME chunk (CFG half) broadcast over the frames of that half (compute_attn, utils.py:25-37)."""
    B = k.shape[0]
    f = B // unet_chunk_size                                             # video_length, utils.py:94
    out_self = plain_attention(q, k, v, heads)
    acc = 0
    for r in range(num_refs):                                            # exactly 4 refs are hard-wired, utils.py:95-102
        idx = torch.arange(B) // f * f + r
        acc = acc + plain_attention(q, k[idx], v[idx], heads)
    return self_attn_coeff * out_self + (1 - self_attn_coeff) * (acc / num_refs)


def attention_layer(w, p, x, ctx, heads, mode, coeff, q8=0):
    """One CrossViewAttnProcessor.__call__ (utils.py:44-133) for a [B,L,C] input; mode in {"plain","xview"}.
    Text cross-attention (ctx given) is ordinary attention in both modes (utils.py:111-117)."""
    wq, wk, wv = w[p + ".to_q.weight"], w[p + ".to_k.weight"], w[p + ".to_v.weight"]
    xq = x
    if q8 & (4 if ctx is None else 2):       # e4m3 emulation: the LayerNorm output and the projection weights it meets
        xq = _q8(x)
        wq = _q8_rows(wq, (id(w), p + ".to_q"))
        if ctx is None:
            wk, wv = _q8_rows(wk, (id(w), p + ".to_k")), _q8_rows(wv, (id(w), p + ".to_v"))
    q = _r(xq @ wq.T)
    src = xq if ctx is None else ctx
    k = _r(src @ wk.T)
    v = _r(src @ wv.T)
    if ctx is None and mode == "xview":
        o = cross_view_attention(q, k, v, heads, coeff)
    else:
        o = plain_attention(q, k, v, heads)
    wo = w[p + ".to_out.0.weight"]
    if q8 & 8:                               # (design study, not a product site: the attention output and the out-projection in e4m3)
        o, wo = _q8(o), _q8_rows(wo, (id(w), p + ".to_out"))
    return _r(o) @ wo.T + w[p + ".to_out.0.bias"]        # (the caller adds the residual, then rounds)


# =========================================================================================== blocks
def timestep_embedding(t, dim):
    """diffusers get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0)."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half
    emb = t.to(torch.float32)[:, None] * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


def _gn(w, p, x, groups, eps):
    return F.group_norm(x, groups, w[p + ".weight"], w[p + ".bias"], eps)


def _conv(w, p, x, stride=1, pad=None):
    k = w[p + ".weight"].shape[-1]
    return F.conv2d(x, w[p + ".weight"], w[p + ".bias"], stride=stride, padding=(k // 2 if pad is None else pad))


def _conv8(w, p, x):
    """3x3 conv of the e4m3-rounded input with the row-wise e4m3-rounded weights (e4m3 emulation of a resnet convolution)"""
    return F.conv2d(_q8(x), _q8_rows(w[p + ".weight"], (id(w), p)), w[p + ".bias"], padding=1)


def resnet(w, p, x, temb, groups, eps=1e-5):
    q8 = FP8_EMU is not None and x.shape[2] * x.shape[3] >= FP8_EMU["min_hw"]
    conv = _conv8 if q8 else _conv
    h = conv(w, p + ".conv1", _r(F.silu(_gn(w, p + ".norm1", x, groups, eps))))
    if temb is not None:
        h = h + (F.silu(temb) @ w[p + ".time_emb_proj.weight"].T + w[p + ".time_emb_proj.bias"])[:, :, None, None]
    h = conv(w, p + ".conv2", _r(F.silu(_gn(w, p + ".norm2", _r(h), groups, eps))))
    if (p + ".conv_shortcut.weight") in w:
        x = _r(_conv(w, p + ".conv_shortcut", x))
    return _r(x + h)


def transformer(w, p, x, ctx, cfg, mode, coeff):
    B, C, H, W = x.shape
    res = x
    h = _r(_conv(w, p + ".proj_in", _r(_gn(w, p + ".norm", x, cfg["groups"], 1e-6))))
    h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    t = p + ".transformer_blocks.0"
    ln = lambda n, z: _r(F.layer_norm(z, (C,), w[f"{t}.{n}.weight"], w[f"{t}.{n}.bias"], 1e-5))
    q8 = FP8_EMU["linears"] if (FP8_EMU is not None and (C % 128 == 0 or FP8_EMU.get("any_c")) and B * H * W >= FP8_EMU["min_rows"]) else 0
    h = _r(attention_layer(w, t + ".attn1", ln("norm1", h), None, cfg["heads"], mode, coeff, q8) + h)
    h = _r(attention_layer(w, t + ".attn2", ln("norm2", h), ctx, cfg["heads"], mode, coeff, q8) + h)
    n3 = ln("norm3", h)
    w1, w2 = w[t + ".ff.net.0.proj.weight"], w[t + ".ff.net.2.weight"]
    if q8 & 1:                               # e4m3 emulation: LayerNorm output, both FF weights, and the GEGLU hidden between them
        n3, w1, w2 = _q8(n3), _q8_rows(w1, (id(w), t + ".ff1")), _q8_rows(w2, (id(w), t + ".ff2"))
    pr = n3 @ w1.T + w[t + ".ff.net.0.proj.bias"]
    hid, gate = pr.chunk(2, dim=-1)
    gg = _r(hid * F.gelu(gate))
    h = _r((_q8(gg) if q8 & 1 else gg) @ w2.T + w[t + ".ff.net.2.bias"] + h)
    h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
    return _r(_conv(w, p + ".proj_out", h) + res)


def _time_embed(w, t, B, cfg):
    t = torch.as_tensor(t, dtype=torch.float32).reshape(-1).expand(B) if not (torch.is_tensor(t) and t.numel() == B) else t
    e = timestep_embedding(t, cfg["block_out_channels"][0])
    e = F.silu(e @ w["time_embedding.linear_1.weight"].T + w["time_embedding.linear_1.bias"])
    return e @ w["time_embedding.linear_2.weight"].T + w["time_embedding.linear_2.bias"]


def _encoder(w, x, temb, ctx, cfg, mode, coeff):
    skips = [x]
    boc = cfg["block_out_channels"]
    for i in range(len(boc)):
        for j in range(cfg["layers_per_block"]):
            x = resnet(w, f"down_blocks.{i}.resnets.{j}", x, temb, cfg["groups"])
            if cfg["attn_levels"][i]:
                x = transformer(w, f"down_blocks.{i}.attentions.{j}", x, ctx, cfg, mode, coeff)
            skips.append(x)
        if i < len(boc) - 1:
            x = _r(_conv(w, f"down_blocks.{i}.downsamplers.0.conv", x, stride=2, pad=1))
            skips.append(x)
    x = resnet(w, "mid_block.resnets.0", x, temb, cfg["groups"])
    x = transformer(w, "mid_block.attentions.0", x, ctx, cfg, mode, coeff)
    x = resnet(w, "mid_block.resnets.1", x, temb, cfg["groups"])
    return x, skips


def controlnet_forward(w, sample, t, ctx, cond, cfg=SD15, conditioning_scale=1.0, mode="xview", coeff=0.0):
    """ControlNetModel.forward -> (12 down residuals, mid residual).  coeff=0: gc_pipeline.py:166-168."""
    temb = _time_embed(w, t, sample.shape[0], cfg)
    c = _r(F.silu(_conv(w, "controlnet_cond_embedding.conv_in", cond)))
    nblk = 2 * (len(cfg["cond_channels"]) - 1)
    for k in range(nblk):
        c = _r(F.silu(_conv(w, f"controlnet_cond_embedding.blocks.{k}", c, stride=2 if k % 2 == 1 else 1, pad=1)))
    c = _r(_conv(w, "controlnet_cond_embedding.conv_out", c))
    x = _r(_conv(w, "conv_in", _r(sample)) + c)
    x, skips = _encoder(w, x, temb, ctx, cfg, mode, coeff)
    down = [_r(_conv(w, f"controlnet_down_blocks.{n}", s) * conditioning_scale) for n, s in enumerate(skips)]
    mid = _r(_conv(w, "controlnet_mid_block", x) * conditioning_scale)
    return down, mid


def unet_forward(w, sample, t, ctx, down_res=None, mid_res=None, cfg=SD15, mode="xview", coeff=0.6):
    """UNet2DConditionModel.forward with ControlNet residuals.  coeff=0.6: gc_pipeline.py:163-165."""
    temb = _time_embed(w, t, sample.shape[0], cfg)
    x = _r(_conv(w, "conv_in", _r(sample)))
    x, skips = _encoder(w, x, temb, ctx, cfg, mode, coeff)
    if down_res is not None:
        skips = [_r(s + r) for s, r in zip(skips, down_res)]
    if mid_res is not None:
        x = _r(x + mid_res)
    boc = cfg["block_out_channels"]
    rev_attn = list(reversed(cfg["attn_levels"]))
    for i in range(len(boc)):
        for j in range(cfg["layers_per_block"] + 1):
            x = torch.cat([x, skips.pop()], dim=1)
            x = resnet(w, f"up_blocks.{i}.resnets.{j}", x, temb, cfg["groups"])
            if rev_attn[i]:
                x = transformer(w, f"up_blocks.{i}.attentions.{j}", x, ctx, cfg, mode, coeff)
        if i < len(boc) - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = _r(_conv(w, f"up_blocks.{i}.upsamplers.0.conv", x))
    x = _r(F.silu(_gn(w, "conv_norm_out", x, cfg["groups"], 1e-5)))
    return _conv(w, "conv_out", x)


def vae_decode(w, z, cfg=VAE_SD):
    """AutoencoderKL.decode(z) (z already divided by 0.18215 by the caller) -> image in [-1,1]."""
    g = cfg["groups"]
    x = _r(_conv(w, "post_quant_conv", _r(z)))              # (_r: ACT_ROUND emulation of the product's 2-byte storage, identity when off)
    x = _r(_conv(w, "decoder.conv_in", x))
    x = resnet(w, "decoder.mid_block.resnets.0", x, None, g, 1e-6)
    a = "decoder.mid_block.attentions.0"
    B, C, H, W = x.shape
    h = _r(_gn(w, a + ".group_norm", x, g, 1e-6)).reshape(B, C, H * W).transpose(1, 2)
    q = _r(h @ w[a + ".to_q.weight"].T + w[a + ".to_q.bias"])
    k = _r(h @ w[a + ".to_k.weight"].T + w[a + ".to_k.bias"])
    v = _r(h @ w[a + ".to_v.weight"].T + w[a + ".to_v.bias"])
    o = _r(plain_attention(q, k, v, 1)) @ w[a + ".to_out.0.weight"].T + w[a + ".to_out.0.bias"]
    x = _r(x + o.transpose(1, 2).reshape(B, C, H, W))
    x = resnet(w, "decoder.mid_block.resnets.1", x, None, g, 1e-6)
    n = len(cfg["block_out_channels"])
    for i in range(n):
        for j in range(cfg["layers_per_block"] + 1):
            x = resnet(w, f"decoder.up_blocks.{i}.resnets.{j}", x, None, g, 1e-6)
        if i < n - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = _r(_conv(w, f"decoder.up_blocks.{i}.upsamplers.0.conv", x))
    x = _r(F.silu(_gn(w, "decoder.conv_norm_out", x, g, 1e-6)))
    return _conv(w, "decoder.conv_out", x)


def make_vae_encoder_weights(cfg=VAE_SD, seed=400):
    I = _Init(seed)
    boc = cfg["block_out_channels"]; lc = cfg["latent_channels"]
    I.conv("encoder.conv_in", 3, boc[0], 3)
    prev = boc[0]
    for i, cout in enumerate(boc):
        for j in range(cfg["layers_per_block"]):
            I.resnet(f"encoder.down_blocks.{i}.resnets.{j}", prev, cout, 0); prev = cout
        if i < len(boc) - 1:
            I.conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", cout, cout, 3)
    I.resnet("encoder.mid_block.resnets.0", boc[-1], boc[-1], 0)
    a = "encoder.mid_block.attentions.0"
    I.norm(a + ".group_norm", boc[-1])
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        I.linear(f"{a}.{n}", boc[-1], boc[-1])
    I.resnet("encoder.mid_block.resnets.1", boc[-1], boc[-1], 0)
    I.norm("encoder.conv_norm_out", boc[-1]); I.conv("encoder.conv_out", boc[-1], 2 * lc, 3)
    I.conv("quant_conv", 2 * lc, 2 * lc, 1)
    return I.w


def vae_encode_mean(w, img, cfg=VAE_SD):
    """AutoencoderKL.encode(img)['latent_dist'].mean (gc_pipeline.py:244) for img in [-1,1], [B,3,H,W]."""
    g = cfg["groups"]
    x = _conv(w, "encoder.conv_in", img)
    n = len(cfg["block_out_channels"])
    for i in range(n):
        for j in range(cfg["layers_per_block"]):
            x = resnet(w, f"encoder.down_blocks.{i}.resnets.{j}", x, None, g, 1e-6)
        if i < n - 1:
            x = F.pad(x, (0, 1, 0, 1))                       # diffusers Downsample2D(padding=0) of the VAE encoder
            x = _conv(w, f"encoder.down_blocks.{i}.downsamplers.0.conv", x, stride=2, pad=0)
    x = resnet(w, "encoder.mid_block.resnets.0", x, None, g, 1e-6)
    a = "encoder.mid_block.attentions.0"
    B, C, H, W = x.shape
    h = _gn(w, a + ".group_norm", x, g, 1e-6).reshape(B, C, H * W).transpose(1, 2)
    q = h @ w[a + ".to_q.weight"].T + w[a + ".to_q.bias"]
    k = h @ w[a + ".to_k.weight"].T + w[a + ".to_k.bias"]
    v = h @ w[a + ".to_v.weight"].T + w[a + ".to_v.bias"]
    o = plain_attention(q, k, v, 1) @ w[a + ".to_out.0.weight"].T + w[a + ".to_out.0.bias"]
    x = x + o.transpose(1, 2).reshape(B, C, H, W)
    x = resnet(w, "encoder.mid_block.resnets.1", x, None, g, 1e-6)
    x = F.silu(_gn(w, "encoder.conv_norm_out", x, g, 1e-6))
    moments = _conv(w, "quant_conv", _conv(w, "encoder.conv_out", x))
    return moments[:, : cfg["latent_channels"]]


# =========================================================================================== scheduler
class DDIM:
    """DDIMScheduler / DDIMInverseScheduler arithmetic for SD1.x scheduler_config:
    scaled_linear betas 0.00085..0.012 over 1000, steps_offset 1, set_alpha_to_one False, leading spacing,
    eta 0, epsilon prediction, no clipping."""

    def __init__(self, num_train=1000):
        betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, num_train, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha = self.alphas_cumprod[0]
        self.num_train = num_train

    def timesteps(self, n, inverse=False):
        ratio = self.num_train // n
        ts = (torch.arange(0, n) * ratio).round().to(torch.int64) + 1          # leading + steps_offset 1
        return ts if inverse else ts.flip(0)

    def step(self, eps, t, x, n):
        """x_t -> x_{t-ratio}"""
        ratio = self.num_train // n
        prev = int(t) - ratio
        a_t = self.alphas_cumprod[int(t)]
        a_p = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha
        x0 = (x - (1 - a_t).sqrt() * eps) / a_t.sqrt()
        return a_p.sqrt() * x0 + (1 - a_p).sqrt() * eps

    def inverse_step(self, eps, t, x, n):
        """DDIMInverseScheduler.step: alpha_prod_t at t - ratio (initial value if negative), next at t."""
        ratio = self.num_train // n
        prev = int(t) - ratio
        a_t = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha   # [recall] diffusers 0.26: initial_alpha_cumprod
        a_n = self.alphas_cumprod[int(t)]
        x0 = (x - (1 - a_t).sqrt() * eps) / a_t.sqrt()
        return a_n.sqrt() * x0 + (1 - a_n).sqrt() * eps


def denoise_chunk(unet_w, cn_w, latents, disparity, ctx_neg, ctx_pos, guidance, steps, cfg=SD15, num_steps_total=None,
                  mode="xview", trace=None):
    """The 20-step loop inside pipe() as edit_images drives it (gc_pipeline.py:209-219):
    latents [f,4,h,w] with the 4 reference frames FIRST, disparity [f,3,8h,8w], text [1,77,768] each.
    `steps` = how many of the `num_steps_total` DDIM steps to run (tests run a prefix); `trace` (a list) receives the
    latents after every step."""
    n = num_steps_total or steps
    sch = DDIM()
    f = latents.shape[0]
    ctx = torch.cat([ctx_neg.expand(f, -1, -1), ctx_pos.expand(f, -1, -1)], 0)     # [negative || positive]
    cond = torch.cat([disparity, disparity], 0)
    x = latents
    for t in sch.timesteps(n)[:steps]:
        xin = torch.cat([x, x], 0)
        down, mid = controlnet_forward(cn_w, xin, t, ctx, cond, cfg, 1.0, mode, 0.0)
        eps = unet_forward(unet_w, xin, t, ctx, down, mid, cfg, mode, 0.6)
        eu, ec = eps.chunk(2)
        x = sch.step(eu + guidance * (ec - eu), t, x, n)
        if trace is not None:
            trace.append(x.clone())
    return x


# =========================================================================================== glue
def depth2disparity(depth):
    """gc_pipeline.py:258-266: depth [1,H,W] -> disparity [1,3,H,W] = 1/(d+1e-5) / max."""
    disp = 1 / (depth + 1e-5)
    disp = disp / disp.max()
    return torch.cat([disp, disp, disp], 0)[None]


def postprocess_image(x):
    """pipe(output_type='pt'): (x/2 + 0.5).clamp(0,1)"""
    return (x / 2 + 0.5).clamp(0, 1)


def mask_composite(edited, unedited_hwc, mask):
    """gc_pipeline.py:226-234: edited [3,H,W], unedited [H,W,3], mask [H,W] -> [H,W,3] fp32."""
    out = edited * mask[None] + unedited_hwc.permute(2, 0, 1) * (1 - mask)[None]
    return out.permute(1, 2, 0).to(torch.float32)


# =========================================================================================== loss (row A8)
def ssim(a, b, window=11, sigma=1.5, valid=True):
    """SSIM of [B,C,H,W] images in [0,1] with a gaussian 11x11 window (sigma 1.5), C1 = 0.01^2, C2 = 0.03^2.
    valid=True restates pytorch_msssim.SSIM(data_range=1.0, size_average=True) -- the module SplatfactoModel.get_loss_dict calls
    [recall nerfstudio 1.0.0 splatfacto.py / pytorch_msssim ssim.py: separable UNPADDED gaussian filter, sigma products with
    compensation 1.0, mean over the (H-10) x (W-10) map]; valid=False is the zero-padded 'same' variant of the 3DGS code base."""
    coords = torch.arange(window, dtype=a.dtype, device=a.device) - window // 2
    g = torch.exp(-(coords ** 2) / (2 * sigma ** 2)); g = (g / g.sum())
    k = (g[:, None] * g[None, :])[None, None].expand(a.shape[1], 1, window, window)
    mu = lambda x: F.conv2d(x, k, padding=0 if valid else window // 2, groups=x.shape[1])
    ma, mb = mu(a), mu(b)
    va, vb, cab = mu(a * a) - ma * ma, mu(b * b) - mb * mb, mu(a * b) - ma * mb
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    return (((2 * ma * mb + c1) * (2 * cab + c2)) / ((ma * ma + mb * mb + c1) * (va + vb + c2))).mean()


def splat_loss(pred_hwc, gt_hwc, ssim_lambda=0.2, valid=True):
    """SplatfactoModel.get_loss_dict main loss: (1 - l) * L1 + l * (1 - SSIM) (inherited by the reference, gc_pipeline.py:284-285)."""
    l1 = (gt_hwc - pred_hwc).abs().mean()
    return (1 - ssim_lambda) * l1 + ssim_lambda * (1 - ssim(gt_hwc.permute(2, 0, 1)[None], pred_hwc.permute(2, 0, 1)[None], valid=valid))
