"""AutoencoderKL decoder (SD1.x VAE) on the HIP kernels, channels-last.

Mirror of `vae.decode(latents / 0.18215)` + `(x/2+0.5).clamp(0,1)` that the reference runs inside
`pipe(..., output_type='pt')` (/root/reference/gaussctrl/gc_pipeline.py:209-219).
The single-head 512-wide mid-block attention (4096 tokens) runs as two MFMA GEMMs + a row softmax.
"""
from __future__ import annotations

import torch

from . import ops

CFG_VAE = dict(block_out_channels=(128, 256, 512, 512), layers_per_block=2, groups=32)


class VAEDecoder:
    def __init__(self, weights: dict, cfg: dict = CFG_VAE):
        self.w = weights
        self.cfg = cfg
        self.dtype = weights["decoder.conv_in.weight"].dtype

    def resnet(self, p, x):
        w = self.w
        g = self.cfg["groups"]
        h = ops.groupnorm(x, w[p + ".norm1.weight"], w[p + ".norm1.bias"], g, 1e-6, True)
        h = ops.conv3x3(h, w[p + ".conv1.weight"], w[p + ".conv1.bias"])
        h = ops.groupnorm(h, w[p + ".norm2.weight"], w[p + ".norm2.bias"], g, 1e-6, True)
        sc = x
        if (p + ".conv_shortcut.weight") in w:
            sc = ops.linear(x, w[p + ".conv_shortcut.weight"], w[p + ".conv_shortcut.bias"])
        return ops.conv3x3(h, w[p + ".conv2.weight"], w[p + ".conv2.bias"], residual=sc)

    def mid_attention(self, x):
        w = self.w
        a = "decoder.mid_block.attentions.0"
        B, H, W_, Cc = x.shape
        L = H * W_
        Lp = (L + 7) // 8 * 8
        h = ops.groupnorm(x, w[a + ".group_norm.weight"], w[a + ".group_norm.bias"], self.cfg["groups"], 1e-6, False).view(B, L, Cc)
        q = ops.linear(h, w[a + ".to_q.weight"], w[a + ".to_q.bias"])
        k = ops.linear(h, w[a + ".to_k.weight"], w[a + ".to_k.bias"])
        vt = torch.zeros(B, Cc, Lp, dtype=x.dtype, device=x.device)
        ops.linear(h, w[a + ".to_v.weight"], w[a + ".to_v.bias"], want_out=False, rows_per_batch=L, out_t=vt, ldt=Lp,
                   t_batch_stride=Cc * Lp)
        o = torch.empty(B, L, Cc, dtype=x.dtype, device=x.device)
        for b in range(B):
            s = torch.zeros(L, Lp, dtype=x.dtype, device=x.device) if Lp != L else torch.empty(L, Lp, dtype=x.dtype, device=x.device)
            ops.linear(q[b], k[b], out=s[:, :L] if Lp != L else s)
            ops.softmax_rows_(s[:, :L] if Lp != L else s, Cc ** -0.5)
            ops.linear(s, vt[b], out=o[b])
        out = ops.linear(o, w[a + ".to_out.0.weight"], w[a + ".to_out.0.bias"], residual=x.view(B, L, Cc))
        return out.view(B, H, W_, Cc)

    def decode(self, z, postprocess=False):
        """z: [B,h,w,8] activation dtype (latents / 0.18215 in channels 0..3) -> image fp32 [B,8h,8w,8] (3 valid), in
        [-1,1]; postprocess=True fuses `(x/2+0.5).clamp(0,1)` of pipe(output_type='pt') into the last conv."""
        w = self.w
        B, H, W_, _ = z.shape
        x = ops.linear(z, w["post_quant_conv.weight"], w["post_quant_conv.bias"])
        x = ops.conv3x3(x, w["decoder.conv_in.weight"], w["decoder.conv_in.bias"])
        x = self.resnet("decoder.mid_block.resnets.0", x)
        x = self.mid_attention(x)
        x = self.resnet("decoder.mid_block.resnets.1", x)
        n = len(self.cfg["block_out_channels"])
        for i in range(n):
            for j in range(self.cfg["layers_per_block"] + 1):
                x = self.resnet(f"decoder.up_blocks.{i}.resnets.{j}", x)
            if i < n - 1:
                p = f"decoder.up_blocks.{i}.upsamplers.0.conv"
                x = ops.conv3x3(x, w[p + ".weight"], w[p + ".bias"], upsample=True)
        x = ops.groupnorm(x, w["decoder.conv_norm_out.weight"], w["decoder.conv_norm_out.bias"], self.cfg["groups"], 1e-6, True)
        return ops.conv3x3(x, w["decoder.conv_out.weight"], w["decoder.conv_out.bias"], out_f32=True, act=2 if postprocess else 0)


class VAEEncoder(VAEDecoder):
    """AutoencoderKL.encode(x).latent_dist.mean -- image2latent of the reference (gc_pipeline.py:239-246)."""

    def __init__(self, weights: dict, cfg: dict = CFG_VAE):
        self.w = weights
        self.cfg = cfg
        self.dtype = weights["encoder.conv_in.weight"].dtype

    def mid_attention(self, x):
        # same block as the decoder's, under the encoder prefix
        w = dict((k.replace("encoder.mid_block", "decoder.mid_block"), v) for k, v in self.w.items() if k.startswith("encoder.mid_block.attentions"))
        saved, self.w = self.w, {**self.w, **w}
        try:
            return VAEDecoder.mid_attention(self, x)
        finally:
            self.w = saved

    def encode_mean(self, img):
        """img [B,H,W,8] activation dtype, channels 0..2 in [-1,1] -> latent mean fp32 [B,H/8,W/8,8] (4 valid), NOT yet
        multiplied by 0.18215."""
        w = self.w
        x = ops.conv3x3(img, w["encoder.conv_in.weight"], w["encoder.conv_in.bias"])
        n = len(self.cfg["block_out_channels"])
        for i in range(n):
            for j in range(self.cfg["layers_per_block"]):
                x = self.resnet(f"encoder.down_blocks.{i}.resnets.{j}", x)
            if i < n - 1:
                p = f"encoder.down_blocks.{i}.downsamplers.0.conv"
                x = ops.conv3x3(x, w[p + ".weight"], w[p + ".bias"], stride=2, pad_lo=0)     # F.pad(0,1,0,1) + stride-2 conv
        x = self.resnet("encoder.mid_block.resnets.0", x)
        x = self.mid_attention(x)
        x = self.resnet("encoder.mid_block.resnets.1", x)
        x = ops.groupnorm(x, w["encoder.conv_norm_out.weight"], w["encoder.conv_norm_out.bias"], self.cfg["groups"], 1e-6, True)
        x = ops.conv3x3(x, w["encoder.conv_out.weight"], w["encoder.conv_out.bias"])          # [.., 8] = mean(4) | logvar(4)
        return ops.linear(x, w["quant_conv.weight"], w["quant_conv.bias"], out_f32=True)


def prepare_vae_weights(sd: dict, dtype, device) -> dict:
    """post_quant_conv is a 4->4 1x1 conv: pad it to 8->8 so it runs on the 8-channel latent layout."""
    from .weights import prepare
    out = prepare(sd, dtype, device)
    wq = torch.zeros(8, 8, dtype=dtype, device=device)
    wq[:4, :4] = out["post_quant_conv.weight"]
    bq = torch.zeros(8, dtype=torch.float32, device=device)
    bq[:4] = out["post_quant_conv.bias"]
    out["post_quant_conv.weight"], out["post_quant_conv.bias"] = wq, bq
    return out


def prepare_vae_encoder_weights(sd: dict, dtype, device) -> dict:
    from .weights import prepare
    return prepare(sd, dtype, device)
