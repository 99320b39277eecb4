"""CrossViewAttnProcessor: drop-in for the reference's attention processor
(/root/reference/gaussctrl/utils.py:39-133) following the diffusers attention-processor protocol

    proc(attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, scale=1.0)

constructed as CrossViewAttnProcessor(self_attn_coeff, unet_chunk_size=2) and installed with
`unet.set_attn_processor(...)` (gc_pipeline.py:163-168).  The q/k/v/out projections run on the MFMA GEMM
kernel and the five attentions of utils.py:86-117 run as ONE fused multi-K/V flash-attention launch.
`attn` only needs the attributes the reference touches: to_q / to_k / to_v / to_out[0] (nn.Linear-like, 2-byte
weights on the GPU), heads, residual_connection, rescale_output_factor (group_norm / spatial_norm / norm_cross
must be None / False, as they are for every SD1.x attention layer)."""
from __future__ import annotations

import torch

from .sd import ops


class CrossViewAttnProcessor:
    def __init__(self, self_attn_coeff, unet_chunk_size=2, num_refs=4):
        self.unet_chunk_size = unet_chunk_size
        self.self_attn_coeff = float(self_attn_coeff)
        self.num_refs = num_refs          # the reference hard-wires 4 (utils.py:95-98)

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, scale=1.0):
        if attention_mask is not None:
            raise NotImplementedError("attention_mask is never set on the GaussCtrl path")
        if getattr(attn, "spatial_norm", None) is not None or getattr(attn, "group_norm", None) is not None \
                or getattr(attn, "norm_cross", False):
            raise NotImplementedError("spatial_norm / group_norm / norm_cross are not used by SD1.x attention layers")
        residual = hidden_states
        input_ndim = hidden_states.ndim
        if input_ndim == 4:
            b, c, h, w = hidden_states.shape
            hidden_states = hidden_states.view(b, c, h * w).transpose(1, 2)
        x = hidden_states.contiguous()
        B, L, C = x.shape
        heads = attn.heads
        q = ops.linear(x, attn.to_q.weight, getattr(attn.to_q, "bias", None))
        is_cross = encoder_hidden_states is not None
        src = encoder_hidden_states.contiguous() if is_cross else x
        Lk = src.shape[1]
        Lp = (Lk + 7) // 8 * 8
        k = ops.linear(src, attn.to_k.weight, getattr(attn.to_k, "bias", None))
        inner = attn.to_v.weight.shape[0]
        vt = torch.zeros(src.shape[0], inner, Lp, dtype=x.dtype, device=x.device)
        ops.linear(src, attn.to_v.weight, getattr(attn.to_v, "bias", None), want_out=False, rows_per_batch=Lk, out_t=vt,
                   ldt=Lp, t_batch_stride=inner * Lp)
        if is_cross:
            o = ops.attention(q, k, vt, heads, [(-1, 1.0)], 1, Lk=Lk)            # utils.py:111-117
        else:
            f = B // self.unet_chunk_size                                            # video_length, utils.py:94
            a = self.self_attn_coeff
            sets = ([(-1, a)] if a != 0.0 else []) + [(r, (1.0 - a) / self.num_refs) for r in range(self.num_refs)]
            o = ops.attention(q, k, vt, heads, sets, f, Lk=Lk)                      # utils.py:86-117
        bias = getattr(attn.to_out[0], "bias", None)
        bias = None if bias is None else bias.float()
        out = ops.linear(o, attn.to_out[0].weight, bias)                            # utils.py:121-123 (dropout p=0)
        if input_ndim == 4:
            out = out.transpose(-1, -2).reshape(b, c, h, w)
        if getattr(attn, "residual_connection", False):
            out = out + residual
        rf = getattr(attn, "rescale_output_factor", 1.0)
        return out if rf == 1.0 else out / rf
