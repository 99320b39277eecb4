"""Generates tests/golden/xview_attn_*.npz by IMPORTING the reference's own cross-view attention
processor (/root/reference/gaussctrl/utils.py:25-133) in the build container and running it on seeded
inputs.  Runs only here (the reference never travels); the .npz files are data: inputs, weights and the
reference's outputs.

utils.py's two missing imports (torchvision.transforms names it never uses, and
diffusers.utils.USE_PEFT_BACKEND) are satisfied with stub modules; the `attn` argument is a minimal
stand-in for diffusers 0.26.0's `Attention` module exposing exactly the attributes the processor touches
(attention-processor surface) with diffusers' semantics [recall]:
  head_to_batch_dim [B,L,C]->[B*H,L,C/H]; get_attention_scores = softmax(scale * q k^T, dim=-1);
  to_q/to_k/to_v without bias, to_out = [Linear with bias, Dropout(0)].

`big` adds the PRODUCTION geometry of the dominant kernel (SD1.5 level 0: L = 4096 tokens, 8 heads x D = 40, f = 5 frames,
CFG batch 10 -> the k_attn4 launch; and level 1: L = 1024, D = 80 -> k_attn3), run through the same imported reference
processor (five [80, L, L] fp32 probability tensors, ~16 GB peak).  Inputs and weights come from seeded CPU generators
(`big_inputs`, re-created by the GPU test), the .npz keeps the reference's output on a strided subset of token rows.

usage: python tests/golden/make_xview_golden.py [small] [big]
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/gaussctrl/utils.py"


def load_reference_utils():
    tv = types.ModuleType("torchvision"); tvt = types.ModuleType("torchvision.transforms")
    tvt.Resize = object; tvt.InterpolationMode = object; tv.transforms = tvt
    df = types.ModuleType("diffusers"); dfu = types.ModuleType("diffusers.utils"); dfu.USE_PEFT_BACKEND = True
    df.utils = dfu
    sys.modules.setdefault("torchvision", tv); sys.modules.setdefault("torchvision.transforms", tvt)
    sys.modules.setdefault("diffusers", df); sys.modules.setdefault("diffusers.utils", dfu)
    spec = importlib.util.spec_from_file_location("ref_gaussctrl_utils", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class FakeAttention(torch.nn.Module):
    """diffusers.models.attention_processor.Attention stand-in (only what utils.py touches)."""

    def __init__(self, query_dim, cross_dim, heads, dim_head):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.to_q = torch.nn.Linear(query_dim, inner, bias=False)
        self.to_k = torch.nn.Linear(cross_dim or query_dim, inner, bias=False)
        self.to_v = torch.nn.Linear(cross_dim or query_dim, inner, bias=False)
        self.to_out = torch.nn.ModuleList([torch.nn.Linear(inner, query_dim), torch.nn.Dropout(0.0)])
        self.spatial_norm = None
        self.group_norm = None
        self.norm_cross = False
        self.residual_connection = False
        self.rescale_output_factor = 1.0

    def prepare_attention_mask(self, attention_mask, target_length, batch_size):
        return attention_mask

    def head_to_batch_dim(self, t):
        b, l, c = t.shape
        h = self.heads
        return t.reshape(b, l, h, c // h).permute(0, 2, 1, 3).reshape(b * h, l, c // h)

    def batch_to_head_dim(self, t):
        bh, l, d = t.shape
        h = self.heads
        return t.reshape(bh // h, h, l, d).permute(0, 2, 1, 3).reshape(bh // h, l, d * h)

    def get_attention_scores(self, query, key, attention_mask=None):
        scores = torch.baddbmm(torch.empty(query.shape[0], query.shape[1], key.shape[1], dtype=query.dtype),
                               query, key.transpose(-1, -2), beta=0, alpha=self.scale)
        return scores.softmax(dim=-1)


BIG_CASES = [  # (name, frames f, tokens L, heads, dim_head, text_len, cross_dim, self_attn_coeff, row stride)
    ("big_l4096_d40", 5, 4096, 8, 40, 77, 768, 0.6, 41),
    ("big_l1024_d80", 5, 1024, 8, 80, 77, 768, 0.6, 23),
]


def big_inputs(seed, f, L, H, D, Lt, Ct):
    """Seeded inputs / weights of a production-geometry case (the GPU test re-creates them with the same calls).  to_q / to_k
    are drawn twice as wide as 1/sqrt(C) so the logits have a spread of ~4 and the softmax is peaked, not uniform."""
    g = torch.Generator().manual_seed(seed)
    C, B = H * D, 2 * f
    r = lambda *s: torch.randn(*s, generator=g)
    out = {}
    for kind, cin in (("self", C), ("text", Ct)):
        out[kind] = dict(x=r(B, L, C), ctx=None if kind == "self" else r(B, Lt, Ct),
                         wq=r(C, C) * (2.0 * C ** -0.5), wk=r(C, cin) * (2.0 * cin ** -0.5), wv=r(C, cin) * cin ** -0.5,
                         wo=r(C, C) * C ** -0.5, bo=r(C) * 0.1)
    return out


def big(ref):
    for case_idx, (name, f, L, H, D, Lt, Ct, coeff, stride) in enumerate(BIG_CASES):
        seed = 2000 + case_idx
        C = H * D
        inp = big_inputs(seed, f, L, H, D, Lt, Ct)
        proc = ref.CrossViewAttnProcessor(self_attn_coeff=coeff, unet_chunk_size=2)
        out = {}
        for kind, cross in (("self", None), ("text", Ct)):
            d = inp[kind]
            attn = FakeAttention(C, cross, H, D)
            with torch.no_grad():
                attn.to_q.weight.copy_(d["wq"]); attn.to_k.weight.copy_(d["wk"]); attn.to_v.weight.copy_(d["wv"])
                attn.to_out[0].weight.copy_(d["wo"]); attn.to_out[0].bias.copy_(d["bo"])
                y = proc(attn, d["x"], encoder_hidden_states=d["ctx"])
            out[f"{kind}_y_rows"] = y[:, ::stride].contiguous().numpy().astype(np.float32)
            out[f"{kind}_y_norm"] = np.array(float(y.double().norm()))
        out["meta"] = np.array([f, L, H, D, Lt, Ct, seed, stride], np.int64); out["coeff"] = np.array(coeff, np.float64)
        np.savez_compressed(os.path.join(HERE, f"xview_{name}.npz"), **out)
        print("wrote", name, {k: v.shape for k, v in out.items() if hasattr(v, "shape")}, flush=True)


def main():
    ref = load_reference_utils()
    what = sys.argv[1:] or ["small"]
    if "big" in what:
        big(ref)
    if "small" not in what:
        return
    cases = [  # (name, frames f, tokens L, heads, dim_head, text_len, cross_dim, self_attn_coeff)
        ("unet_f5", 5, 32, 8, 8, 11, 48, 0.6),
        ("unet_f7", 7, 48, 8, 8, 11, 48, 0.6),
        ("controlnet_f5", 5, 32, 8, 8, 11, 48, 0.0),
        ("unet_f12_d40", 12, 16, 2, 40, 7, 24, 0.6),
    ]
    for case_idx, (name, f, L, H, D, Lt, Ct, coeff) in enumerate(cases):
        torch.manual_seed(1000 + case_idx)
        C = H * D
        B = 2 * f                                   # CFG doubling, unet_chunk_size = 2 (utils.py:94)
        proc = ref.CrossViewAttnProcessor(self_attn_coeff=coeff, unet_chunk_size=2)
        out = {}
        for kind, cross in (("self", None), ("text", Ct)):
            attn = FakeAttention(C, cross, H, D)
            x = torch.randn(B, L, C)
            ctx = None if cross is None else torch.randn(B, Lt, Ct)
            with torch.no_grad():
                y = proc(attn, x, encoder_hidden_states=ctx)
            out[f"{kind}_x"] = x.numpy(); out[f"{kind}_y"] = y.numpy()
            if ctx is not None:
                out[f"{kind}_ctx"] = ctx.numpy()
            out[f"{kind}_wq"] = attn.to_q.weight.detach().numpy(); out[f"{kind}_wk"] = attn.to_k.weight.detach().numpy()
            out[f"{kind}_wv"] = attn.to_v.weight.detach().numpy(); out[f"{kind}_wo"] = attn.to_out[0].weight.detach().numpy()
            out[f"{kind}_bo"] = attn.to_out[0].bias.detach().numpy()
        out["meta"] = np.array([f, L, H, D, Lt, Ct], np.int64); out["coeff"] = np.array(coeff, np.float64)
        np.savez_compressed(os.path.join(HERE, f"xview_attn_{name}.npz"), **{k: (v.astype(np.float32) if v.dtype == np.float32 else v) for k, v in out.items()})
        print("wrote", name, {k: v.shape for k, v in out.items() if hasattr(v, "shape")})


if __name__ == "__main__":
    main()
