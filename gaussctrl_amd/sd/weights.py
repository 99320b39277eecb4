"""Weight preparation: diffusers-format state dicts (fp32, NCHW conv kernels) -> the layouts the HIP
kernels consume.  Key names follow diffusers so real SD1.5 / sd-controlnet-depth / VAE checkpoints load
unchanged (the reference loads them at /root/reference/gaussctrl/gc_pipeline.py:97-102).

  conv 3x3  [Cout,Cin,3,3] -> [Cout_p, 9*Cin_p]  (tap-major, channels-last; Cin/Cout zero-padded to multiples of 8)
  conv 1x1  [Cout,Cin,1,1] -> [Cout, Cin]        (a Linear on channels-last tokens)
  Linear    [out,in]       -> as is
  GEGLU     ff.net.0.proj rows permuted in 16-blocks [x | gate] so the GEMM epilogue pairs them lane-locally
  norms / biases stay fp32.
"""
from __future__ import annotations

import math

import torch


def _pad8(n):
    return (n + 7) // 8 * 8


def conv3x3_weight(w, dtype):
    cout, cin = w.shape[0], w.shape[1]
    cp, op = _pad8(cin), _pad8(cout)
    out = torch.zeros(op, 3, 3, cp, dtype=torch.float32, device=w.device)
    out[:cout, :, :, :cin] = w.float().permute(0, 2, 3, 1)
    return out.reshape(op, 9 * cp).to(dtype).contiguous()


def pad_bias(b, mult=8):
    n = b.shape[0]
    out = torch.zeros(_pad8(n) if mult == 8 else n, dtype=torch.float32, device=b.device)
    out[:n] = b.float()
    return out


def geglu_permute(w, b):
    """rows [x_0..x_{n-1} | g_0..g_{n-1}] -> 16-blocks [x_0..15 | g_0..15 | x_16..31 | g_16..31 ...]"""
    n = w.shape[0] // 2
    assert n % 16 == 0
    idx = torch.arange(n, device=w.device).reshape(n // 16, 16)
    perm = torch.cat([idx, idx + n], dim=1).reshape(-1)
    return w[perm].contiguous(), (None if b is None else b[perm].contiguous())


LOG2E = 1.4426950408889634

# ---- operand streams of the fused transformer tail (csrc/dn_ttail.hip) -------------------------------------------------------
# One MFMA (v_mfma_f32_32x32x16) of a wave consumes one 1 KB block: lane l = 32 hg + row holds A[row][8 hg .. 8 hg + 8) of a 32 x 16
# tile.  The k-values of a tile are stored in the order PERM16, so that k-slot 8 hg + t of lane (m, hg) is the channel the lane already
# holds in register t (t < 4) / 4 + t of the previous GEMM's 32x32 accumulator (rows 8 g + 4 hg + c in register 4 g + c).
PERM16 = [0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15]


def mfma_blocks(w):
    """w [R, K] (R % 32 == 0, K % 16 == 0) -> [K/16, R/32, 64 lanes, 8] operand blocks (k-step major)."""
    R, K = w.shape
    assert R % 32 == 0 and K % 16 == 0
    t = w.reshape(R // 32, 32, K // 16, 16)[..., PERM16].reshape(R // 32, 32, K // 16, 2, 8)
    return t.permute(2, 0, 3, 1, 4).contiguous().reshape(K // 16, R // 32, 64, 8)


def lane_order(v):
    """per-channel vector [N] (N % 32 == 0) -> the order lane (m, hg) reads it: index 32 nb + 16 hg + 4 g + c <-> channel 32 nb + 8 g + 4 hg + c"""
    N = v.shape[0]
    return v.reshape(N // 32, 4, 2, 4).permute(0, 2, 1, 3).contiguous().reshape(N)


GELU_N, GELU_STEP = 2048, 128.0


def gelu_table(device):
    """(gelu(x_i), gelu(x_i+1) - gelu(x_i)) for x_i = (i - GELU_N / 2) / GELU_STEP, exact erf in float64: the fused tail interpolates it linearly"""
    x = (torch.arange(GELU_N + 1, dtype=torch.float64) - GELU_N // 2) / GELU_STEP
    g = 0.5 * x * (1 + torch.erf(x / math.sqrt(2.0)))
    return torch.stack([g[:-1], g[1:] - g[:-1]], 1).reshape(-1).float().to(device)


def tail_streams(out, sd_f32, p, heads, dtype):
    """Stream segments A / B and the parameter table of transformer `p` (C = 320 only).  sd_f32(name) returns the fp32 master of a tensor
    (softmax scale folded into attn2.to_q when the network was prepared with `heads`).  Rounded ONCE to `dtype`, like the per-op weights."""
    t = p + ".transformer_blocks.0"
    W = lambda n: sd_f32(n).to(dtype)
    Cc = sd_f32(p + ".proj_out.weight").shape[0]
    assert Cc == 320 and heads == 8
    seg_a = torch.cat([mfma_blocks(W(t + ".attn1.to_out.0.weight")).reshape(-1), mfma_blocks(W(t + ".attn2.to_q.weight")).reshape(-1)])
    up = W(t + ".ff.net.0.proj.weight")                      # [2 * 1280, 320] = [hidden | gate]
    n = up.shape[0] // 2
    # rows of up-block nb: [hidden 16 nb .. +8 | their gates | hidden 16 nb + 8 .. +8 | their gates]
    idx = torch.arange(n, device=up.device).reshape(n // 16, 2, 8)
    rows = torch.stack([idx, idx + n], dim=2).reshape(-1)     # [nb][half][hidden|gate][8]
    upb = mfma_blocks(up[rows])                               # [20 ks][80 nb]
    dnb = mfma_blocks(W(t + ".ff.net.2.weight"))              # [80 ks][10 nb]
    nit = n // 64
    upb = upb.reshape(20, nit, 4, 64, 8).permute(1, 0, 2, 3, 4)      # [it][ks][j]
    dnb = dnb.reshape(nit, 4, 10, 64, 8)                               # [it][j][nb]
    ffs = torch.cat([upb.reshape(nit, -1), dnb.reshape(nit, -1)], dim=1).reshape(-1)
    po = W(p + ".proj_out.weight").reshape(Cc, Cc)
    seg_b = torch.cat([mfma_blocks(W(t + ".attn2.to_out.0.weight")).reshape(-1), ffs, mfma_blocks(po).reshape(-1)])
    f = lambda name: sd_f32(name).float()
    params = torch.cat([lane_order(f(t + ".attn1.to_out.0.bias")), lane_order(f(t + ".norm2.weight")), lane_order(f(t + ".norm2.bias")),
                        lane_order(f(t + ".attn2.to_out.0.bias")), lane_order(f(t + ".norm3.weight")), lane_order(f(t + ".norm3.bias")),
                        lane_order(f(t + ".ff.net.2.bias")), lane_order(f(p + ".proj_out.bias")),
                        lane_order(f(t + ".ff.net.0.proj.bias")[rows]), gelu_table(sd_f32(p + ".proj_out.bias").device)])
    out[p + ".tail.a"] = seg_a.contiguous(); out[p + ".tail.b"] = seg_b.contiguous(); out[p + ".tail.params"] = params.contiguous()


def head_stream(out, sd_f32, p, heads, dtype):
    """Operand stream and parameter table of the fused head of transformer `p` (csrc/dn_thead.hip): proj_in, attn1.to_q (softmax scale
    folded in), to_k, to_v as MFMA blocks; proj_in bias and LayerNorm1 affine in lane order."""
    t = p + ".transformer_blocks.0"
    W = lambda n: sd_f32(n).to(dtype)
    Cc = sd_f32(p + ".proj_in.weight").shape[0]
    assert Cc == 320
    out[p + ".head.w"] = torch.cat([mfma_blocks(W(p + ".proj_in.weight").reshape(Cc, Cc)).reshape(-1), mfma_blocks(W(t + ".attn1.to_q.weight")).reshape(-1),
                                    mfma_blocks(W(t + ".attn1.to_k.weight")).reshape(-1), mfma_blocks(W(t + ".attn1.to_v.weight")).reshape(-1)]).contiguous()
    f = lambda name: sd_f32(name).float()
    out[p + ".head.params"] = torch.cat([lane_order(f(p + ".proj_in.bias")), lane_order(f(t + ".norm1.weight")), lane_order(f(t + ".norm1.bias"))]).contiguous()


def tail_text_stream(k, vt, Lt, heads):
    """text K [halves, Lt, C] / V^T [halves, C, >= Lt] -> the per-half K / V^T segment of the tail stream: per head 9 K blocks
    (k-step major over d padded 40 -> 48, 3 key blocks of 32) and 12 V^T blocks (key k-step major, 2 channel blocks; channel row 40 = ones,
    so that the P V MFMAs also produce the softmax denominator)."""
    Hh, _, Cc = k.shape
    D = Cc // heads
    segs = []
    for hf in range(Hh):
        for h in range(heads):
            km = torch.zeros(96, 48, dtype=k.dtype, device=k.device)
            km[:Lt, :D] = k[hf, :Lt, h * D:(h + 1) * D]
            vm = torch.zeros(64, 96, dtype=k.dtype, device=k.device)
            vm[:D, :Lt] = vt[hf, h * D:(h + 1) * D, :Lt]
            vm[D, :Lt] = 1
            segs += [mfma_blocks(km).reshape(-1), mfma_blocks(vm).reshape(-1)]
    return torch.cat(segs).reshape(Hh, -1).contiguous()


def prepare(sd: dict, dtype, device, heads=None, fold_ln=False) -> dict:
    """Generic pass over a diffusers state dict.  `heads` (attention heads of the network) folds the softmax scale
    head_dim**-0.5 * log2(e) into the Q projection weights in fp32, before the single rounding to `dtype`: the attention
    kernel then exponentiates the QK^T MFMA output directly (gc_attn_desc.q_prescaled).  fold_ln folds the three LayerNorms of every
    transformer block into the GEMMs that consume them (norm1 -> Q|K|V, norm2 -> attn2.to_q, norm3 -> GEGLU projection); OFF by default:
    on MI355X the longer epilogues of the one-workgroup-per-CU GEMMs cost more than the LayerNorm launches they remove."""
    out = {}
    for k, v in sd.items():
        v = v.to(device)
        if heads and (k.endswith(".attn1.to_q.weight") or k.endswith(".attn2.to_q.weight")):
            v = v.float() * ((v.shape[0] // heads) ** -0.5 * LOG2E)
        if k.endswith(".weight") and v.dim() == 4:
            if v.shape[-1] == 3:
                out[k] = conv3x3_weight(v, dtype)
            else:
                out[k] = v.reshape(v.shape[0], v.shape[1]).to(dtype).contiguous()
        elif k.endswith(".weight") and v.dim() == 2:
            out[k] = v.to(dtype).contiguous()
        elif k.endswith(".bias") and (k[:-5] + ".weight") in sd and sd[k[:-5] + ".weight"].dim() == 4 and sd[k[:-5] + ".weight"].shape[-1] == 3:
            out[k] = pad_bias(v)
        else:
            out[k] = v.float().contiguous()       # norm affine, linear / 1x1 biases
    out["_attn_q_prescaled"] = bool(heads)
    out["_ln_folded"] = bool(fold_ln)
    # fold_ln == 2: fold only the blocks that have no one-launch head / tail (C = 640 / 1280); the C = 320 blocks keep their LayerNorms inside
    # the row-resident kernels
    if heads == 8 and fold_ln in (False, 0, 2):        # level-0 transformers (C = 320): operand streams of the one-launch tail (dn_ttail.hip)
        def master(name):
            v = sd[name].to(device).float()
            return v * ((v.shape[0] // heads) ** -0.5 * LOG2E) if name.endswith((".attn1.to_q.weight", ".attn2.to_q.weight")) else v
        for k in list(sd.keys()):
            if k.endswith(".transformer_blocks.0.attn1.to_q.weight") and sd[k].shape[0] == 320:
                tail_streams(out, master, k[:-len(".transformer_blocks.0.attn1.to_q.weight")], heads, dtype)
                if not any(kk.endswith("attn1.to_q.bias") for kk in sd):            # SD1.5: no bias on the attention projections
                    head_stream(out, master, k[:-len(".transformer_blocks.0.attn1.to_q.weight")], heads, dtype)
    for k in list(out.keys()):
        if k.endswith(".attn1.to_q.weight"):            # fused Q|K|V projection of the self-attention layers
            a = k[:-len("to_q.weight")]
            t = a[:-len("attn1.")]                       # "...transformer_blocks.0."
            if fold_ln and not (fold_ln == 2 and (t[:-len("transformer_blocks.0.")] + "tail.a") in out):
                # LayerNorm folded into the GEMM that consumes it (gc_gemm_desc.ln_*): W' = W diag(gamma) rounded ONCE from fp32,
                # bias' = b + W beta, colsum = sum_k W'[n][k] of the ROUNDED weights (the epilogue subtracts mean * colsum exactly)
                qkv32 = torch.cat([_f32(sd, out, a + "to_q.weight", heads, device), _f32(sd, out, a + "to_k.weight", None, device),
                                   _f32(sd, out, a + "to_v.weight", None, device)], 0)
                _fold_ln(out, a + "to_qkv", qkv32, None, out[t + "norm1.weight"], out[t + "norm1.bias"], dtype)
                q2 = _f32(sd, out, t + "attn2.to_q.weight", heads, device)
                _fold_ln(out, t + "attn2.to_q", q2, None, out[t + "norm2.weight"], out[t + "norm2.bias"], dtype)
                ff = sd[t + "ff.net.0.proj.weight"].to(device).float()
                _fold_ln(out, t + "ff.net.0.proj", ff, out[t + "ff.net.0.proj.bias"], out[t + "norm3.weight"], out[t + "norm3.bias"], dtype)
                # feed-forward down projection and the block's proj_out as ONE GEMM over [ff | h] (K = 5 C):
                #   out = (ff Wd^T + bd + h) Wp^T + bp + x  =  [ff | h] [Wp Wd | Wp]^T + (Wp bd + bp) + x
                # (constant folding of the checkpoint's weights, in fp32 from the fp32 masters, ONE rounding to the activation type)
                pfx = t[:-len("transformer_blocks.0.")]
                wp = sd[pfx + "proj_out.weight"].to(device).float()
                wp = wp.reshape(wp.shape[0], wp.shape[1])
                wd = sd[t + "ff.net.2.weight"].to(device).float()
                out[t + "ffout.weight"] = torch.cat([wp @ wd, wp], 1).to(dtype).contiguous()                    # [C, 5 C]
                out[t + "ffout.bias"] = (wp @ sd[t + "ff.net.2.bias"].to(device).float() + sd[pfx + "proj_out.bias"].to(device).float()).contiguous()
            else:
                out[a + "to_qkv.weight"] = torch.cat([out[a + "to_q.weight"], out[a + "to_k.weight"], out[a + "to_v.weight"]], 0).contiguous()
    names = sorted(k[:-len(".time_emb_proj.weight")] for k in out if k.endswith(".time_emb_proj.weight"))
    if names:                                          # all time-embedding projections of a network as ONE [sum Cout, 1280] GEMM
        out["_temb_all.weight"] = torch.cat([out[n + ".time_emb_proj.weight"] for n in names], 0).contiguous()
        out["_temb_all.bias"] = torch.cat([out[n + ".time_emb_proj.bias"] for n in names], 0).contiguous()
        off = 0
        idx = {}
        for n in names:
            c = out[n + ".time_emb_proj.weight"].shape[0]
            idx[n] = (off, c); off += c
        out["_temb_all.index"] = idx
    for k in list(out.keys()):
        if k.endswith("ff.net.0.proj.weight"):
            b = k[:-6] + "bias"
            out[k], out[b] = geglu_permute(out[k], out[b])
            if fold_ln and (k[:-len("ff.net.0.proj.weight")] + "attn1.to_qkv.colsum") in out:
                out[k[:-6] + "colsum"] = out[k].float().sum(1).contiguous()      # of the rounded, permuted rows
    return out


def _f32(sd, out, key, heads, device):
    """fp32 master of a projection weight (with the softmax scale folded into Q rows when `heads` is given)."""
    v = sd[key].to(device).float()
    if heads:
        v = v * ((v.shape[0] // heads) ** -0.5 * LOG2E)
    return v


def _fold_ln(out, name, w32, bias, gamma, beta, dtype):
    """y = LN(x) W^T + b  ==  rstd (x W'^T - mean colsum) + b'   with W' = W diag(gamma), b' = b + W beta."""
    wf = (w32 * gamma.float()[None, :]).to(dtype).contiguous()
    out[name + ".weight"] = wf
    out[name + ".colsum"] = wf.float().sum(1).contiguous()
    b2 = w32 @ beta.float()
    out[name + ".bias"] = (b2 if bias is None else bias.float() + b2).contiguous()


# ------------------------------------------------------------------------------------------------ fp8 (BASELINE configs[3])
def quantize_rows_e4m3(w32: torch.Tensor):
    """[N, K] fp32 -> (e4m3 bytes [N, K] uint8, E8M0 scale bytes [N] uint8): per output row a power-of-two scale that puts the row
    maximum just under the e4m3 maximum (448); real value = stored * 2^(byte - 127), the factor the block-scaled MFMA applies."""
    amax = w32.abs().amax(dim=1).clamp_min(1e-30)
    e = torch.floor(torch.log2(448.0 / amax)).clamp(-100, 100)                 # stored = w * 2^e
    q = (w32 * torch.exp2(e)[:, None]).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    return q.view(torch.uint8).contiguous(), (127 - e).to(torch.uint8).contiguous()


def conv3x3_weight_fp8(w: torch.Tensor):
    """[Cout, Cin, 3, 3] -> (bytes [Cout_p8, 9 * pad128(Cin)] in (tap, cin) order with zero channel padding, scale bytes [Cout_p8])"""
    cout, cin = w.shape[0], w.shape[1]
    cp, op = (cin + 127) // 128 * 128, _pad8(cout)
    t = torch.zeros(op, 3, 3, cp, dtype=torch.float32, device=w.device)
    t[:cout, :, :, :cin] = w.float().permute(0, 2, 3, 1)
    return quantize_rows_e4m3(t.reshape(op, 9 * cp))


def add_fp8_convs(out: dict, sd: dict, device) -> dict:
    """fp8 copies of the resnet 3x3 convolutions (conv1 / conv2: their inputs are GroupNorm + SiLU outputs, quantised by
    gc_dn_groupnorm_apply_fp8); everything else of the network stays in the 2-byte type."""
    for k, v in sd.items():
        if k.endswith((".conv1.weight", ".conv2.weight")) and ".resnets." in k and v.dim() == 4 and v.shape[-1] == 3:
            q, sc = conv3x3_weight_fp8(v.to(device))
            out[k[:-len("weight")] + "w8"], out[k[:-len("weight")] + "w8_scale"] = q, sc
    out["_fp8_convs"] = True
    return out


def add_fp8_linears(out: dict, which: int = 7) -> dict:
    """fp8 copies of the transformer-block linears whose input a LayerNorm or the GEGLU epilogue can write as e4m3 (levels with
    C % 128 == 0, i.e. the C = 640 / 1280 blocks; the C = 320 blocks run in the row-resident head / tail kernels): the fused Q | K | V
    projection, attn2.to_q, the GEGLU projection and the FF down projection -- ~85 % of a block's linear FLOPs.  Quantised from the PREPARED
    tensors of `out` (softmax scale folded into Q, GEGLU rows permuted), per output row (quantize_rows_e4m3).  Not with folded LayerNorms."""
    assert not out.get("_ln_folded"), "fp8 linears take their input from the LayerNorm kernel: prepare(fold_ln=False)"
    for k in list(out.keys()):
        if k.endswith(".attn1.to_qkv.weight") and out[k].shape[1] % 128 == 0:
            t = k[:-len("attn1.to_qkv.weight")]
            for name in ("attn1.to_qkv", "attn2.to_q", "ff.net.0.proj", "ff.net.2"):
                q, sc = quantize_rows_e4m3(out[t + name + ".weight"].float())
                out[t + name + ".w8"], out[t + name + ".w8_scale"] = q, sc
    out["_fp8_linears"] = int(which)        # bit 0: feed-forward, bit 1: attn2.to_q, bit 2: Q | K | V (SDNet.fp8_lin)
    return out
