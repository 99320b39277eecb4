"""The one-launch transformer tail (gc_dn_transformer_tail, csrc/dn_ttail.hip) against (a) a plain PyTorch fp32 restatement of the same
nine operations (diffusers BasicTransformerBlock.forward after attn1 + Transformer2DModel.proj_out: what the reference's UNet executes
behind gaussctrl/gc_pipeline.py:224-227) and (b) the per-op HIP path it replaces; then through the UNet (GC_FUSED_TAIL)."""
import math

import pytest
from _margins import within
import torch

pytestmark = pytest.mark.gpu
C, H, FFN, CTX = 320, 8, 1280, 768
P = "tb"; T = P + ".transformer_blocks.0"


def _sd(seed=0, gate_shift=0.0):
    """gate_shift: added to the GEGLU gate biases (half of them negated): pushes gates beyond the +-8 of the tail kernel's GELU table"""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    sd = {P + ".proj_in.weight": r(C, C, 1, 1, sc=C ** -0.5), P + ".proj_in.bias": r(C, sc=0.1),
          P + ".proj_out.weight": r(C, C, 1, 1, sc=C ** -0.5), P + ".proj_out.bias": r(C, sc=0.1),
          T + ".ff.net.0.proj.weight": r(2 * FFN, C, sc=C ** -0.5), T + ".ff.net.0.proj.bias": r(2 * FFN, sc=0.1),
          T + ".ff.net.2.weight": r(C, FFN, sc=FFN ** -0.5), T + ".ff.net.2.bias": r(C, sc=0.1)}
    if gate_shift:
        sgn = torch.where(torch.arange(FFN) % 2 == 0, 1.0, -1.0)
        sd[T + ".ff.net.0.proj.bias"][FFN:] += gate_shift * sgn
    for n in ("norm1", "norm2", "norm3"):
        sd[T + f".{n}.weight"] = 1 + r(C, sc=0.1); sd[T + f".{n}.bias"] = r(C, sc=0.1)
    for a, kin in (("attn1", C), ("attn2", CTX)):
        sd[T + f".{a}.to_q.weight"] = r(C, C, sc=C ** -0.5 * 2)
        sd[T + f".{a}.to_k.weight"] = r(C, kin, sc=kin ** -0.5 * 2); sd[T + f".{a}.to_v.weight"] = r(C, kin, sc=kin ** -0.5)
        sd[T + f".{a}.to_out.0.weight"] = r(C, C, sc=C ** -0.5); sd[T + f".{a}.to_out.0.bias"] = r(C, sc=0.1)
    return sd


def _torch_tail(sd, o1, h, x, ctx, f):
    """fp32, unfused, straight from the module definitions"""
    F = torch.nn.functional
    W = lambda n: sd[n].float().to(o1.device)
    o1, h, x, ctx = o1.float(), h.float(), x.float(), ctx.float()
    B, HW, _ = o1.shape
    h1 = F.linear(o1, W(T + ".attn1.to_out.0.weight"), W(T + ".attn1.to_out.0.bias")) + h
    n2 = F.layer_norm(h1, (C,), W(T + ".norm2.weight"), W(T + ".norm2.bias"), 1e-5)
    q = F.linear(n2, W(T + ".attn2.to_q.weight")).view(B, HW, H, C // H).transpose(1, 2)
    cx = ctx.repeat_interleave(f, 0)                                   # frame b reads the text of its CFG half
    k = F.linear(cx, W(T + ".attn2.to_k.weight")).view(B, -1, H, C // H).transpose(1, 2)
    v = F.linear(cx, W(T + ".attn2.to_v.weight")).view(B, -1, H, C // H).transpose(1, 2)
    o = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(C // H), -1) @ v
    h2 = F.linear(o.transpose(1, 2).reshape(B, HW, C), W(T + ".attn2.to_out.0.weight"), W(T + ".attn2.to_out.0.bias")) + h1
    n3 = F.layer_norm(h2, (C,), W(T + ".norm3.weight"), W(T + ".norm3.bias"), 1e-5)
    hid, gate = F.linear(n3, W(T + ".ff.net.0.proj.weight"), W(T + ".ff.net.0.proj.bias")).chunk(2, -1)
    h3 = F.linear(hid * F.gelu(gate), W(T + ".ff.net.2.weight"), W(T + ".ff.net.2.bias")) + h2
    return F.linear(h3, W(P + ".proj_out.weight").view(C, C), W(P + ".proj_out.bias")) + x


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,HW,f,Lt,gate_shift", [(2, 256, 1, 77, 0.0), (4, 128, 2, 50, 0.0), (6, 384, 3, 96, 0.0), (2, 128, 1, 77, 14.0)])
def test_tail_matches_torch_and_the_per_op_path(dtype, B, HW, f, Lt, gate_shift):
    """gate_shift = 14: GEGLU gates around +-14, far outside the GELU table's +-8 (gelu(x) = x / 0 there: the end segments extrapolate)"""
    from gaussctrl_amd.sd import ops, weights
    dev = "cuda:0"
    sd = _sd(gate_shift=gate_shift)
    w = weights.prepare(sd, dtype, dev, heads=H)
    g = torch.Generator().manual_seed(1)
    o1, h, x = (torch.randn(B, HW, C, generator=g).to(dtype).to(dev) for _ in range(3))
    ctx = torch.randn(B // f, Lt, CTX, generator=g).to(dtype).to(dev)
    Lp = (Lt + 7) // 8 * 8
    k = ops.linear(ctx, w[T + ".attn2.to_k.weight"])
    vt = torch.zeros(B // f, C, Lp, dtype=dtype, device=dev)
    ops.linear(ctx, w[T + ".attn2.to_v.weight"], want_out=False, rows_per_batch=Lt, out_t=vt, ldt=Lp, t_batch_stride=C * Lp)
    kv = weights.tail_text_stream(k, vt, Lt, H)
    got = ops.transformer_tail(o1, h, x, w[P + ".tail.a"], kv, w[P + ".tail.b"], w[P + ".tail.params"], H, f, Lt).float()
    # (a) fp32 torch: the error is the 16-bit rounding of the intermediates (per-op path measured the same way below)
    ref = _torch_tail(sd, o1, h, x, ctx, f)
    # (b) the nine launches it replaces
    h1 = ops.linear(o1, w[T + ".attn1.to_out.0.weight"], w[T + ".attn1.to_out.0.bias"], residual=h)
    q = ops.linear(ops.layernorm(h1, w[T + ".norm2.weight"], w[T + ".norm2.bias"]), w[T + ".attn2.to_q.weight"])
    o = ops.attention(q, k, vt, H, [(-2, 1.0)], f, Lk=Lt, q_prescaled=True)
    h2 = ops.linear(o, w[T + ".attn2.to_out.0.weight"], w[T + ".attn2.to_out.0.bias"], residual=h1)
    ff = ops.linear(ops.layernorm(h2, w[T + ".norm3.weight"], w[T + ".norm3.bias"]), w[T + ".ff.net.0.proj.weight"],
                    w[T + ".ff.net.0.proj.bias"], geglu=True)
    h3 = ops.linear(ff, w[T + ".ff.net.2.weight"], w[T + ".ff.net.2.bias"], residual=h2)
    per_op = ops.linear(h3, w[P + ".proj_out.weight"], w[P + ".proj_out.bias"], residual=x).float()
    scale = ref.abs().max()
    e_fused, e_per_op = (got - ref).abs().max() / scale, (per_op - ref).abs().max() / scale
    bar = 2e-2 if dtype == torch.bfloat16 else 3e-3            # of the output range: ~2.5 ulp of the 16-bit format at the largest value
    within("e_fused", e_fused, bar, strict=True)
    within("e_fused", e_fused, 2 * e_per_op + 1e-3, strict=True)  # no worse than the path it replaces
    within("(got - per_op).abs().max() / scale", (got - per_op).abs().max() / scale, bar, strict=True)


def test_unet_with_fused_tail_matches_per_op_unet():
    """the whole UNet (random SD1.5-shaped weights, 2 frames = one per CFG half, 32x32 latents -> 1024 tokens at level 0) with the heads and tails of the five
    level-0 transformer blocks fused vs per-op: same eps up to the 16-bit rounding of the intermediates"""
    from gaussctrl_amd.sd import arch, unet as U, weights
    dev = "cuda:0"
    sd = arch.random_state_dict(arch.unet_shapes(), 0, dev)
    w = weights.prepare(sd, torch.bfloat16, dev, heads=8)
    assert "down_blocks.0.attentions.0.tail.a" in w and "down_blocks.1.attentions.0.tail.a" not in w       # C = 320 only
    net = U.UNet(w)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 32, 32, 8, generator=g).to(torch.bfloat16).to(dev)
    ctx = torch.randn(2, 77, 768, generator=g).to(torch.bfloat16).to(dev)
    outs = []
    for fused in (False, True):
        net.fused_tail = net.fused_head = fused
        actx = U.AttnCtx("plain", 0.0, 1, {}, None, "unet")
        outs.append(net.forward(x, 500.0, ctx, None, None, actx)[..., :4].float())
    d = (outs[0] - outs[1]).abs().max() / outs[0].abs().max()
    assert 0 < d < 3e-2, d                          # different kernels (not bit-equal), same function


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,HW", [(2, 256), (3, 1024)])
def test_head_matches_torch_and_the_per_op_path(dtype, B, HW):
    """gc_dn_transformer_head (GroupNorm apply, proj_in, LayerNorm1, Q | K | V^T) vs fp32 torch and vs the four launches it replaces"""
    from gaussctrl_amd.sd import ops, weights
    F = torch.nn.functional
    dev = "cuda:0"
    sd = _sd()
    sd[P + ".norm.weight"] = 1 + 0.1 * torch.randn(C, generator=torch.Generator().manual_seed(5)); sd[P + ".norm.bias"] = 0.1 * torch.randn(C, generator=torch.Generator().manual_seed(6))
    w = weights.prepare(sd, dtype, dev, heads=H)
    assert P + ".head.w" in w
    g = torch.Generator().manual_seed(2)
    x = (torch.randn(B, HW, C, generator=g) * 1.5 + 0.3).to(dtype).to(dev)
    coef = ops.groupnorm_coef(x, w[P + ".norm.weight"], w[P + ".norm.bias"], 32, 1e-6)
    h, qk, vt = ops.transformer_head(x, coef, w[P + ".head.w"], w[P + ".head.params"])
    # per-op path
    xn = ops.groupnorm(x, w[P + ".norm.weight"], w[P + ".norm.bias"], 32, 1e-6, False)
    h_ref = ops.linear(xn, w[P + ".proj_in.weight"], w[P + ".proj_in.bias"])
    n1 = ops.layernorm(h_ref, w[T + ".norm1.weight"], w[T + ".norm1.bias"])
    vt_ref = torch.empty(B, C, HW, dtype=dtype, device=dev)
    qk_ref = ops.linear(n1, w[T + ".attn1.to_qkv.weight"], None, rows_per_batch=HW, out_t=vt_ref, ldt=HW, t_batch_stride=C * HW, t_col0=2 * C, out_cols=2 * C)
    # fp32 torch (Q carries the folded softmax scale: D^-1/2 log2 e)
    W = lambda n: sd[n].float().to(dev)
    xt = F.group_norm(x.float().transpose(1, 2), 32, W(P + ".norm.weight"), W(P + ".norm.bias"), 1e-6).transpose(1, 2)
    ht = F.linear(xt, W(P + ".proj_in.weight").view(C, C), W(P + ".proj_in.bias"))
    nt = F.layer_norm(ht, (C,), W(T + ".norm1.weight"), W(T + ".norm1.bias"), 1e-5)
    qt = F.linear(nt, W(T + ".attn1.to_q.weight")) * ((C // H) ** -0.5 * 1.4426950408889634)
    kt = F.linear(nt, W(T + ".attn1.to_k.weight")); vtt = F.linear(nt, W(T + ".attn1.to_v.weight")).transpose(1, 2)
    bar = 2e-2 if dtype == torch.bfloat16 else 3e-3
    for name, got, ref, tt in (("h", h, h_ref, ht), ("q", qk[..., :C], qk_ref[..., :C], qt), ("k", qk[..., C:], qk_ref[..., C:], kt), ("vt", vt, vt_ref, vtt)):
        sc = tt.abs().max()
        e_f, e_p = (got.float() - tt).abs().max() / sc, (ref.float() - tt).abs().max() / sc
        within("e_f", e_f, bar, strict=True); within("e_f", e_f, 2 * e_p + 1e-3, strict=True)


def test_head_to_tail_fragment_layout_is_the_same_function():
    """h handed from the head kernel to the tail kernel as MFMA fragments (h_fragment_layout) or as rows: identical tail output"""
    from gaussctrl_amd.sd import ops, weights
    dev, dtype, B, HW, f, Lt = "cuda:0", torch.bfloat16, 4, 256, 2, 77
    sd = _sd()
    sd[P + ".norm.weight"] = torch.ones(C); sd[P + ".norm.bias"] = torch.zeros(C)
    w = weights.prepare(sd, dtype, dev, heads=H)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, HW, C, generator=g).to(dtype).to(dev); o1 = torch.randn(B, HW, C, generator=g).to(dtype).to(dev)
    ctx = torch.randn(B // f, Lt, CTX, generator=g).to(dtype).to(dev)
    k = ops.linear(ctx, w[T + ".attn2.to_k.weight"])
    vt = torch.zeros(B // f, C, 80, dtype=dtype, device=dev)
    ops.linear(ctx, w[T + ".attn2.to_v.weight"], want_out=False, rows_per_batch=Lt, out_t=vt, ldt=80, t_batch_stride=C * 80)
    kv = weights.tail_text_stream(k, vt, Lt, H)
    coef = ops.groupnorm_coef(x, w[P + ".norm.weight"], w[P + ".norm.bias"], 32, 1e-6)
    outs = []
    for fr in (False, True):
        h, qk, vtt = ops.transformer_head(x, coef, w[P + ".head.w"], w[P + ".head.params"], h_frags=fr)
        outs.append((ops.transformer_tail(o1, h, x, w[P + ".tail.a"], kv, w[P + ".tail.b"], w[P + ".tail.params"], H, f, Lt, resid_frags=fr), qk, vtt))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])


@pytest.mark.parametrize("dtype,bar", [(torch.float16, 2e-3), (torch.bfloat16, 1.5e-2)])
@pytest.mark.parametrize("mode", ["xview", "plain"])
def test_level0_block_against_the_oracle(dtype, bar, mode):
    """A whole level-0 transformer block in its product wiring -- GroupNorm statistics, one-launch head, the (cross-view) attention kernel,
    one-launch tail -- against the CPU oracle's restatement of the reference block (oracle/sd15_torch.py::transformer: Transformer2DModel
    + BasicTransformerBlock with CrossViewAttnProcessor, utils.py:44-133), SD1.5 level-0 shapes: C = 320, 8 heads, 77 x 768 text."""
    from gaussctrl_amd.sd import unet as U, weights
    from oracle import sd15_torch as sd
    dev = "cuda:0"
    f, Hh, Ww = 5, 8, 16                                        # 2 CFG halves x 5 frames, 128 tokens each
    B = 2 * f
    sdw = _sd(seed=7)
    g = torch.Generator().manual_seed(8)
    sdw[P + ".norm.weight"] = 1 + 0.1 * torch.randn(C, generator=g); sdw[P + ".norm.bias"] = 0.1 * torch.randn(C, generator=g)
    r16 = lambda v: v.to(dtype).float()
    wq = {k: r16(v) for k, v in sdw.items()}                    # the oracle sees the 16-bit weights the kernels get
    x = torch.randn(B, C, Hh, Ww, generator=g) * 1.2
    ctx = torch.randn(2, 77, CTX, generator=g)
    cfg = dict(groups=32, heads=H)
    ref = sd.transformer(wq, P, r16(x), r16(ctx).repeat_interleave(f, 0), cfg, mode, 0.6)              # [B, C, H, W]
    w = weights.prepare(sdw, dtype, dev, heads=H)
    w["conv_in.weight"] = torch.zeros(1, dtype=dtype, device=dev)                                       # SDNet reads its dtype here
    net = U.SDNet(w, dict(U.CFG_SD15), "unet")
    assert net.fused_head and net.fused_tail
    net.begin_forward(dev)
    actx = U.AttnCtx(mode, 0.6, f, {}, None, "unet")
    xg = x.permute(0, 2, 3, 1).contiguous().to(dtype).to(dev)
    out, _ = net.transformer(P, xg, None, ctx.to(dtype).to(dev), actx)
    got = out.float().cpu().permute(0, 3, 1, 2)
    rel = float((got - ref).norm() / ref.norm())
    within("rel", rel, bar, strict=True)
    # and the per-op launches give the same answer within the same bar
    net.fused_head = net.fused_tail = False
    out2, _ = net.transformer(P, xg, None, ctx.to(dtype).to(dev), U.AttnCtx(mode, 0.6, f, {}, None, "unet"))
    rel2 = float((out2.float().cpu().permute(0, 3, 1, 2) - ref).norm() / ref.norm())
    within("rel2", rel2, bar, strict=True); within("rel", rel, 2 * rel2 + 1e-4, strict=True)
