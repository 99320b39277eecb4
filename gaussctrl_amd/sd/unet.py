"""SD1.x UNet2DConditionModel + depth ControlNetModel forward graphs on the HIP kernels (channels-last, bf16/f16).

Host-side mirror of what diffusers 0.26.0 executes when the reference calls `self.pipe(...)`
(/root/reference/gaussctrl/gc_pipeline.py:142-145,209-219) with the reference's attention processor
(/root/reference/gaussctrl/utils.py:39-133) installed on every attention layer
(gc_pipeline.py:136-137 plain, :163-168 cross-view).  Layer inventory: SURVEY.md Appendix B.

Differences from the reference's execution that do not change results:
  * one fused multi-K/V attention kernel instead of five materialised probability tensors per layer;
  * ControlNet's self-attention term has weight 0 (gc_pipeline.py:167) and is skipped, not multiplied by 0;
  * text K/V are computed once per CFG half (the prompt is the same for every frame, gc_pipeline.py:210-211)
    and cached per layer for the whole trajectory (they do not depend on t);
  * the conditioning embedding of ControlNet depends only on the disparity image and is computed once per chunk;
  * reference-frame K / V^T can be read from a cache (`RefBank`) instead of being recomputed in every chunk
    (reference frames attend only to reference frames, utils.py:94-117, so their trajectory is chunk independent).
"""
from __future__ import annotations

import math
import os

import torch

from . import ops
from . import weights as weights_mod

CFG_SD15 = dict(block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, heads=8, cross_dim=768,
                groups=32, attn_levels=(True, True, True, False), n_cond_blocks=6)


def timestep_embedding(t: float, dim: int) -> torch.Tensor:
    """diffusers get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0) for one timestep (host)."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half
    emb = float(t) * torch.exp(exponent)
    return torch.cat([torch.cos(emb), torch.sin(emb)])[None, :]


class StatArena:
    """Per-forward scratch for producer-side normalisation statistics (GEMM epilogues / concat accumulate (sum, sum^2) with float
    atomics, so the buffers must start at zero): ONE zeroed fp32 buffer per network forward, bump-allocated; `reset()` is a single
    memset on the launch stream at the start of the forward (the two networks of a denoise step run on two streams: one arena each)."""

    def __init__(self):
        self.buf = None
        self.off = 0
        self.peak = 0

    def reset(self, device):
        need = max(self.peak, 1 << 16)
        if self.buf is None or self.buf.numel() < need or self.buf.device != device:
            self.buf = torch.empty(int(need * 1.25), dtype=torch.float32, device=device)
        self.buf.zero_()
        self.off = 0

    def alloc(self, *shape):
        n = 1
        for v in shape:
            n *= int(v)
        n4 = (n + 3) // 4 * 4
        if self.off + n4 > self.buf.numel():        # first forward of a new shape: grow (a fresh zeroed buffer; earlier views stay valid)
            self.peak = max(self.peak, (self.off + n4) * 2)
            self.buf = torch.zeros(max(self.peak, n4), dtype=torch.float32, device=self.buf.device)
            self.off = 0
        v = self.buf[self.off:self.off + n].view(*shape)
        self.off += n4
        self.peak = max(self.peak, self.off)
        return v


class RefBank:
    """Cache of the reference frames' per-layer K / V^T for every denoise step.
    mode 'record': layers append their K / V^T of the reference batch; mode 'use': layers read them."""

    def __init__(self):
        self.store = {}
        self.mode = "off"
        self.step = 0
        self.shard = None           # mode 'record' on a SHARDED reference trajectory (dist.RefShard): this rank's forward holds only its
                                    # share of the 2 x 4 reference samples; every layer all-gathers K / V^T (the collective of north_star)

    def key(self, layer):
        return (self.step, layer)


class AttnCtx:
    """Per-forward attention configuration (what `set_attn_processor` configures in the reference)."""

    def __init__(self, mode: str, coeff: float, frames_per_half: int, text_kv: dict, bank: RefBank | None = None,
                 net: str = "unet"):
        self.mode = mode                    # "plain" (AttnProcessor) | "xview" (CrossViewAttnProcessor)
        self.coeff = coeff                  # self_attn_coeff
        self.f = frames_per_half            # video_length = B // unet_chunk_size
        self.text_kv = text_kv              # layer prefix -> (K [2,77,C], Vt [2,C,80]) cache
        self.bank = bank
        self.net = net
        # CFG-shared prefix (round 5): with classifier-free guidance the network batch is [unconditional ; conditional] copies of the SAME
        # latents (cat([latents] * 2), gc_pipeline.py:209-219 through the diffusers pipeline): everything before the first text cross-attention
        # -- conv_in, the first resnet, and the first transformer block up to its (cross-view) self-attention -- is identical in the two
        # halves, so it is computed for the first half only and duplicated where the halves diverge.  Exact in real arithmetic; in floating
        # point equal up to the accumulation-order noise a different batch size already causes (bit-identical in batch-invariant mode).
        self.share = False


def dup2(t):
    """[f, ...] -> [2 f, ...]: the two CFG halves of a tensor computed once (one device copy)"""
    return torch.cat([t, t], 0)


class SDNet:
    """Shared machinery of the UNet / ControlNet graphs."""

    def __init__(self, weights: dict, cfg: dict = CFG_SD15, name: str = "unet"):
        self.w = weights
        self.cfg = cfg
        self.name = name
        self.dtype = weights["conv_in.weight"].dtype
        self._temb_cache = {}
        self.qpre = bool(weights.get("_attn_q_prescaled", False))   # softmax scale folded into the Q weights (weights.prepare(heads=))
        self.ln_folded = bool(weights.get("_ln_folded", False))     # LayerNorms folded into their consumer GEMMs (weights.prepare(fold_ln=))
        self.fuse_stats = False                                     # GroupNorm statistics from the producing kernel's epilogue: opt-in
                                                                    # (measured slower than the stand-alone statistics pass)
        self.gn_two_pass = False                                    # stand-alone GroupNorm as group-sums + apply (2 launches, not 3): opt-in,
                                                                    # measured neutral (7.02 vs 6.97 views/s) and it gives up the shifted sums
        self.fp8 = bool(weights.get("_fp8_convs", False))           # resnet 3x3 convs on e4m3 operands (weights.add_fp8_convs)
        self.fp8_a_scale = 127                                      # E8M0 byte of the conv inputs (GroupNorm + SiLU outputs are O(1): 2^0)
        self.fp8_min_hw = ops.OPTIONS.fp8_min_hw                    # smallest map whose resnet convs run on e4m3 (16 x 16: k-sliced k_gemm8q)
        # transformer-block linears of the C = 640 / 1280 levels on e4m3 operands (weights.add_fp8_linears): the three LayerNorms write e4m3,
        # the GEGLU epilogue writes the FF hidden as e4m3; E8M0 bytes of the two activation kinds (LayerNorm outputs, GEGLU hidden): 2^0
        self.fp8_lin = int(weights.get("_fp8_linears", 0))          # bit 0: feed-forward, bit 1: attn2.to_q, bit 2: Q | K | V
        self.fp8_ln_scale = 127
        self.fp8_ff_scale = 127
        # level-0 transformer blocks: everything after the self-attention in ONE launch (ops.transformer_tail, csrc/dn_ttail.hip): 148 us
        # against 225 us for the nine per-op launches at 6 x 4096 tokens, +4.3 % views/s end to end.  ops.KernelOptions.fused_tail.
        self.fused_tail = ops.OPTIONS.fused_tail
        # the same for everything before the self-attention (GroupNorm apply, proj_in, LayerNorm1, Q | K | V: ops.transformer_head, csrc/dn_thead.hip)
        self.fused_head = ops.OPTIONS.fused_head
        # GroupNorm statistics as per-channel partials from the producing conv / linear / concat (ops.ChanParts travel in the `xs` slot)
        self.gn_parts = ops.OPTIONS.gn_parts
        # text cross-attention of the LayerNorm-folded blocks as two GEMMs (SDNet._text_fold)
        self.text_fold = ops.OPTIONS.text_fold
        self.ffout_merge = ops.OPTIONS.ffout_merge     # ... and FF-down + proj_out as one GEMM over [ff | h]
        self._arenas = {}
        self.arena = None

    def begin_forward(self, device):
        """zeroed statistics arena of this forward (one per launch stream: independent trajectories may run on different streams)"""
        key = torch.cuda.current_stream().cuda_stream
        a = self._arenas.get(key)
        if a is None:
            a = self._arenas[key] = StatArena()
        a.reset(device)
        self.arena = a

    def _cs(self, B, C, HW, streaming=False):
        """zeroed [B, G, 2] buffer for a producer's GroupNorm-group sums, or None (the consumer computes the statistics itself).
        GEMM-epilogue producers: only with fuse_stats (opt-in) and when the epilogue's 16-row tiles stay inside one image (HW % 16 == 0);
        streaming producers (the decoder's skip concat, which reads every element anyway): whenever gn_two_pass is on."""
        if streaming:
            return self.arena.alloc(B, self.cfg["groups"], 2) if (self.gn_two_pass or self.fuse_stats) else None
        return self.arena.alloc(B, self.cfg["groups"], 2) if (self.fuse_stats and HW % 16 == 0) else None

    def gn_fp8(self, x, xs, p, eps):
        """GroupNorm + SiLU with an e4m3 output (input of an fp8 convolution): from the producer's group sums when it left them,
        else the stand-alone statistics pass + a quantising apply"""
        w = self.w
        g = self.cfg["groups"]
        if isinstance(xs, ops.ChanParts):    # the producer left its partial sums: one launch
            return ops.groupnorm_apply_parts_fp8(x, xs, w[p + ".weight"], w[p + ".bias"], g, eps, True, self.fp8_a_scale)
        if xs is not None:
            return ops.groupnorm_apply_fp8(x, xs, w[p + ".weight"], w[p + ".bias"], g, eps, True, self.fp8_a_scale)
        return ops.groupnorm_fp8(x, w[p + ".weight"], w[p + ".bias"], g, eps, True, self.fp8_a_scale)

    def gn(self, x, xs, p, eps, silu):
        """GroupNorm(+SiLU): one launch when the producer of x left its channel sums (xs), else the three-kernel stand-alone path"""
        w = self.w
        g = self.cfg["groups"]
        if isinstance(xs, ops.ChanParts):
            return ops.groupnorm(x, w[p + ".weight"], w[p + ".bias"], g, eps, silu, parts=xs)
        if xs is not None:
            return ops.groupnorm_apply(x, xs, w[p + ".weight"], w[p + ".bias"], g, eps, silu)
        if self.gn_two_pass and x.numel() > (1 << 20):      # (<= 2 MB: the one-launch small-map kernel inside ops.groupnorm wins)
            # two launches instead of three: group sums (float atomics into the zeroed arena) + apply with the coefficient prologue
            gs = self.arena.alloc(x.shape[0], g, 2)
            ops.group_stats(x, gs)
            return ops.groupnorm_apply(x, gs, w[p + ".weight"], w[p + ".bias"], g, eps, silu)
        return ops.groupnorm(x, w[p + ".weight"], w[p + ".bias"], g, eps, silu)

    # ---------------------------------------------------------------------------------------- blocks
    def time_embed(self, t: float, device):
        """-> {resnet prefix: fp32 row-vector [1, Cout]} = time_emb_proj(silu(time_embedding(t))) for EVERY resnet of the
        network from one batched GEMM; depends only on t, so it is cached per timestep (20 entries per trajectory)."""
        key = float(t)
        hit = self._temb_cache.get(key)
        if hit is not None:
            return hit
        w = self.w
        e = timestep_embedding(t, self.cfg["block_out_channels"][0]).to(device).to(self.dtype)
        h = ops.linear(e, w["time_embedding.linear_1.weight"], w["time_embedding.linear_1.bias"], act=1)
        h = ops.linear(h, w["time_embedding.linear_2.weight"], w["time_embedding.linear_2.bias"], act=1)
        allv = ops.linear(h, w["_temb_all.weight"], w["_temb_all.bias"], out_f32=True)          # [1, sum Cout]
        out = {n: allv[:, o:o + c] for n, (o, c) in w["_temb_all.index"].items()}
        if len(self._temb_cache) < 256:
            self._temb_cache[key] = out
        return out

    def resnet(self, p, x, xs, temb_act, eps=1e-5):
        """ResnetBlock2D on (x, xs = channel sums of x or None) -> (out, channel sums of out)"""
        w = self.w
        B, HW = x.shape[0], x.shape[1] * x.shape[2]
        rv = None if temb_act is None else temb_act[p]
        cout = w[p + ".conv1.weight"].shape[0]
        hs = self._cs(B, cout, HW)
        g = self.cfg["groups"]
        q8 = self.fp8 and (p + ".conv1.w8") in w and HW >= self.fp8_min_hw      # 8x8 maps: few tiles, long K -> the split-K bf16 kernels
        if q8:        # fp8 path: the GroupNorm writes e4m3, the conv runs on the block-scaled MFMA
            h8 = self.gn_fp8(x, xs, p + ".norm1", eps)
            if self.gn_parts and hs is None:       # (the k-sliced 16 x 16-map problems leave the partials of their output; others return None)
                h, hs = ops.conv3x3_fp8(h8, w[p + ".conv1.w8"], w[p + ".conv1.w8_scale"], x.dtype, w[p + ".conv1.bias"], rowvec=rv, ld_rowvec=0,
                                        a_scale=self.fp8_a_scale, chan_parts=True)
            else:
                h = ops.conv3x3_fp8(h8, w[p + ".conv1.w8"], w[p + ".conv1.w8_scale"], x.dtype, w[p + ".conv1.bias"], rowvec=rv, ld_rowvec=0,
                                    a_scale=self.fp8_a_scale, group_stats=hs)
        else:
            h = self.gn(x, xs, p + ".norm1", eps, True)
            if self.gn_parts and hs is None:
                h, hs = ops.conv3x3(h, w[p + ".conv1.weight"], w[p + ".conv1.bias"], rowvec=rv, ld_rowvec=0, chan_parts=True)
            else:
                h = ops.conv3x3(h, w[p + ".conv1.weight"], w[p + ".conv1.bias"], rowvec=rv, ld_rowvec=0, group_stats=hs)
        sc = x
        if (p + ".conv_shortcut.weight") in w:
            sc = ops.linear(x, w[p + ".conv_shortcut.weight"], w[p + ".conv_shortcut.bias"])
        os_ = self._cs(B, cout, HW)
        if q8:
            h8 = self.gn_fp8(h, hs, p + ".norm2", eps)
            if self.gn_parts and os_ is None:
                return ops.conv3x3_fp8(h8, w[p + ".conv2.w8"], w[p + ".conv2.w8_scale"], x.dtype, w[p + ".conv2.bias"], residual=sc,
                                       a_scale=self.fp8_a_scale, chan_parts=True)
            return ops.conv3x3_fp8(h8, w[p + ".conv2.w8"], w[p + ".conv2.w8_scale"], x.dtype, w[p + ".conv2.bias"], residual=sc,
                                   a_scale=self.fp8_a_scale, group_stats=os_), os_
        h = self.gn(h, hs, p + ".norm2", eps, True)
        if self.gn_parts and os_ is None:
            return ops.conv3x3(h, w[p + ".conv2.weight"], w[p + ".conv2.bias"], residual=sc, chan_parts=True)
        return ops.conv3x3(h, w[p + ".conv2.weight"], w[p + ".conv2.bias"], residual=sc, group_stats=os_), os_

    def _self_attention(self, p, n, actx: AttnCtx, ln=None, dt=None, dup=False):
        """n: LayerNorm-ed tokens, or the raw tokens with ln = (row sums, colsum, eps) when norm1 is folded into the Q|K|V GEMM, or (dt given)
        the e4m3 bytes of the LayerNorm-ed tokens for the fp8 projection (weights.add_fp8_linears)"""
        w = self.w
        heads = self.cfg["heads"]
        B, L, Cc = n.shape
        Lp = (L + 7) // 8 * 8
        adt = dt or n.dtype
        bank = actx.bank
        if (dt is None and actx.mode == "xview" and actx.coeff == 0.0 and bank is not None and bank.mode == "use" and
                getattr(bank, "shard", None) is None and ops.OPTIONS.q_only):
            # ControlNet (self_attn_coeff = 0, utils.py:95-102: the self term has weight 0) against a cached reference bank: the frame's own K / V^T
            # are never read -- project Q only (the first C rows of the fused Q | K | V weight), a third of the GEMM
            kr, vtr = bank.store[bank.key((actx.net, p))]
            wq = w[p + ".to_qkv.weight"][:Cc]
            if ln is not None:
                q = ops.linear(n, wq, w[p + ".to_qkv.bias"][:Cc], ln=(ln[0], ln[1][:Cc], ln[2]))
            else:
                q = ops.linear(n, wq)
            sets = [(r, 1.0 / 4.0) for r in range(4)]
            return ops.attention(q, kr, vtr, heads, sets, actx.f, Lk=L, kref=kr, vtref=vtr, ref_fph=kr.shape[0] // 2, q_prescaled=self.qpre)
        vt = torch.zeros(B, Cc, Lp, dtype=adt, device=n.device) if Lp != L else torch.empty(B, Cc, Lp, dtype=adt, device=n.device)
        # one GEMM for Q | K | V: columns [0,2C) -> qk [B,L,2C], columns [2C,3C) -> V^T [B,C,Lp]
        if dt is not None:
            qk = ops.linear_fp8(n, w[p + ".to_qkv.w8"], w[p + ".to_qkv.w8_scale"], dt, rows_per_batch=L, out_t=vt, ldt=Lp, t_batch_stride=Cc * Lp,
                                t_col0=2 * Cc, out_cols=2 * Cc, a_scale=self.fp8_ln_scale)
        else:
            qk = ops.linear(n, w[p + ".to_qkv.weight"], w.get(p + ".to_qkv.bias") if ln is not None else None, rows_per_batch=L, out_t=vt,
                            ldt=Lp, t_batch_stride=Cc * Lp, t_col0=2 * Cc, out_cols=2 * Cc, ln=ln)
        return self._attend(p, qk[..., :Cc], qk[..., Cc:], vt, actx, dup)

    def _attend(self, p, q, k, vt, actx: AttnCtx, dup=False):
        """the attention processor proper on projected Q / K / V^T (utils.py:60-117)"""
        heads = self.cfg["heads"]
        L = q.shape[1]
        if actx.mode == "plain":
            return ops.attention(q, k, vt, heads, [(-1, 1.0)], actx.f, Lk=L, q_prescaled=self.qpre)
        a = actx.coeff
        sets = ([(-1, a)] if a != 0.0 else []) + [(r, (1.0 - a) / 4.0) for r in range(4)]    # utils.py:95-102,117
        bank = actx.bank
        if bank is not None and bank.mode == "record" and bank.shard is not None:
            # sharded reference trajectory: K / V^T of ALL 2 x 4 reference samples by one all-gather per layer; the local queries attend to
            # the four references of their own CFG half out of the gathered bank (which is, at the same time, the finished bank entry)
            kr, vtr = bank.shard.gather_kv(k, vt)
            bank.store[bank.key((actx.net, p))] = (kr, vtr)
            h0 = bank.shard.half_base                    # first reference row of the half the local batch starts with
            return ops.attention(q, k, vt, heads, sets, actx.f, Lk=L, kref=kr[h0:], vtref=vtr[h0:], ref_fph=4, q_prescaled=self.qpre)
        if bank is not None and bank.mode == "record":
            if dup:     # CFG-shared prefix: K / V^T of the first half serve both halves; the bank keeps its [2 * 4] layout (K with the Q|K row stride)
                Bk, Lk_, Ck = k.shape
                ld = k.stride(1)
                buf = torch.empty(2 * Bk, Lk_, ld, dtype=k.dtype, device=k.device)
                k2 = buf[..., ld - Ck:]
                k2[:Bk].copy_(k); k2[Bk:].copy_(k)
                bank.store[bank.key((actx.net, p))] = (k2, dup2(vt))
            else:
                bank.store[bank.key((actx.net, p))] = (k, vt)            # the batch IS the reference batch [2*4]
        if bank is not None and bank.mode == "use":
            kr, vtr = bank.store[bank.key((actx.net, p))]
            return ops.attention(q, k, vt, heads, sets, actx.f, Lk=L, kref=kr, vtref=vtr, ref_fph=kr.shape[0] // 2, q_prescaled=self.qpre)
        return ops.attention(q, k, vt, heads, sets, actx.f, Lk=L, q_prescaled=self.qpre)

    def _text_kv(self, p, ctx, actx: AttnCtx):
        key = (actx.net, p)
        if key not in actx.text_kv:
            w = self.w
            Bc, Lt, _ = ctx.shape
            Cc = w[p + ".to_k.weight"].shape[0]
            Lp = (Lt + 7) // 8 * 8
            k = ops.linear(ctx, w[p + ".to_k.weight"])
            vt = torch.zeros(Bc, Cc, Lp, dtype=ctx.dtype, device=ctx.device)
            ops.linear(ctx, w[p + ".to_v.weight"], want_out=False, rows_per_batch=Lt, out_t=vt, ldt=Lp, t_batch_stride=Cc * Lp)
            actx.text_kv[key] = (k, vt, Lt)
        return actx.text_kv[key]

    def _text_stream(self, p, ctx, actx: AttnCtx):
        """the text K / V^T of attention layer `p` as MFMA operand blocks per CFG half (stream segment of the fused tail)"""
        key = (actx.net, p, "stream")
        if key not in actx.text_kv:
            k, vt, Lt = self._text_kv(p, ctx, actx)
            actx.text_kv[key] = weights_mod.tail_text_stream(k, vt, Lt, self.cfg["heads"])
        return actx.text_kv[key]

    def _text_fold(self, t, ctx, actx: AttnCtx):
        """The text cross-attention of transformer block `t` folded into two GEMM weight sets per CFG half g:
             scores[m, (h, j)] = LN2(x)[m] . A_g[(h, j)],   A_g[(h, j)] = sum_d K_g[j, h D + d] Wq'[h D + d, :]      (Wq' carries gamma2 and log2(e) / sqrt(D))
             out[m]            = softmax_h(scores)[m] . Bm_g^T + b_o + x[m],   Bm_g[n, (h, j)] = sum_d Wo[n, h D + d] V_g[j, h D + d]
        i.e. attn2.to_q -> attention over the 77 text keys -> attn2.to_out is  GEMM (softmax-heads epilogue) -> GEMM: the attention launch is
        gone and for C = 1280 both GEMMs shrink (640 score columns instead of 1 280 channels).  The text K / V^T are fixed per prompt, so the
        products are formed ONCE per (prompt pair, block) -- by this library's GEMM with an fp32 output -- and rounded to the activation type.
        Returns (A [2, 640, C], a [2, 640] fp32, colsum [2, 640] fp32, Bm [2, C, 640], b_o [2, C] fp32), or None when the block is not eligible."""
        key = (actx.net, t, "fold")
        if key in actx.text_kv:
            return actx.text_kv[key]
        w = self.w
        heads = self.cfg["heads"]
        k, vt, Lt = self._text_kv(t + ".attn2", ctx, actx)          # [2, Lt, C], [2, C, Lp]
        res = None
        if k.shape[0] == 2 and Lt <= 80 and heads == 8 and (t + ".attn2.to_q.colsum") in w and vt.shape[-1] >= 80:
            Cc = k.shape[-1]
            D = Cc // heads
            wq, bq = w[t + ".attn2.to_q.weight"], w[t + ".attn2.to_q.bias"]
            wo, bo = w[t + ".attn2.to_out.0.weight"], w[t + ".attn2.to_out.0.bias"]
            dt, dev = k.dtype, k.device
            A = torch.zeros(2, heads, 80, Cc, dtype=torch.float32, device=dev)
            Bm = torch.zeros(2, Cc, heads, 80, dtype=torch.float32, device=dev)
            for h in range(heads):
                wqT = wq[h * D:(h + 1) * D, :].t().contiguous()                   # [C, D]: W operand of  K_h (Wq'_h)  (layout only)
                woh = wo[:, h * D:(h + 1) * D]                                    # [C, D] view, row stride C
                for g in range(2):
                    A[g, h, :Lt] = ops.linear(k[g, :, h * D:(h + 1) * D], wqT, out_f32=True)                      # [Lt, C]
                    vh = vt[g, h * D:(h + 1) * D, :80].t().contiguous()           # [80, D] (rows >= Lt are the V^T buffer's zero padding)
                    Bm[g, :, h] = ops.linear(woh, vh, out_f32=True)                                              # [C, 80]
            A = A.reshape(2, heads * 80, Cc).to(dt).contiguous()
            Bm = Bm.reshape(2, Cc, heads * 80).to(dt).contiguous()
            a = torch.zeros(2, heads, 80, dtype=torch.float32, device=dev)
            a[:, :, :Lt] = (k.float() * bq).view(2, Lt, heads, D).sum(-1).transpose(1, 2)            # K_g b'  (b' = Wq beta2, prescaled)
            res = (A, a.reshape(2, heads * 80).contiguous(), A.float().sum(-1).contiguous(), Bm,
                   torch.stack([bo, bo]).float().contiguous(), Lt)
        actx.text_kv[key] = res
        return res

    def tail_eligible(self, p, ctx) -> bool:
        """static part of the fused-tail predicate for transformer block `p` (the shape part is checked per call)"""
        return bool(self.fused_tail and (p + ".tail.a") in self.w and (p + ".transformer_blocks.0.attn1.to_qkv.colsum") not in self.w and not self.fuse_stats
                    and ctx.shape[1] <= 96 and ctx.shape[0] <= 2)

    def transformer(self, p, x, xs, ctx, actx: AttnCtx, expand=False):
        """Transformer2DModel on (x, xs) -> (out, channel sums of out).  With folded LayerNorms (weights.prepare(fold_ln=True)) the
        three LayerNorm launches disappear: each producer GEMM leaves the row sums of its output, the consumer GEMM (whose weights
        carry gamma / beta) normalises in its epilogue -- 11 launches per block instead of 16."""
        # expand (CFG-shared prefix, AttnCtx.share): x holds the first CFG half only; GroupNorm, proj_in, norm1, Q | K | V and the self-attention
        # run on it, then (attention output, residual stream, block input) are duplicated and the text-dependent rest runs on both halves
        w = self.w
        B, H, W_, Cc = x.shape
        B0 = B
        if expand:
            B = 2 * B0
        t = p + ".transformer_blocks.0"
        fold = self.ln_folded and (t + ".attn1.to_qkv.colsum") in w          # (prepare(fold_ln=2) folds only the blocks without a fused tail)
        # (the tail kernel reads one text block per CFG half: ctx rows <= 2 and an equal number of frames per row; anything else -- e.g. a
        # direct caller with one ctx row per frame -- takes the per-op path)
        tail = self.tail_eligible(p, ctx) and (H * W_) % 128 == 0 and B % ctx.shape[0] == 0
        hfr = False
        # (C = 640 / 1280 blocks: weights.add_fp8_linears; k_gemm8q has no k-slices, so the few-row problems -- the 8 x 8 maps, where the 5120 -> 1280
        # down projection is 30 tiles of 40 k-steps -- stay on the split-K bf16 kernels, as the small-map convolutions do)
        q8 = self.fp8_lin if (not fold and not tail and (t + ".ff.net.0.proj.w8") in w and B * H * W_ >= 1024) else 0
        if self.fused_head and (p + ".head.w") in w and not fold and (xs is None or isinstance(xs, ops.ChanParts)) and (H * W_) % 128 == 0:
            x3 = x.view(B0, H * W_, Cc)
            coef = ops.groupnorm_coef(x3, w[p + ".norm.weight"], w[p + ".norm.bias"], self.cfg["groups"], 1e-6, parts=xs)
            hfr = tail                       # h goes from one fused kernel to the other: stored as MFMA fragments
            h, qk, vt = ops.transformer_head(x3, coef, w[p + ".head.w"], w[p + ".head.params"], h_frags=hfr)
            o = self._attend(t + ".attn1", qk[..., :Cc], qk[..., Cc:], vt, actx, expand)
        else:
            h = self.gn(x, xs, p + ".norm", 1e-6, False)
            rs = ops.RowStats() if fold else None
            h = ops.linear(h.view(B0, H * W_, Cc), w[p + ".proj_in.weight"], w[p + ".proj_in.bias"], row_stats=rs)
            if fold:
                o = self._self_attention(t + ".attn1", h, actx, ln=(rs, w[t + ".attn1.to_qkv.colsum"], 1e-5), dup=expand)
            elif q8 & 4:
                o = self._self_attention(t + ".attn1", ops.layernorm_fp8(h, w[t + ".norm1.weight"], w[t + ".norm1.bias"], a_scale=self.fp8_ln_scale),
                                         actx, dt=h.dtype, dup=expand)
            else:
                o = self._self_attention(t + ".attn1", ops.layernorm(h, w[t + ".norm1.weight"], w[t + ".norm1.bias"]), actx, dup=expand)
        if tail and expand and not ops.OPTIONS.tail_in_rows:      # (A/B: duplicate the three inputs instead of gc_ttail_desc.in_rows)
            o, h, x, B0, expand = dup2(o), dup2(h), dup2(x), B, False
        if tail:                             # (expand: the tail reads the shared rows for both CFG halves -- gc_ttail_desc.in_rows -- no duplicate copies)
            kv = self._text_stream(t + ".attn2", ctx, actx)
            out = ops.transformer_tail(o, h, x.view(B0, H * W_, Cc), w[p + ".tail.a"], kv, w[p + ".tail.b"], w[p + ".tail.params"],
                                       self.cfg["heads"], B // kv.shape[0], ctx.shape[1], resid_frags=hfr, halves=2 if expand else 1)
            return out.view(B, H, W_, Cc), None
        if expand:                           # the halves diverge at the text cross-attention: both get the shared result
            o, h, x = dup2(o), dup2(h), dup2(x)
        rs = ops.RowStats() if fold else None
        h = ops.linear(o, w[t + ".attn1.to_out.0.weight"], w[t + ".attn1.to_out.0.bias"], residual=h, row_stats=rs)
        # feed-forward down projection + proj_out as ONE GEMM over [ff | h] (weights.prepare: ffout.weight = [Wp Wd | Wp]): the residual stream after
        # attn2 and the GEGLU hidden are written side by side into one [M, 5 C] buffer (ldc = lda = 5 C), proj_out's launch disappears
        fb = None
        if fold and self.ffout_merge and (t + ".ffout.weight") in w and not q8:
            fb = torch.empty(B, H * W_, 5 * Cc, dtype=h.dtype, device=h.device)
        h_out = None if fb is None else fb[..., 4 * Cc:]
        tf = None
        if fold and self.text_fold and not ops.BATCH_INVARIANT:
            Mh = (B // 2) * H * W_
            if B % 2 == 0 and any(Mh % (64 * mt) == 0 for mt in (2, 3, 4)):
                tf = self._text_fold(t, ctx, actx)
        if tf is not None:
            # text cross-attention as two GEMMs (weight set per CFG half): scores + softmax per head in the first one's epilogue (norm2 folded in),
            # probabilities x (Wo V^T) + bias + residual in the second, which also leaves the row partials of norm3
            A, a, acs, Bm, bo2, Lt = tf
            pr = ops.linear(h, A, a, ln=(rs, acs, 1e-5), w_set_rows=Mh, softmax_keys=Lt)
            rs = ops.RowStats()
            h = ops.linear(pr, Bm, bo2, residual=h, row_stats=rs, w_set_rows=Mh, out=h_out)
        elif fold:
            q = ops.linear(h, w[t + ".attn2.to_q.weight"], w[t + ".attn2.to_q.bias"], ln=(rs, w[t + ".attn2.to_q.colsum"], 1e-5))
        elif q8 & 2:
            q = ops.linear_fp8(ops.layernorm_fp8(h, w[t + ".norm2.weight"], w[t + ".norm2.bias"], a_scale=self.fp8_ln_scale),
                               w[t + ".attn2.to_q.w8"], w[t + ".attn2.to_q.w8_scale"], h.dtype, a_scale=self.fp8_ln_scale)
        else:
            q = ops.linear(ops.layernorm(h, w[t + ".norm2.weight"], w[t + ".norm2.bias"]), w[t + ".attn2.to_q.weight"])
        if tf is None:
            k, vt, Lt = self._text_kv(t + ".attn2", ctx, actx)
            # ctx holds one text row per CFG half ([negative || positive]); frame b reads row b // f (kind -2)
            o = ops.attention(q, k, vt, self.cfg["heads"], [(-2, 1.0)], B // k.shape[0], Lk=Lt, q_prescaled=self.qpre)
            rs = ops.RowStats() if fold else None
            h = ops.linear(o, w[t + ".attn2.to_out.0.weight"], w[t + ".attn2.to_out.0.bias"], residual=h, row_stats=rs, out=h_out)
        if fb is not None:
            ops.linear(h, w[t + ".ff.net.0.proj.weight"], w[t + ".ff.net.0.proj.bias"], geglu=True,
                       ln=(rs, w[t + ".ff.net.0.proj.colsum"], 1e-5), out=fb[..., :4 * Cc])
            os_ = self._cs(B, Cc, H * W_)
            if self.gn_parts and os_ is None:
                out, os_ = ops.linear(fb, w[t + ".ffout.weight"], w[t + ".ffout.bias"], residual=x.view(B, H * W_, Cc), rows_per_batch=H * W_,
                                      chan_parts=True)
                return out.view(B, H, W_, Cc), os_
            out = ops.linear(fb, w[t + ".ffout.weight"], w[t + ".ffout.bias"], residual=x.view(B, H * W_, Cc), rows_per_batch=H * W_,
                             group_stats=os_)
            return out.view(B, H, W_, Cc), os_
        if fold:
            ff = ops.linear(h, w[t + ".ff.net.0.proj.weight"], w[t + ".ff.net.0.proj.bias"], geglu=True,
                            ln=(rs, w[t + ".ff.net.0.proj.colsum"], 1e-5))
        elif q8 & 1:      # LayerNorm -> e4m3, GEGLU projection on the block-scaled MFMA writing the hidden as e4m3, down projection on it
            ff = ops.linear_fp8(ops.layernorm_fp8(h, w[t + ".norm3.weight"], w[t + ".norm3.bias"], a_scale=self.fp8_ln_scale),
                                w[t + ".ff.net.0.proj.w8"], w[t + ".ff.net.0.proj.w8_scale"], h.dtype, w[t + ".ff.net.0.proj.bias"], geglu=True,
                                a_scale=self.fp8_ln_scale, out_fp8=self.fp8_ff_scale)
        else:
            ff = ops.linear(ops.layernorm(h, w[t + ".norm3.weight"], w[t + ".norm3.bias"]), w[t + ".ff.net.0.proj.weight"],
                            w[t + ".ff.net.0.proj.bias"], geglu=True)
        if q8 & 1:
            h = ops.linear_fp8(ff, w[t + ".ff.net.2.w8"], w[t + ".ff.net.2.w8_scale"], h.dtype, w[t + ".ff.net.2.bias"], residual=h,
                               a_scale=self.fp8_ff_scale)
        else:
            h = ops.linear(ff, w[t + ".ff.net.2.weight"], w[t + ".ff.net.2.bias"], residual=h)
        os_ = self._cs(B, Cc, H * W_)
        if self.gn_parts and os_ is None:
            out, os_ = ops.linear(h, w[p + ".proj_out.weight"], w[p + ".proj_out.bias"], residual=x.view(B, H * W_, Cc), rows_per_batch=H * W_,
                                  chan_parts=True)
            return out.view(B, H, W_, Cc), os_
        out = ops.linear(h, w[p + ".proj_out.weight"], w[p + ".proj_out.bias"], residual=x.view(B, H * W_, Cc), rows_per_batch=H * W_,
                         group_stats=os_)
        return out.view(B, H, W_, Cc), os_

    def shares_prefix(self, xin, actx: AttnCtx) -> bool:
        """CFG-shared prefix applies: two identical halves in the batch and a transformer block right after the first resnet"""
        return bool(actx.share and xin.shape[0] == 2 * actx.f and self.cfg["attn_levels"][0])

    def encoder(self, x, xs, temb_act, ctx, actx, share=False):
        """(x, xs = channel sums of x) -> (mid-block output, its channel sums, skip tensors).  share: x is the first CFG half only
        (AttnCtx.share); the first transformer block expands it to the full batch."""
        cfg = self.cfg
        skips = [dup2(x) if share else x]
        n = len(cfg["block_out_channels"])
        for i in range(n):
            for j in range(cfg["layers_per_block"]):
                x, xs = self.resnet(f"down_blocks.{i}.resnets.{j}", x, xs, temb_act)
                if cfg["attn_levels"][i]:
                    x, xs = self.transformer(f"down_blocks.{i}.attentions.{j}", x, xs, ctx, actx, expand=share and i == 0 and j == 0)
                skips.append(x)
            if i < n - 1:
                p = f"down_blocks.{i}.downsamplers.0.conv"
                xs = self._cs(x.shape[0], self.w[p + ".weight"].shape[0], (x.shape[1] // 2) * (x.shape[2] // 2))
                if self.gn_parts and xs is None:
                    x, xs = ops.conv3x3(x, self.w[p + ".weight"], self.w[p + ".bias"], stride=2, chan_parts=True)
                else:
                    x = ops.conv3x3(x, self.w[p + ".weight"], self.w[p + ".bias"], stride=2, group_stats=xs)
                skips.append(x)
        x, xs = self.resnet("mid_block.resnets.0", x, xs, temb_act)
        x, xs = self.transformer("mid_block.attentions.0", x, xs, ctx, actx)
        x, xs = self.resnet("mid_block.resnets.1", x, xs, temb_act)
        return x, xs, skips


class ControlNet(SDNet):
    def cond_embedding(self, cond):
        """controlnet_cond_embedding on the [B,H,W,8] (3 used) disparity image -> [B,H/8,W/8,320].  Depends only
        on the control image: computed once per chunk, not per step."""
        w = self.w
        p = "controlnet_cond_embedding"
        c = ops.conv3x3(cond, w[p + ".conv_in.weight"], w[p + ".conv_in.bias"], act=1)
        for k in range(self.cfg["n_cond_blocks"]):
            c = ops.conv3x3(c, w[f"{p}.blocks.{k}.weight"], w[f"{p}.blocks.{k}.bias"], stride=2 if k % 2 == 1 else 1, act=1)
        return ops.conv3x3(c, w[p + ".conv_out.weight"], w[p + ".conv_out.bias"])

    def forward(self, xin, t, ctx, cond_emb, actx: AttnCtx, conditioning_scale=1.0):
        """xin [B,h,w,8] -> (12 down residuals, mid residual), all [B,*,*,C] channels-last."""
        w = self.w
        self.begin_forward(xin.device)
        temb_act = self.time_embed(t, xin.device)
        share = self.shares_prefix(xin, actx)
        if share:
            xin, cond_emb = xin[:actx.f], cond_emb[:actx.f]
        xs = self._cs(xin.shape[0], w["conv_in.weight"].shape[0], xin.shape[1] * xin.shape[2])
        if self.gn_parts and xs is None:
            x, xs = ops.conv3x3(xin, w["conv_in.weight"], w["conv_in.bias"], residual=cond_emb, chan_parts=True)
        else:
            x = ops.conv3x3(xin, w["conv_in.weight"], w["conv_in.bias"], residual=cond_emb, group_stats=xs)
        x, _, skips = self.encoder(x, xs, temb_act, ctx, actx, share)
        down = []
        for n, s in enumerate(skips):
            B, H, W_, Cc = s.shape
            o = ops.linear(s.view(B, H * W_, Cc), w[f"controlnet_down_blocks.{n}.weight"], w[f"controlnet_down_blocks.{n}.bias"],
                           scale=conditioning_scale)
            down.append(o.view(B, H, W_, Cc))
        B, H, W_, Cc = x.shape
        mid = ops.linear(x.view(B, H * W_, Cc), w["controlnet_mid_block.weight"], w["controlnet_mid_block.bias"],
                         scale=conditioning_scale).view(B, H, W_, Cc)
        return down, mid


class UNet(SDNet):
    def encode(self, xin, t, ctx, actx: AttnCtx):
        """conv_in + down blocks + mid block: everything that does not need the ControlNet residuals."""
        w = self.w
        self.begin_forward(xin.device)
        temb_act = self.time_embed(t, xin.device)
        share = self.shares_prefix(xin, actx)
        if share:
            xin = xin[:actx.f]
        xs = self._cs(xin.shape[0], w["conv_in.weight"].shape[0], xin.shape[1] * xin.shape[2])
        if self.gn_parts and xs is None:
            x, xs = ops.conv3x3(xin, w["conv_in.weight"], w["conv_in.bias"], chan_parts=True)
        else:
            x = ops.conv3x3(xin, w["conv_in.weight"], w["conv_in.bias"], group_stats=xs)
        x, _, skips = self.encoder(x, xs, temb_act, ctx, actx, share)
        return x, skips, temb_act

    def decode(self, x, skips, temb_act, ctx, down_res, mid_res, actx: AttnCtx):
        """mid residual add + up blocks + conv_out -> eps fp32 [B,h,w,8] (channels 0..3 valid)."""
        w = self.w
        cfg = self.cfg
        if mid_res is not None:
            x = ops.axpby(x, 1.0, mid_res, 1.0)
        n = len(cfg["block_out_channels"])
        rev_attn = list(reversed(cfg["attn_levels"]))
        for i in range(n):
            for j in range(cfg["layers_per_block"] + 1):
                s = skips.pop()
                r = down_res.pop() if down_res is not None else None
                xs = self._cs(x.shape[0], x.shape[-1] + s.shape[-1], 16, streaming=True)     # the concat streams every element anyway
                if self.gn_parts and xs is None:
                    x, xs = ops.concat_add(x, s, r, chan_parts=True)   # cat([x, skip + controlnet residual]) + per-channel partials of it
                else:
                    x = ops.concat_add(x, s, r, group_stats=xs)   # cat([x, skip + controlnet residual]) + its channel sums
                x, xs = self.resnet(f"up_blocks.{i}.resnets.{j}", x, xs, temb_act)
                if rev_attn[i]:
                    x, xs = self.transformer(f"up_blocks.{i}.attentions.{j}", x, xs, ctx, actx)
            if i < n - 1:
                p = f"up_blocks.{i}.upsamplers.0.conv"
                x = ops.conv3x3(x, w[p + ".weight"], w[p + ".bias"], upsample=True)
        x = self.gn(x, xs, "conv_norm_out", 1e-5, True)
        return ops.conv3x3(x, w["conv_out.weight"], w["conv_out.bias"], out_f32=True)

    def forward(self, xin, t, ctx, down_res, mid_res, actx: AttnCtx):
        """xin [B,h,w,8] -> eps fp32 [B,h,w,8] (channels 0..3 valid)."""
        x, skips, temb_act = self.encode(xin, t, ctx, actx)
        return self.decode(x, skips, temb_act, ctx, down_res, mid_res, actx)
