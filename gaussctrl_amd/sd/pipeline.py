"""Denoise pipeline: the 20-step ControlNet + UNet loop with CFG and DDIM that the reference drives through
diffusers' StableDiffusionControlNetPipeline.__call__ (/root/reference/gaussctrl/gc_pipeline.py:142-145 for
DDIM inversion, :209-219 for the cross-view edit), on the HIP kernels.

Everything stays on the device between steps (the reference stages latents / depth through CPU numpy,
gc_pipeline.py:268-274,186-187,200-204).  Text embeddings are inputs: the CLIP text tower runs once per
prompt, outside the hot path.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from .unet import AttnCtx, ControlNet, RefBank, UNet
from .vae import VAEDecoder


class DDIMSchedule:
    """DDIMScheduler / DDIMInverseScheduler constants of the SD1.x scheduler_config."""

    def __init__(self, num_train: int = 1000):
        betas = np.linspace(0.00085 ** 0.5, 0.012 ** 0.5, num_train, dtype=np.float32) ** 2
        self.alphas_cumprod = np.cumprod(1.0 - betas).astype(np.float32)
        self.final_alpha = float(self.alphas_cumprod[0])
        self.num_train = num_train

    def timesteps(self, n: int, inverse: bool = False):
        ratio = self.num_train // n
        ts = [i * ratio + 1 for i in range(n)]              # leading spacing, steps_offset 1
        return ts if inverse else ts[::-1]

    def alphas(self, t: int, n: int, inverse: bool = False):
        """(alpha of the state the step starts from, alpha of the state it produces)."""
        ratio = self.num_train // n
        prev = t - ratio
        a_prev = float(self.alphas_cumprod[prev]) if prev >= 0 else self.final_alpha
        a_t = float(self.alphas_cumprod[t])
        return (a_prev, a_t) if inverse else (a_t, a_prev)


def to_nhwc8(x_nchw: torch.Tensor, dtype) -> torch.Tensor:
    """[B,C<=8,H,W] float -> [B,H,W,8] activation dtype, zero padded channels (layout plumbing)."""
    B, Cc, H, W = x_nchw.shape
    out = torch.zeros(B, H, W, 8, dtype=dtype, device=x_nchw.device)
    out[..., :Cc] = x_nchw.permute(0, 2, 3, 1).to(dtype)
    return out


class DenoisePipeline:
    def __init__(self, unet_w: dict, controlnet_w: dict, vae_w: dict | None, num_inference_steps: int = 20,
                 guidance_scale: float = 5.0, controlnet_conditioning_scale: float = 1.0):
        self.unet = UNet(unet_w, name="unet")
        self.controlnet = ControlNet(controlnet_w, name="controlnet")
        self.vae = VAEDecoder(vae_w) if vae_w is not None else None
        self.dtype = self.unet.dtype
        self.sched = DDIMSchedule()
        self.n = num_inference_steps
        self.guidance = guidance_scale
        self.cn_scale = controlnet_conditioning_scale
        self.text_kv = {}          # per-layer text K / V^T cache, valid for one (negative, positive) prompt pair
        self._text_key = None
        self.two_streams = ops.OPTIONS.two_streams     # ControlNet || UNet encoder on two HIP streams
        self._side = {}

    def _side_stream(self, dev):
        """the ControlNet stream that pairs with the CURRENT stream (independent trajectories may run on different streams)"""
        key = (dev, torch.cuda.current_stream().cuda_stream)
        st = self._side.get(key)
        if st is None:
            st = self._side[key] = torch.cuda.Stream(device=dev)
        return st

    def _ctx(self, ctx_neg, ctx_pos):
        ctx = torch.cat([ctx_neg, ctx_pos], 0).to(self.dtype).contiguous() if ctx_neg is not None else ctx_pos.to(self.dtype).contiguous()
        key = (tuple(ctx.shape), str(ctx.dtype), float(ctx.float().abs().sum()), float(ctx.float()[..., ::7].sum()))   # content, not address
        if key != self._text_key:
            self.text_kv = {}
            self._text_key = key
            self._ctx_tensor = ctx
        return self._ctx_tensor

    def warm_caches(self, ctx_neg, ctx_pos, steps: int | None = None, inverse: bool = False):
        """Fill, on the CURRENT stream, every lazily built cache a trajectory with these prompts reads: per-layer text K / V^T (and the
        fused tail's text operand streams) of both networks, and the time-embedding rows of every timestep.  A caller that then runs
        several trajectories on independent streams (GaussCtrlPipeline.edit_images, bench.py) orders those streams after this point and
        no stream ever reads a cache entry another stream is still writing -- with a received reference bank nothing else would have
        filled them before the first chunk (the bank owner fills them while it records the bank)."""
        ctx = self._ctx(ctx_neg, ctx_pos)
        ts = self.sched.timesteps(self.n, inverse)
        ts = ts if steps is None else ts[:steps]
        for net in (self.unet, self.controlnet):
            a = AttnCtx("plain", 0.0, 1, self.text_kv, None, net.name)
            for key in list(net.w):
                if isinstance(key, str) and key.endswith(".attn2.to_k.weight"):
                    p = key[:-len(".to_k.weight")]
                    net._text_kv(p, ctx, a)
                    if net.tail_eligible(p[:-len(".transformer_blocks.0.attn2")], ctx):
                        net._text_stream(p, ctx, a)
            for t in ts:
                net.time_embed(t, ctx.device)
        return ctx

    # ---------------------------------------------------------------------------------- core loop
    def _begin(self, latents, disparity, ctx, cfg: bool, mode: str, coeff_unet: float, coeff_cn: float,
               guidance: float, inverse: bool, steps: int | None, bank: RefBank | None, fph: int, rep: int | None = None) -> dict:
        """State of one denoise trajectory (advanced by _advance): latents fp32 [f,4,h,w]; disparity fp32 [f,3,H,W].
        rep: copies of the frames in the network batch (default 2 with CFG; 1 on a rank that holds ONE CFG half of a sharded trajectory)."""
        rep = (2 if cfg else 1) if rep is None else rep
        lat = latents.permute(0, 2, 3, 1).contiguous().float()            # master copy fp32 [f,h,w,4]
        xin = to_nhwc8(latents, self.dtype)
        xin = torch.cat([xin] * rep, 0).contiguous()                        # cat([latents]*2)
        # control image: [f,3,H,W] float (reference layout) or already channels-last [f,H,W,8] in the activation dtype
        cond = disparity if (disparity.dim() == 4 and disparity.shape[-1] == 8 and disparity.dtype == self.dtype) else to_nhwc8(disparity, self.dtype)
        cemb = self.controlnet.cond_embedding(cond)                         # once per chunk
        cemb = torch.cat([cemb] * rep, 0).contiguous() if rep > 1 else cemb
        ts = self.sched.timesteps(self.n, inverse)
        ts = ts if steps is None else ts[:steps]
        return dict(lat=lat, xin=xin, cemb=cemb, ts=ts, i=0, ctx=ctx, cfg=cfg, mode=mode, cu=coeff_unet, cc=coeff_cn, g=guidance,
                    inverse=inverse, bank=bank, fph=fph, rep=rep, dev=latents.device)

    def _advance(self, st: dict, nsteps: int | None = None) -> bool:
        """Run up to `nsteps` more DDIM steps of the trajectory (all remaining when None); True when it is complete."""
        ts, bank, dev = st["ts"], st["bank"], st["dev"]
        end = len(ts) if nsteps is None else min(len(ts), st["i"] + nsteps)
        xin, lat, ctx, cemb = st["xin"], st["lat"], st["ctx"], st["cemb"]
        while st["i"] < end:
            i = st["i"]; t = ts[i]
            if bank is not None:
                bank.step = i
            tkv = st.get("text_kv", self.text_kv)          # (a sharded trajectory on ONE CFG half keeps its own one-row text K / V^T cache)
            a_cn = AttnCtx(st["mode"], st["cc"], st["fph"], tkv, bank, "controlnet")
            a_un = AttnCtx(st["mode"], st["cu"], st["fph"], tkv, bank, "unet")
            # CFG-shared prefix (AttnCtx.share): both halves of the batch are copies of the same latents (rep = 2) -- not on a sharded
            # reference trajectory (its ranks hold other sample subsets) and not with the group-statistics experiment paths
            a_cn.share = a_un.share = bool(ops.OPTIONS.cfg_share and st["cfg"] and st["rep"] == 2 and st["mode"] == "xview" and
                                           (bank is None or getattr(bank, "shard", None) is None))
            if self.two_streams:
                # The ControlNet and the UNet encoder + mid block only share their input: run them on two HIP streams so that
                # the part-filled grids of the 16x16 / 8x8 layers and every kernel's fill / epilogue phase overlap with the
                # other network's work; the UNet decoder joins both.
                main = torch.cuda.current_stream()
                side = self._side_stream(dev)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    down, mid = self.controlnet.forward(xin, t, ctx, cemb, a_cn, self.cn_scale)
                x, skips, temb = self.unet.encode(xin, t, ctx, a_un)
                main.wait_stream(side)
                # (the join also publishes everything the side stream cached in this step -- text K / V^T, time-embedding rows --
                # on `main`: a caller that later switches streams and orders the new one after `main`, the usual torch contract,
                # is ordered after those writes too; nothing produced on `side` is ever consumed before a join)
                eps = self.unet.decode(x, skips, temb, ctx, down, mid, a_un)
            else:
                down, mid = self.controlnet.forward(xin, t, ctx, cemb, a_cn, self.cn_scale)
                eps = self.unet.forward(xin, t, ctx, down, mid, a_un)
            a_from, a_to = self.sched.alphas(t, self.n, st["inverse"])
            if st.get("shard") is not None and st["rep"] == 1:
                eps = st["shard"].gather_eps_pairs(eps)         # this rank holds one CFG half: the partner half's eps of the same frames
            ops.cfg_ddim_step(eps, lat, xin, st["g"], st["cfg"], a_from, a_to, st["rep"])
            st["i"] = i + 1
            if st.get("on_step") is not None:          # parity tests read the latents after every DDIM step
                st["on_step"](i, lat)
        return st["i"] >= len(ts)

    def _denoise(self, latents, disparity, ctx, cfg: bool, mode: str, coeff_unet: float, coeff_cn: float,
                 guidance: float, inverse: bool, steps: int | None, bank: RefBank | None, fph: int, on_step=None):
        """Returns latents fp32 [f,4,h,w].  on_step(i, latents fp32 [f,h,w,4]) is called after every DDIM step."""
        st = self._begin(latents, disparity, ctx, cfg, mode, coeff_unet, coeff_cn, guidance, inverse, steps, bank, fph)
        st["on_step"] = on_step
        self._advance(st)
        return st["lat"].permute(0, 3, 1, 2).contiguous()

    # ---------------------------------------------------------------------------------- public API
    def edit_chunk(self, latents, disparity, ctx_neg, ctx_pos, steps=None, on_step=None):
        """Reference-faithful chunk: `latents` / `disparity` hold the 4 reference frames FIRST, then the chunk
        (gc_pipeline.py:206-219); every frame attends to frames 0..3 of its CFG half."""
        ctx = self._ctx(ctx_neg, ctx_pos)
        return self._denoise(latents, disparity, ctx, True, "xview", 0.6, 0.0, self.guidance, False, steps, None,
                             latents.shape[0], on_step)

    def build_ref_bank(self, ref_latents, ref_disparity, ctx_neg, ctx_pos, steps=None) -> RefBank:
        """Run the 4 reference frames once and keep every layer's K / V^T for every step."""
        ctx = self._ctx(ctx_neg, ctx_pos)
        bank = RefBank()
        bank.mode = "record"
        self._denoise(ref_latents, ref_disparity, ctx, True, "xview", 0.6, 0.0, self.guidance, False, steps, bank,
                      ref_latents.shape[0])
        bank.mode = "use"
        return bank

    def begin_ref_bank(self, ref_latents, ref_disparity, ctx_neg, ctx_pos, steps=None) -> dict:
        """Incremental form of build_ref_bank: returns a trajectory; `advance_ref_bank(tr, n)` runs n DDIM steps of it and returns
        the finished RefBank (else None).  Lets a caller that streams scenes spread the NEXT scene's reference trajectory over the
        chunks of the current one instead of stalling on it."""
        ctx = self._ctx(ctx_neg, ctx_pos)
        bank = RefBank()
        bank.mode = "record"
        return self._begin(ref_latents, ref_disparity, ctx, True, "xview", 0.6, 0.0, self.guidance, False, steps, bank,
                           ref_latents.shape[0])

    def begin_ref_bank_sharded(self, ref_latents, ref_disparity, ctx_neg, ctx_pos, shard, steps=None) -> dict:
        """The reference trajectory SHARDED over the ranks (north_star: "RCCL all-gather of reference-view K/V"): of the 2 CFG halves x 4
        reference frames, this rank runs the samples `shard` (dist.RefShard) assigns to it; every cross-view attention layer all-gathers
        K / V^T, so each rank ends with the complete bank and nobody carries the reference work alone.  ref_latents / ref_disparity hold
        all 4 frames (cheap to have everywhere); advance with advance_ref_bank.  Same result as begin_ref_bank (bit-identical in
        batch-invariant mode: every kernel then accumulates per frame)."""
        ctx = torch.cat([ctx_neg, ctx_pos], 0).to(self.dtype).contiguous()
        bank = RefBank()
        bank.mode = "record"
        bank.shard = shard
        fr = shard.frames
        tr = self._begin(ref_latents[fr], ref_disparity[fr], ctx[shard.halves[0]:shard.halves[-1] + 1].contiguous(), True, "xview", 0.6, 0.0,
                         self.guidance, False, steps, bank, len(fr), rep=len(shard.halves))
        tr["shard"] = shard
        if len(shard.halves) == 2:
            tr["ctx"] = self._ctx(ctx_neg, ctx_pos)         # (resets the shared text K / V^T cache if it belongs to another prompt pair)
            tr["text_kv"] = self.text_kv
        else:
            tr["text_kv"] = {}                              # one CFG half: a one-row text cache of its own
        return tr

    def advance_ref_bank(self, tr: dict, nsteps: int | None = None):
        if self._advance(tr, nsteps):
            tr["bank"].mode = "use"
            tr["bank"].shard = None
            return tr["bank"]
        return None

    def edit_chunk_cached(self, latents, disparity, ctx_neg, ctx_pos, bank: RefBank, steps=None, on_step=None):
        """Chunk frames only; reference K / V^T come from `bank` (same result as edit_chunk()[4:])."""
        ctx = self._ctx(ctx_neg, ctx_pos)
        return self._denoise(latents, disparity, ctx, True, "xview", 0.6, 0.0, self.guidance, False, steps, bank,
                             latents.shape[0], on_step)

    def invert(self, latents, disparity, ctx_pos, steps=None, on_step=None):
        """DDIM inversion with plain attention, guidance 0 -> no CFG batch (gc_pipeline.py:136-145), batched over views."""
        ctx = self._ctx(None, ctx_pos)
        return self._denoise(latents, disparity, ctx, False, "plain", 0.0, 0.0, 0.0, True, steps, None, latents.shape[0], on_step)

    @staticmethod
    def decode_group(h, w):
        """frames of one VAE decode pass at h x w latents: [f, 8h, 8w, 512] elements stay below 2^31 (32-bit element offsets in the kernels)"""
        return max(1, ((1 << 31) - 1) // (64 * h * w * 512))

    def decode(self, latents):
        """latents fp32 [f,4,h,w] -> images fp32 [f,3,8h,8w] in [0,1] (vae.decode(z/0.18215); (x/2+0.5).clamp(0,1))."""
        z = to_nhwc8(latents / 0.18215, self.dtype)
        # the GEMM / conv kernels index an operand with 32-bit element offsets: the decoder's widest full-resolution maps ([f, 8h, 8w, <= 512]) bound the
        # frames of one pass (15 at 512 x 512); larger batches decode in groups (frames are independent: GroupNorm is per sample)
        f, h, w = z.shape[0], z.shape[1], z.shape[2]
        grp = self.decode_group(h, w)
        if f <= grp:
            img = self.vae.decode(z, postprocess=True)[..., :3]
        else:
            img = torch.cat([self.vae.decode(z[i:i + grp], postprocess=True)[..., :3] for i in range(0, f, grp)], 0)
        return img.permute(0, 3, 1, 2).contiguous()
