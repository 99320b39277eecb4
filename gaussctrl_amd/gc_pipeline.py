"""GaussCtrlPipeline: drop-in for /root/reference/gaussctrl/gc_pipeline.py on the HIP kernels.

Same surface: ctor (config, device, test_mode, world_size, local_rank, grad_scaler); attributes
datamanager / model / _model / test_mode / config.render_rate; methods render_reverse() (:122-157),
edit_images() (:159-237), image2latent (:239-246), depth2disparity (:248-266), update_datasets (:268-274),
get_train_loss_dict(step) (:276-287); forward() raises (:289-291).  Config flags of :48-73 keep their names.

What is done differently (results identical up to arithmetic precision):
  * every tensor stays on the GPU between the phases (the reference stages through CPU numpy, :268-274,186-204);
  * DDIM inversion of all views is batched (the reference runs batch 1 x V, :124-145);
  * the 4 reference views' denoise trajectory runs ONCE per scene and its per-layer K / V^T are cached
    (the reference re-denoises and re-decodes the references in every chunk and discards them, :206-219);
  * with world_size > 1 the views are sharded over ranks (view v belongs to rank v % world_size), the reference
    K / V cache is replicated (no data-path collective) and the edited images are all-gathered at the end so every
    rank trains on the full edited set (the reference's `pipe_device = 'cuda:0'` cannot shard at all, :96,102).
"""
from __future__ import annotations

import random
from dataclasses import dataclass, field
from typing import Literal, Optional

import torch
from torch import nn

from .gc_datamanager import GaussCtrlDataManager, GaussCtrlDataManagerConfig, SimpleDataManager  # noqa: F401
from .gc_model import GaussCtrlModel, GaussCtrlModelConfig
from .ns_compat import HAVE_NERFSTUDIO, Cameras
from .sd import arch, ops as sdops
from .sd.pipeline import DenoisePipeline, to_nhwc8
from .sd.vae import VAEEncoder, prepare_vae_encoder_weights, prepare_vae_weights
from .sd.weights import prepare

if HAVE_NERFSTUDIO:  # the reference's bases (gc_pipeline.py:48,76): VanillaPipelineConfig / VanillaPipeline
    from nerfstudio.pipelines.base_pipeline import VanillaPipeline as _PipelineBase, VanillaPipelineConfig as _PipelineConfigBase  # type: ignore
else:
    _PipelineBase = nn.Module

    @dataclass
    class _PipelineConfigBase:
        def setup(self, **kw):
            return self._target(self, **kw)


@dataclass
class GaussCtrlPipelineConfig(_PipelineConfigBase):
    """gc_pipeline.py:48-73 (names and defaults kept)."""
    _target: type = field(default_factory=lambda: GaussCtrlPipeline)
    datamanager: GaussCtrlDataManagerConfig = field(default_factory=GaussCtrlDataManagerConfig)
    model: GaussCtrlModelConfig = field(default_factory=GaussCtrlModelConfig)
    render_rate: int = 500
    edit_prompt: str = ""
    reverse_prompt: str = ""
    langsam_obj: str = ""
    guidance_scale: float = 5
    num_inference_steps: int = 20
    chunk_size: int = 5
    ref_view_num: int = 4
    diffusion_ckpt: str = "CompVis/stable-diffusion-v1-4"
    # --- additions of this implementation
    controlnet_ckpt: str = "lllyasviel/sd-controlnet-depth"      # hard-coded in the reference (:100)
    dtype: str = "f16"                 # the reference runs fp16 (:101); "bf16" is the throughput default of bench.py
    fp8: int = 0                       # BASELINE configs[3] "fp8 MFMA UNet path" (with dtype "bf16" / "f16" for everything else): 1 = resnet 3x3
                                       # convolutions on OCP e4m3 operands (block-scaled MFMA), 2 = also the transformer linears of the C = 640 /
                                       # 1280 levels (LayerNorm and GEGLU write e4m3); latents within 6e-2 relative L2 of the fp32 oracle
    cache_reference_kv: bool = True
    ref_bank_owner: int = -1           # world_size > 1: rank that computes the reference trajectory and broadcasts its K / V^T step
                                       # by step (-1: every rank computes it itself -- no data-path collective)
    ref_bank_allgather: bool = False   # world_size in {2, 4, 8}: shard the reference trajectory itself by sample; every cross-view attention
                                       # layer all-gathers K / V^T (dist.RefShard) -- no owner, no extra work on any rank; wins over ref_bank_owner
    inflight_chunks: int = 2           # chunks of edit_images in flight on independent HIP stream pairs (consecutive chunks only share the
                                       # read-only reference bank; 1 = strictly one after the other, as the reference runs them)
    chunks_per_launch: int = 4         # with the reference K / V^T cached a view's result does not depend on which views share its network batch
                                       # (each view attends to itself and to the 4 references, utils.py:95-102), so `chunks_per_launch` consecutive
                                       # chunks of `chunk_size` views run as ONE batch: every GEMM sees that many times the rows (the levels-1..3
                                       # linears of a 3-view chunk are launch-granularity bound).  1 = one chunk per network batch, as the reference
                                       # runs them; ignored (= 1) without cache_reference_kv, where the references ride in every batch
    round_like_reference: bool = False  # True: round the rendered rgb / depth to fp16 before inversion, disparity and the mask composite,
                                       # exactly where the reference does (gc_pipeline.py:132-133,155); False keeps the fp32 renders
    batch_invariant: Optional[bool] = None    # kernel planning that makes a view's result independent of its chunk / rank count
                                       # (sd.ops.KernelOptions.batch_invariant): True / False set it explicitly, None keeps the process-wide setting
    kernel_options: Optional[object] = None   # a gaussctrl_amd.sd.ops.KernelOptions replacing the process-wide kernel switches (None: keep)
    render_batch: int = 8              # render_reverse renders this many cameras per batched launch set (GaussCtrlModel.get_outputs_for_cameras;
                                       # 1 = one camera per call, as the reference does at gc_pipeline.py:124-130; per view the results are identical)
    train_mode: str = "parity"         # world_size > 1: "parity" = every rank takes the SAME single-view step on replicated
                                       # parameters (the reference's 500 single-view iterations, gc_trainer.py:186-201; no gradient collective:
                                       # replicas stay bit-identical), "throughput" = each rank renders its own view of the step, the N x 59
                                       # gradients are averaged by one flat in-place RCCL all-reduce (dist.FlatGrads) -- an N-view batch per step;
                                       # "sharded" = the same batch with reduce-scatter -> Adam on each rank's 1 / N slice of ONE flat parameter
                                       # buffer (optimizer state 1 / N per rank) -> in-place all-gather (dist.ShardedAdam)
    fold_layernorms: bool = True       # LayerNorms of the C = 640 / 1280 transformer blocks folded into their producer / consumer GEMM epilogues
                                       # (weights.prepare(fold_ln=2): +2 % views/s, same latents to the storage type's rounding); ignored with
                                       # fp8 >= 2, whose linears take e4m3 activations from the LayerNorm kernel
    synthetic_weights: bool = False    # True: seeded random SD1.5-shaped weights + hashed prompt embeddings (bench / tests; there
                                       # are no checkpoints on the build machines).  False: checkpoints are REQUIRED -- no silent fallback.


class GaussCtrlPipeline(_PipelineBase):
    config: GaussCtrlPipelineConfig

    def __init__(self, config: GaussCtrlPipelineConfig, device: str, test_mode: Literal["test", "val", "inference"] = "val",
                 world_size: int = 1, local_rank: int = 0, grad_scaler=None, *, datamanager=None, model=None,
                 diffusion_weights: Optional[dict] = None, text_encoder=None, mask_fn=None):
        if HAVE_NERFSTUDIO and datamanager is None:
            super().__init__(config, device, test_mode, world_size, local_rank)      # builds datamanager + _model from the config tree
        else:
            nn.Module.__init__(self)
            self.config = config
            self.datamanager = datamanager
            self._model = model
        # nerfstudio's Pipeline.device is a read-only property (= model.device): keep our own handle under another name
        self._dev = torch.device(device)
        self.test_mode = test_mode
        self.world_size, self.local_rank = world_size, local_rank
        self.edit_prompt, self.reverse_prompt = config.edit_prompt, config.reverse_prompt
        added_prompt = "best quality, extremely detailed"                                        # :104-107
        self.positive_prompt = self.edit_prompt + ", " + added_prompt
        self.positive_reverse_prompt = self.reverse_prompt + ", " + added_prompt
        self.negative_prompts = ("longbody, lowres, bad anatomy, bad hands, missing fingers, extra digit, fewer digits, "
                                 "cropped, worst quality, low quality")
        view_num = len(self.datamanager.cameras)                                                # :109-114
        anchors = [(view_num * i) // config.ref_view_num for i in range(config.ref_view_num)] + [view_num]
        random.seed(13789)
        # the reference's randint upper bound is inclusive and can return view_num: clamp
        self.ref_indices = [min(random.randint(a, anchors[i + 1]), view_num - 1) for i, a in enumerate(anchors[:-1])]
        self.num_ref_views = len(self.ref_indices)
        if self.num_ref_views != 4:
            raise ValueError("the reference's attention processor hard-wires exactly 4 reference views (utils.py:95-102)")
        self.num_inference_steps, self.guidance_scale = config.num_inference_steps, config.guidance_scale
        self.controlnet_conditioning_scale, self.eta, self.chunk_size = 1.0, 0.0, config.chunk_size
        self.dtype = torch.float16 if config.dtype == "f16" else torch.bfloat16
        dev = self._dev
        self.weights_source = {}
        w = diffusion_weights
        if w is None and not config.synthetic_weights:                 # the reference's from_pretrained calls (:97-102)
            from .sd.checkpoint import load_diffusion_weights, load_text_encoder
            w = load_diffusion_weights(config.diffusion_ckpt, config.controlnet_ckpt)
            if text_encoder is None:
                text_encoder = load_text_encoder(config.diffusion_ckpt)
            self.weights_source = {k: config.controlnet_ckpt if k == "controlnet" else config.diffusion_ckpt for k in w}
        w = w or {}

        def get(name, shapes, seed):
            sd = w.get(name)
            if sd is None:
                if not config.synthetic_weights:
                    raise KeyError(f"diffusion_weights has no '{name}' state dict (have {sorted(w)}); seeded random weights are used "
                                   "only with GaussCtrlPipelineConfig.synthetic_weights=True")
                self.weights_source[name] = f"synthetic(seed={seed})"
                return arch.random_state_dict(shapes, seed, dev)
            arch.check_state_dict(sd, shapes)
            self.weights_source.setdefault(name, "caller-supplied state dict")
            return sd
        if config.kernel_options is not None:
            sdops.configure(options=config.kernel_options)
        if config.batch_invariant is not None:          # explicit both ways: a later pipeline with False resets what an earlier one set
            sdops.configure(batch_invariant=bool(config.batch_invariant))
        def prepared(name, shapes, seed):
            sd = get(name, shapes, seed)
            out = prepare(sd, self.dtype, dev, heads=8, fold_ln=2 if (config.fold_layernorms and config.fp8 < 2) else False)
            if config.fp8 >= 1:
                from .sd.weights import add_fp8_convs, add_fp8_linears
                add_fp8_convs(out, sd, dev)
                if config.fp8 >= 2:
                    add_fp8_linears(out)
            return out
        self.pipe = DenoisePipeline(prepared("unet", arch.unet_shapes(), 100), prepared("controlnet", arch.controlnet_shapes(), 200),
                                    prepare_vae_weights(get("vae_decoder", arch.vae_decoder_shapes(), 300), self.dtype, dev),
                                    self.num_inference_steps, self.guidance_scale, self.controlnet_conditioning_scale)
        self.vae_encoder = VAEEncoder(prepare_vae_encoder_weights(get("vae_encoder", arch.vae_encoder_shapes(), 400), self.dtype, dev))
        if text_encoder is None:
            if not config.synthetic_weights:
                raise ValueError("a text_encoder (prompt -> [1,77,768]) is required; the hashed stand-in is used only with "
                                 "GaussCtrlPipelineConfig.synthetic_weights=True")
            text_encoder = _hash_text_encoder
            self.weights_source["text_encoder"] = "synthetic(hash of the prompt)"
        self.text_encoder = text_encoder                           # CLIP text tower: outside the hot path
        self.mask_fn = mask_fn                                     # LangSAM stand-in: image[H,W,3] -> mask[H,W] (out of scope)
        self.bank_hook = None                                      # optional callable(RefBank | None), called once inside edit_images (tests)
        self._spread_calls = 0                                     # training steps taken in train_mode "throughput" / "sharded" (view schedule)
        print("[gaussctrl_amd] diffusion weights: " + ", ".join(f"{k} <- {v}" for k, v in sorted(self.weights_source.items())))

    if not HAVE_NERFSTUDIO:
        @property
        def device(self):
            """nerfstudio's Pipeline exposes this as a read-only property (the model's device); same here without it"""
            return self._dev

    @property
    def model(self):
        return self._model.module if hasattr(self._model, "module") and not isinstance(self._model, GaussCtrlModel) else self._model

    def get_training_callbacks(self, training_callback_attributes):
        """gc_trainer.py:112-118 asks the pipeline; VanillaPipeline = datamanager callbacks + model callbacks."""
        if HAVE_NERFSTUDIO and hasattr(_PipelineBase, "get_training_callbacks"):
            return super().get_training_callbacks(training_callback_attributes)
        return list(self.model.get_training_callbacks(training_callback_attributes))

    def _encode(self, prompt):
        return self.text_encoder(prompt).to(self._dev)

    def _my_views(self):
        from .dist import shard_views
        return shard_views(len(self.datamanager.cameras), self.world_size, self.local_rank)

    # ------------------------------------------------------------------------------------ :122-157
    @torch.no_grad()
    def render_reverse(self, views=None):
        """Render rgb + depth of every (local) view and DDIM-invert the renders to z_0 (batched)."""
        views = self._my_views() if views is None else list(views)
        td = self.datamanager.train_data
        rb = max(1, int(self.config.render_batch))
        batched = rb > 1 and hasattr(self.model, "get_outputs_for_cameras")
        def keep(cam_idx, out):
            if "depth" not in out:          # gc_model.py:155-156: a view that meets no Gaussian returns only {"rgb": background}
                raise RuntimeError(f"render_reverse: view {cam_idx} intersects no Gaussian (no depth to condition the ControlNet on); "
                                   "the reference fails here too (gc_pipeline.py:131 reads rendered_image['depth'])")
            rgb, depth = out["rgb"], out["depth"][..., 0]
            if self.config.round_like_reference:                              # :132-133 `.to(torch.float16)` (values kept in fp32 storage)
                rgb, depth = rgb.to(torch.float16).float(), depth.to(torch.float16).float()
            td[cam_idx]["unedited_image"] = rgb.clone() if batched else rgb   # [H,W,3] fp32, stays on the GPU (clone: drop the batch's other planes)
            td[cam_idx]["depth_image"] = depth.clone() if batched else depth  # [H,W]
            if self.config.langsam_obj != "" and self.mask_fn is not None:
                td[cam_idx]["mask_image"] = self.mask_fn(rgb, self.config.langsam_obj)
        if batched:                       # each group is consumed right after its launch set: only one batch of outputs is alive at a time
            for s0 in range(0, len(views), rb):
                grp = views[s0:s0 + rb]
                for cam_idx, out in zip(grp, self.model.get_outputs_for_cameras([self.datamanager.cameras[i] for i in grp])):
                    keep(cam_idx, out)
        else:
            for cam_idx in views:         # (views may repeat: ref_indices)
                keep(cam_idx, self._model.get_outputs_for_camera(self.datamanager.cameras[cam_idx]))
        ctx = self._encode(self.positive_reverse_prompt)
        for s in range(0, len(views), max(self.chunk_size, 1)):
            chunk = views[s:s + max(self.chunk_size, 1)]
            lat0 = torch.cat([self.image2latent(td[i]["unedited_image"]) for i in chunk], 0)
            disp = torch.stack([self.depth2disparity_torch(td[i]["depth_image"]) for i in chunk])
            z0 = self.pipe.invert(lat0, disp, ctx)                            # guidance_scale=0 -> no CFG batch (:142-145)
            for j, i in enumerate(chunk):
                td[i]["z_0_image"] = z0[j:j + 1]

    # ------------------------------------------------------------------------------------ :159-237
    @torch.no_grad()
    def edit_images(self):
        """Chunked cross-view ControlNet denoise of the (local) views; writes train_data[i]['image'] (HWC fp32)."""
        td = self.datamanager.train_data
        cn, cp = self._encode(self.negative_prompts), self._encode(self.positive_prompt)
        owner = self.config.ref_bank_owner if (self.world_size > 1 and self.config.cache_reference_kv) else -1
        if self.config.ref_bank_allgather and self.config.cache_reference_kv and self.world_size > 1 and self.world_size not in (2, 4, 8):
            raise ValueError(f"ref_bank_allgather shards the 2 x 4 reference samples over 2, 4 or 8 ranks, not {self.world_size}: "
                             "use ref_bank_owner (broadcast) or the replicated default")
        gather = bool(self.config.ref_bank_allgather) and self.config.cache_reference_kv and self.world_size in (2, 4, 8)
        if gather:
            owner = -1
        need_refs = owner < 0 or self.local_rank == owner or not self.config.cache_reference_kv
        if need_refs:
            self.render_reverse_refs()        # views are sharded: the 4 reference views may live on other ranks (cheap to redo here)
        ref_z0 = ref_disp = None
        if need_refs:
            ref_z0 = torch.cat([td[i]["z_0_image"] for i in self.ref_indices], 0)
            ref_disp = torch.stack([self.depth2disparity_torch(td[i]["depth_image"]) for i in self.ref_indices])
        bank = None
        if self.config.cache_reference_kv:
            if gather:
                # the reference trajectory sharded by sample, K / V^T all-gathered per attention layer (north_star's collective)
                from .dist import RefShard
                tr = self.pipe.begin_ref_bank_sharded(ref_z0, ref_disp, cn, cp, RefShard(self.world_size, self.local_rank))
                bank = self.pipe.advance_ref_bank(tr, None)
            elif owner >= 0:
                # one owner rank runs the 4-view reference trajectory; each DDIM step's K / V^T is broadcast (one flat RCCL message
                # per step) while the owner already computes the next step (SURVEY.md 8e collective 1)
                from .dist import broadcast_ref_bank_pipelined
                bank = broadcast_ref_bank_pipelined(self.pipe, ref_z0, ref_disp, cn, cp, owner, self.world_size, self.local_rank,
                                                    self._dev, self.num_inference_steps)
            else:
                bank = self.pipe.build_ref_bank(ref_z0, ref_disp, cn, cp)
        if self.bank_hook is not None:            # inspection only (tests hash the bank here); the bank itself is NOT retained: 10-15 GB of
            self.bank_hook(bank)                  # K / V^T that would otherwise stay pinned through the whole training phase
        views = self._my_views()
        # consecutive chunks only share the (read-only) bank: run them on alternating streams so that one chunk's part-filled grids and
        # fill / drain phases are covered by the other's kernels (bench.py --inflight: +3 % views/s at chunk_size 3)
        main = torch.cuda.current_stream()
        n_fly = max(1, int(self.config.inflight_chunks)) if (bank is not None and self._dev.type == "cuda") else 1
        if n_fly > 1 and len(getattr(self, "_chunk_streams", None) or ()) != n_fly:      # (inflight_chunks may change between calls)
            self._chunk_streams = [torch.cuda.Stream(device=self._dev) for _ in range(n_fly)]
        ready = torch.cuda.Event() if n_fly > 1 else None
        if ready is not None:
            # the chunk streams are ordered after `main` only: everything they share must be complete there first.  With a bank
            # received by broadcast nothing has touched the pipeline's lazily filled caches yet (text K / V^T, time-embedding rows).
            self.pipe.warm_caches(cn, cp)
            ready.record(main)
        per_launch = self.chunk_size * (max(1, int(self.config.chunks_per_launch)) if bank is not None else 1)
        if per_launch > self.chunk_size:
            # one launch set = 2 CFG halves x per_launch views in one network batch; validated up to 21 views (42 frames: 7 chunks of 3,
            # profiles/r06_cobatch_sweep.txt) -- whole chunks only, never fewer than one
            per_launch = max(self.chunk_size, min(per_launch, (21 // self.chunk_size) * self.chunk_size))
        for ci, s in enumerate(range(0, len(views), per_launch)):
            chunk = views[s:s + per_launch]
            stream = self._chunk_streams[ci % n_fly] if n_fly > 1 else main
            if n_fly > 1:
                stream.wait_event(ready)          # bank, z_0 and depth images were produced on the caller's stream
            with torch.cuda.stream(stream):
                lat = torch.cat([td[i]["z_0_image"] for i in chunk], 0)
                disp = torch.stack([self.depth2disparity_torch(td[i]["depth_image"]) for i in chunk])
                if bank is not None:
                    out = self.pipe.edit_chunk_cached(lat, disp, cn, cp, bank)
                else:                                                             # reference order: refs first (:206-207), drop them (:219)
                    out = self.pipe.edit_chunk(torch.cat([ref_z0, lat]), torch.cat([ref_disp, disp]), cn, cp)[self.num_ref_views:]
                z = to_nhwc8(out / 0.18215, self.dtype)
                imgs = self.pipe.vae.decode(z, postprocess=True)                  # [c,H,W,8] fp32, channels 0..2 in [0,1]
                for j, i in enumerate(chunk):
                    mask = td[i].get("mask_image")
                    td[i]["image"] = sdops.mask_composite(imgs[j], td[i]["unedited_image"] if mask is not None else None,
                                                          None if mask is None else mask.float())        # :226-234
                    if n_fly > 1:
                        td[i]["image"].record_stream(main)        # consumed on the caller's stream from now on
        if n_fly > 1:
            for st in self._chunk_streams:
                main.wait_stream(st)
        if self.world_size > 1:
            self._allgather_images()

    def render_reverse_refs(self):
        """world_size > 1: every rank needs the 4 reference views' z_0 / depth (cheap to recompute locally)."""
        td = self.datamanager.train_data
        self.render_reverse([i for i in self.ref_indices if "z_0_image" not in td[i]])

    def _allgather_images(self):
        from .dist import allgather_view_images
        td = self.datamanager.train_data
        mine = self._my_views()
        cam = self.datamanager.cameras[0]          # a rank may own no view (V < world_size): the image shape comes from the camera
        shape = (int(cam.height.reshape(-1)[0]), int(cam.width.reshape(-1)[0]), 3)
        allv = allgather_view_images({i: td[i]["image"] for i in mine}, len(td), self.world_size, self.local_rank, shape, self._dev)
        for i, img in allv.items():
            td[i]["image"] = img

    # ------------------------------------------------------------------------------------ :239-274
    @torch.no_grad()
    def image2latent(self, image):
        """image [H,W,3] in [0,1] -> latents [1,4,H/8,W/8] = vae.encode(2x-1).mean * 0.18215"""
        x = to_nhwc8((image * 2 - 1).permute(2, 0, 1)[None], self.dtype)
        mean = self.vae_encoder.encode_mean(x)[..., :4]
        return (mean * 0.18215).permute(0, 3, 1, 2).contiguous()

    def depth2disparity_torch(self, depth):
        """depth [H,W] (or [1,H,W]) -> disparity [3,H,W] fp32 = 1/(d+1e-5)/max (:258-266)"""
        d = depth.reshape(depth.shape[-2], depth.shape[-1]).float()
        if self.config.round_like_reference:
            # the reference evaluates :263-264 on the fp16 depth, i.e. with one fp16 rounding after each of the three operations
            # (glue, three tiny elementwise launches; only under this flag -- the product path is the fused kernel below)
            h = 1 / (d.to(torch.float16) + 1e-5)
            h = h / torch.max(h)
            return h[None].expand(3, -1, -1).float()
        disp = sdops.depth_to_disparity(d, self.dtype)[..., :3]
        return disp.permute(2, 0, 1).float()

    depth2disparity = depth2disparity_torch

    def update_datasets(self, cam_idx, unedited_image, depth, latent, mask):
        td = self.datamanager.train_data[cam_idx]
        td["unedited_image"], td["depth_image"], td["z_0_image"] = unedited_image, depth, latent
        if mask is not None:
            td["mask_image"] = mask

    # ------------------------------------------------------------------------------------ mid-result cache
    def save_mid_results(self, root, views=None):
        """Write depth_npy / z_0 / mask_npy / unedited in the reference's on-disk layout (gc_dataparser_ns.py:408-420)."""
        from . import midcache
        td = self.datamanager.train_data
        for i in (self._my_views() if views is None else views):
            if "z_0_image" in td[i]:
                midcache.save_view(root, i, unedited_image=td[i].get("unedited_image"), depth=td[i].get("depth_image"),
                                   z_0=td[i]["z_0_image"], mask=td[i].get("mask_image"))

    def load_mid_results(self, root, views=None) -> list:
        """Fill train_data from a scene folder prepared earlier (by this code or by the reference); returns the views found."""
        from . import midcache
        td = self.datamanager.train_data
        found = []
        for i in (self._my_views() if views is None else views):
            if midcache.has_view(root, i):
                d = midcache.load_view(root, i, self._dev)
                d["depth_image"] = d["depth_image"][0]                                   # [H,W] as render_reverse stores it
                td[i].update(d)
                found.append(i)
        return found

    # ------------------------------------------------------------------------------------ :276-291
    def get_train_loss_dict(self, step: int):
        camera, batch = self.datamanager.next_train(step)
        model_outputs = self._model(camera)
        metrics_dict = self._model.get_metrics_dict(model_outputs, batch)
        loss_dict = self._model.get_loss_dict(model_outputs, batch, metrics_dict)
        return model_outputs, loss_dict, metrics_dict          # (world_size > 1: the gradient reduction happens after backward, train_iteration)

    def reduce_gradients(self):
        """world_size > 1, stand-alone callers: one flat all-reduce (average) of whatever .grad tensors the parameters hold (gather copy +
        blocking collective + copy back).  train_iteration does NOT use this: it reduces the FlatGrads buffer the backward wrote, in place."""
        from .dist import allreduce_gradients
        allreduce_gradients(list(self._model.parameters()), self.world_size)

    def get_param_groups(self):
        return self._model.get_param_groups()

    _GRAD_KEYS = ("means", "scales", "quats", "opacities", "features_dc", "features_rest")

    def _flat_grads(self):
        """the six leaf gradients as views of ONE flat fp32 buffer (dist.FlatGrads) that the fused backward writes and RCCL reduces in place"""
        from .dist import FlatGrads
        m = self.model
        fg = getattr(self, "_fg", None)
        if fg is None or fg.views["means"].shape != m.means.shape or fg.flat.device != m.means.device or fg.flat.numel() != fg.n:
            fg = self._fg = FlatGrads({k: getattr(m, k) for k in self._GRAD_KEYS})
        return fg

    _GROUP_OF = {"xyz": "means", "scaling": "scales", "rotation": "quats", "opacity": "opacities", "features_dc": "features_dc", "features_rest": "features_rest"}

    def _sharded_adam(self):
        """dist.ShardedAdam over the flat parameter / gradient buffers of the six leaf tensors (rebuilt when densification changes their size)"""
        from .dist import FlatGrads, FlatParams, ShardedAdam
        m = self.model
        params = {k: getattr(m, k) for k in self._GRAD_KEYS}
        sa = getattr(self, "_sa", None)
        if sa is None or not sa.fp.matches(params):
            old = sa
            fp = FlatParams(params, self.world_size)
            self._fg = FlatGrads(params, pad_to=fp.flat.numel())
            sa = self._sa = ShardedAdam(fp, self._fg, self.world_size, self.local_rank)
            if old is not None:
                # the parameter set changed under us: a cull (gc_trainer.CullCallback / SplatfactoModel.cull_gaussians) pruned every leaf tensor
                # with one row mask.  The replicated path prunes Adam's moments with the same mask; so do we (ShardedAdam.adopt) -- otherwise
                # the bias correction would restart and every surviving Gaussian take an lr * sign(g) step.
                keep = getattr(m, "_cull_keep", None)
                n_old = old.fp.spans["means"][1] // 3 - old.fp.spans["means"][0] // 3
                if keep is not None and keep.numel() == n_old and int(keep.sum()) == m.means.shape[0]:
                    sa.adopt(old, keep)
                else:
                    import warnings
                    warnings.warn("train_mode 'sharded': the Gaussian set changed without a cull mask (densification?); Adam moments restart at zero")
            m._cull_keep = None
        return sa

    def sharded_adam_state(self) -> Optional[dict]:
        """train_mode "sharded": the optimizer state for a checkpoint (moments of all parameters in flat order + step count; a collective when
        world_size > 1 -- call it on every rank, save on one).  None before the first sharded step."""
        sa = getattr(self, "_sa", None)
        return None if sa is None else sa.state_dict(full=True)

    def load_sharded_adam_state(self, sd: dict) -> None:
        self._sharded_adam().load_state_dict(sd)

    def _sync_view(self, draw):
        """train_mode "parity": every rank trains on the view rank 0 drew, against the background rank 0 drew (background_color "random" is
        torch.rand(3) of each rank's OWN generator: replicas would diverge from the first step) -- one small control-path broadcast, no
        data-path collective.  The replicas' Gaussian counts travel along and must agree (a diverged replica culls differently)."""
        import torch.distributed as dist
        m = self.model
        box = [(draw(), [random.random() for _ in range(3)], int(m.num_points)) if self.local_rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        view, bg, n0 = box[0]
        if int(m.num_points) != n0:
            raise RuntimeError(f"train_mode 'parity': rank {self.local_rank} holds {int(m.num_points)} Gaussians, rank 0 {n0} -- the replicas diverged")
        m.background_override = torch.tensor(bg, dtype=torch.float32, device=m.device)
        return int(view)

    def _spread_view(self, draw):
        """train_modes "throughput" / "sharded": the N ranks of a step render N DISTINCT views.  Every rank derives the same per-epoch
        permutation of the views from a private generator (never the global `random`, which every rank seeds identically in __init__ and in
        the datamanager) and rank r takes element step * N + r of it: no collective, no duplicated gradient."""
        n = len(self.datamanager.train_data)
        k = self._spread_calls * self.world_size + self.local_rank
        self._spread_calls += 1
        epoch, pos = divmod(k, n)
        return random.Random(0x5EED ^ (epoch * 7919)).sample(range(n), n)[pos]

    def train_iteration(self, optimizers: dict, step: int):
        """One splat-optimisation iteration as GaussCtrlTrainer.train_iteration runs it (gc_trainer.py:257-301): zero grads,
        forward + loss (gc_pipeline.py:276-287), backward, Adam steps (gc_config.py:58-87).  Returns (loss, loss_dict, metrics_dict).
        world_size > 1:
          train_mode "parity"     -- the reference's schedule: ONE view per step (gc_trainer.py:186-201).  Every rank renders the view rank 0
                                     drew, on replicated parameters: identical gradients, identical Adam steps, no gradient collective;
          train_mode "throughput" -- each rank renders the view ITS datamanager draws (an N-view batch per step); the fused backward writes
                                     the six leaf gradients into one flat buffer (dist.FlatGrads / RenderAux.grad_into: no autograd .grad
                                     tensors, no gather copy), the loss carries the 1 / N so that ONE in-place RCCL all-reduce (sum) of that
                                     buffer yields the batch mean, and the optimizers read the buffer's views -- the path bench.py measures."""
        for opt in optimizers.values():
            opt.zero_grad(set_to_none=True)
        loss, loss_dict, metrics_dict = self.train_forward_backward(step)
        if self.world_size > 1 and self.config.train_mode == "sharded":
            # train_mode "sharded" -- "throughput" with SURVEY.md 8e's second form of collective 2: reduce-scatter of the flat gradient buffer, every
            # rank runs the fused Adam on its 1 / N slice of the flat parameter buffer (moment state 1 / N per rank), all-gather of the slices in
            # place (dist.ShardedAdam); the optimizers passed in only supply this step's lr / eps per group (their schedulers keep working)
            hyper = {key: (optimizers[g].param_groups[0]["lr"], optimizers[g].param_groups[0]["eps"]) for g, key in self._GROUP_OF.items() if g in optimizers}
            self._sharded_adam().step(hyper)
            for g, opt in optimizers.items():            # groups outside the six leaf tensors (e.g. camera_opt) keep their own optimizer
                if g not in self._GROUP_OF:
                    opt.step()
            return loss, loss_dict, metrics_dict
        for opt in optimizers.values():
            opt.step()
        return loss, loss_dict, metrics_dict

    def train_forward_backward(self, step: int, accumulating: bool = False):
        """forward + loss + backward of one training step, gradients left where the optimizers read them (world_size > 1: per train_mode,
        see train_iteration).  accumulating: the caller accumulates gradients over several steps (gc_trainer gradient_accumulation_steps
        > 1): the autograd .grad path with a flat all-reduce of the accumulated tensors is used instead of the write-once flat buffer."""
        multi = self.world_size > 1
        mode = self.config.train_mode if multi else "single"
        if mode not in ("single", "parity", "throughput", "sharded"):
            raise ValueError(f"train_mode must be 'parity', 'throughput' or 'sharded', not {mode!r}")
        if mode == "sharded" and accumulating:
            raise ValueError("train_mode 'sharded' writes the flat gradient buffer once per step: no gradient accumulation")
        m = self.model
        fg = None
        if mode == "sharded":
            fg = self._sharded_adam().fg
            m.grad_into = fg.views
        elif mode == "throughput" and not accumulating:
            fg = self._flat_grads()
            m.grad_into = fg.views
        self.datamanager.view_sync = self._sync_view if mode == "parity" else self._spread_view if mode in ("throughput", "sharded") else None
        try:
            _, loss_dict, metrics_dict = self.get_train_loss_dict(step)
        finally:
            m.grad_into = None
            m.background_override = None
        loss = sum(loss_dict.values())
        if fg is not None:
            aux = getattr(m, "_aux", None)
            honoured = aux is not None and aux.grad_into is not None       # the model only takes grad_into without a crop box (gc_model.get_outputs)
            (loss / self.world_size).backward()
            if not honoured:
                # autograd owns this step's gradients (crop box during training, or nothing rendered): move them into the flat buffer so that the
                # collective below sees this step's values, never stale buffer contents
                fg.flat.zero_()
                for k in self._GRAD_KEYS:
                    g = getattr(m, k).grad
                    if g is not None:
                        fg.views[k].copy_(g)
            elif aux.xys_grad is None:
                fg.flat.zero_()                          # this rank's view rendered nothing (gc_model.py:155-156): it contributes zeros
            if mode == "sharded":
                return loss.detach(), loss_dict, metrics_dict          # ShardedAdam.step reduce-scatters the buffer (train_iteration)
            fg.reduce_async(self.world_size)
            fg.wait()
            for k in self._GRAD_KEYS:
                getattr(m, k).grad = fg.views[k]
        else:
            loss.backward()
            if mode == "throughput":
                self.reduce_gradients()
        return loss.detach(), loss_dict, metrics_dict

    def forward(self):
        """Not implemented since we only want the parameter saving of the nn module, but not forward()"""
        raise NotImplementedError


def _hash_text_encoder(prompt: str) -> torch.Tensor:
    """Deterministic synthetic [1,77,768] embedding of a prompt (stand-in when no CLIP weights are available)."""
    import hashlib
    seed = int.from_bytes(hashlib.sha256(prompt.encode()).digest()[:4], "little")
    g = torch.Generator().manual_seed(seed)
    return torch.randn(1, 77, 768, generator=g)
