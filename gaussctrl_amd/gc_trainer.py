"""GaussCtrlTrainer: the caller of the hot path (/root/reference/gaussctrl/gc_trainer.py).

The reference's trainer is nerfstudio's Trainer with two changes: `setup()` runs the whole edit (render_reverse, then
edit_images when test_mode == "val") between checkpoint loading and the viewer set-up (gc_trainer.py:58-78), and `train()`
runs `pipeline.config.render_rate` iterations starting at the loaded checkpoint's step (gc_trainer.py:186-187).  The trainer
itself is control plane (out of scope as code); what is kept here is its CALL CONTRACT on the pipeline:

    pipeline = config.pipeline.setup(device, test_mode, world_size, local_rank, grad_scaler)     gc_trainer.py:67-73
    optimizers (7 Adam groups + exp-decay schedulers, gc_config.py:58-87);  _load_checkpoint()   :74-75
    pipeline.render_reverse();  pipeline.edit_images() if test_mode == "val"                      :76-78
    callbacks = pipeline.get_training_callbacks(TrainingCallbackAttributes(...))                   :112-118
    render_rate x train_iteration(step): zero_grad_some / get_train_loss_dict / backward / optimizer_step_some /
    scheduler_step_all, callbacks before and after                                                 :186-207,257-301

Under nerfstudio `GaussCtrlTrainer` subclasses `nerfstudio.engine.trainer.Trainer` (so `ns-train gaussctrl` drives it);
without nerfstudio the same sequence runs on a small built-in loop (tests, bench, multi-GPU checks)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch

from .ns_compat import HAVE_NERFSTUDIO, PARAM_GROUPS, exp_decay_lr

if HAVE_NERFSTUDIO:  # pragma: no cover - executed with nerfstudio (or tests/fake_nerfstudio) on the path
    from nerfstudio.engine.trainer import Trainer as _TrainerBase, TrainerConfig as _TrainerConfigBase  # type: ignore

    @dataclass
    class GaussCtrlTrainerConfig(_TrainerConfigBase):
        """gc_trainer.py:42-47"""
        _target: type = field(default_factory=lambda: GaussCtrlTrainer)
        steps_per_save: int = 500

    class GaussCtrlTrainer(_TrainerBase):
        """nerfstudio Trainer + the edit phase in setup() and the render_rate-iteration train()."""

        def setup(self, test_mode="val") -> None:
            super().setup(test_mode=test_mode)      # pipeline, optimizers, checkpoint, viewer, callbacks, writers
            self.pipeline.render_reverse()          # gc_trainer.py:76
            if self.pipeline.test_mode == "val":    # :77-78
                self.pipeline.edit_images()

        def train(self) -> None:
            # gc_trainer.py:186-187: `render_rate` iterations from the loaded checkpoint's step (max_num_iterations is ignored there).
            # nerfstudio 1.0.0 (the reference's pin) loops `range(start, start + max_num_iterations)`; later releases loop to an ABSOLUTE
            # max_num_iterations, which would give zero iterations from a step-30000 checkpoint: refuse rather than train nothing.
            import nerfstudio  # type: ignore
            ver = getattr(nerfstudio, "__version__", None)
            if ver is not None and not str(ver).startswith("1.0."):
                raise RuntimeError(f"gaussctrl_amd's trainer follows nerfstudio 1.0.x's training loop (found {ver}); see INTEGRATION.md 3")
            keep = self.config.max_num_iterations
            self.config.max_num_iterations = self.pipeline.config.render_rate
            try:
                super().train()
            finally:
                self.config.max_num_iterations = keep
else:
    @dataclass
    class GaussCtrlTrainerConfig:
        """Field names / defaults of nerfstudio's TrainerConfig that the reference sets (gc_config.py:41-50,88-89) plus the
        class default of gc_trainer.py:42-47; values are filled in by gaussctrl_amd.gc_config."""
        _target: type = field(default_factory=lambda: GaussCtrlTrainer)
        method_name: str = "gaussctrl"
        steps_per_save: int = 500
        steps_per_eval_image: int = 500
        steps_per_eval_batch: int = 500
        steps_per_eval_all_images: int = 25000
        max_num_iterations: int = 1000000
        save_only_latest_checkpoint: bool = True
        mixed_precision: bool = False
        gradient_accumulation_steps: Dict[str, int] = field(default_factory=dict)
        pipeline: object = None
        optimizers: Dict[str, dict] = field(default_factory=dict)
        viewer: Dict[str, int] = field(default_factory=dict)
        vis: str = "viewer"
        load_step: int = 30000            # the splatfacto checkpoint a GaussCtrl run starts from (gc_trainer.py:75, scripts/*.sh)

        def setup(self, local_rank: int = 0, world_size: int = 1, **kw):
            return self._target(self, local_rank=local_rank, world_size=world_size, **kw)

    class TrainingCallbackAttributes:
        def __init__(self, optimizers, grad_scaler, pipeline):
            self.optimizers, self.grad_scaler, self.pipeline = optimizers, grad_scaler, pipeline

    class GaussCtrlTrainer:
        """Built-in stand-in for nerfstudio's Trainer running the reference's sequence (module docstring)."""

        def __init__(self, config: GaussCtrlTrainerConfig, local_rank: int = 0, world_size: int = 1, device: Optional[str] = None,
                     pipeline_kwargs: Optional[dict] = None):
            self.config = config
            self.local_rank, self.world_size = local_rank, world_size
            self.device = device or (f"cuda:{local_rank}" if torch.cuda.is_available() else "cpu")
            self.pipeline_kwargs = pipeline_kwargs or {}
            self._start_step = config.load_step
            self.gradient_accumulation_steps = dict(config.gradient_accumulation_steps)
            self.grad_scaler = None
            self.pipeline = None
            self.optimizers: Dict[str, torch.optim.Optimizer] = {}
            self.callbacks: List = []

        # -- gc_trainer.py:58-134 (viewer / writers / profiler left out: control plane)
        def setup(self, test_mode="val") -> None:
            self.pipeline = self.config.pipeline.setup(device=self.device, test_mode=test_mode, world_size=self.world_size,
                                                       local_rank=self.local_rank, grad_scaler=self.grad_scaler,
                                                       **self.pipeline_kwargs)
            self.optimizers = self.setup_optimizers()
            self.pipeline.render_reverse()
            if self.pipeline.test_mode == "val":
                self.pipeline.edit_images()
            self.callbacks = self.pipeline.get_training_callbacks(
                TrainingCallbackAttributes(optimizers=self.optimizers, grad_scaler=self.grad_scaler, pipeline=self.pipeline))

        def setup_optimizers(self) -> Dict[str, torch.optim.Optimizer]:
            from .gc_config import build_optimizers
            return build_optimizers(self.pipeline.model, self.config.optimizers or None)

        def lr_at(self, group: str, step: int) -> float:
            from .gc_config import scheduled_lr
            return scheduled_lr(group, step, self.config.optimizers or None)

        # -- gc_trainer.py:176-207 (no viewer lock / writers)
        def train(self) -> List[float]:
            losses = []
            for step in range(self._start_step, self._start_step + self.pipeline.config.render_rate):
                self.pipeline.train()
                for cb in self.callbacks:
                    cb.run_callback_at_location(step, "before_train_iteration")
                loss, _, _ = self.train_iteration(step)
                for cb in self.callbacks:
                    cb.run_callback_at_location(step, "after_train_iteration")
                losses.append(loss)
            return losses

        # -- gc_trainer.py:257-301
        def train_iteration(self, step: int):
            acc = lambda g: self.gradient_accumulation_steps.get(g, 1)
            for g, opt in self.optimizers.items():                       # zero_grad_some
                if step % acc(g) == 0:
                    opt.zero_grad(set_to_none=True)
            # forward + loss + backward; world_size > 1: GaussCtrlPipelineConfig.train_mode -- "parity" (every rank takes rank 0's view: the
            # reference's single-view schedule on replicas, no gradient collective) or "throughput" (one view per rank, the fused backward
            # writes one flat buffer that RCCL all-reduces in place: dist.FlatGrads)
            if hasattr(self.pipeline, "train_forward_backward"):
                loss, loss_dict, metrics_dict = self.pipeline.train_forward_backward(step, accumulating=any(acc(g) > 1 for g in self.optimizers))
            else:                                                        # a plain VanillaPipeline-shaped object (gc_trainer.py:272-275)
                _, loss_dict, metrics_dict = self.pipeline.get_train_loss_dict(step=step)
                loss = sum(loss_dict.values())
                loss.backward()
                self.pipeline.reduce_gradients()
                loss = loss.detach()
            sharded = getattr(self.pipeline, "world_size", 1) > 1 and getattr(getattr(self.pipeline, "config", None), "train_mode", "") == "sharded"
            for g, opt in self.optimizers.items():                       # optimizer_scaler_step_some
                if step % acc(g) == acc(g) - 1:
                    for pg in opt.param_groups:
                        pg["lr"] = self.lr_at(g, step)                   # scheduler value for this step (scheduler_step_all, :294-298)
                    if not sharded or g not in getattr(self.pipeline, "_GROUP_OF", {}):      # (groups outside the six leaf tensors keep their optimizer)
                        opt.step()
            if sharded:        # reduce-scatter -> Adam on this rank's slice of the flat parameter buffer -> all-gather (dist.ShardedAdam), this step's lr / eps
                p = self.pipeline
                p._sharded_adam().step({key: (self.optimizers[g].param_groups[0]["lr"], self.optimizers[g].param_groups[0]["eps"])
                                        for g, key in p._GROUP_OF.items() if g in self.optimizers})
            return loss, loss_dict, metrics_dict


# ------------------------------------------------------------------------------------------------------------------------
# Training callbacks of the stand-alone model (under nerfstudio SplatfactoModel supplies its own).
class _Callback:
    where = ()

    def run_callback_at_location(self, step: int, location) -> None:
        name = getattr(location, "name", str(location)).lower()
        if name in self.where:
            self.run(step)


class StepCallback(_Callback):
    """SplatfactoModel.step_cb: the model tracks the global step (drives the SH degree schedule, gc_model.py:165)."""
    where = ("before_train_iteration",)

    def __init__(self, model):
        self.model = model

    def run(self, step):
        self.model.step = step


class CullCallback(_Callback):
    """What SplatfactoModel.refinement_after still does after stop_split_at (15000 < 30000): every `refine_every` steps cull
    Gaussians with sigmoid(opacity) < cull_alpha_thresh, and those with max(exp(scale)) > cull_scale_thresh once
    step > refine_every * reset_alpha_every; parameters and Adam state are pruned together [recall nerfstudio 1.0.0]."""
    where = ("after_train_iteration",)

    def __init__(self, model, optimizers: dict):
        self.model, self.optimizers = model, optimizers
        self.n_culled = 0

    @torch.no_grad()
    def run(self, step):
        c = self.model.config
        if not c.continue_cull_post_densification or step < c.stop_split_at or step % c.refine_every != 0:
            return
        m = self.model
        culls = (torch.sigmoid(m.opacities) < c.cull_alpha_thresh).squeeze(-1)
        if step > c.refine_every * c.reset_alpha_every:
            culls = culls | (torch.exp(m.scales).max(dim=-1).values > c.cull_scale_thresh)
        if not bool(culls.any()):
            return
        keep = ~culls
        m._cull_keep = keep              # train_mode "sharded" prunes its optimizer-state slices with the same mask (GaussCtrlPipeline._sharded_adam)
        self.n_culled += int(culls.sum())
        groups = m.get_param_groups()
        for gname, params in groups.items():
            opt = self.optimizers.get(gname)
            for p in params:
                st = opt.state.pop(p, None) if opt is not None else None
                p.data = p.data[keep].contiguous()
                p.grad = None
                if st:
                    for k in ("exp_avg", "exp_avg_sq"):
                        if k in st:
                            st[k] = st[k][keep].contiguous()
                    opt.state[p] = st
