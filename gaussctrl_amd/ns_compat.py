"""nerfstudio surface the plugin needs.  When nerfstudio (pinned 1.0.0 by /root/reference/README.md:53-56) is
importable the plugin classes SUBCLASS its own bases (Trainer / VanillaPipeline / SplatfactoModel /
FullImageDatamanager and their configs) so that `ns-train gaussctrl` instantiates them through the usual config tree;
otherwise minimal stand-ins with the same attribute names keep the Pipeline / Model / Trainer classes usable (the build
environment has no nerfstudio and no network).  tests/test_plugin_config.py runs the nerfstudio branches against the
stand-in package tests/fake_nerfstudio."""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

try:  # pragma: no cover - exercised where nerfstudio (or the test double) is importable
    from nerfstudio.cameras.cameras import Cameras  # type: ignore
    HAVE_NERFSTUDIO = True
except Exception:  # noqa: BLE001
    HAVE_NERFSTUDIO = False

    class Cameras:  # type: ignore
        """Subset of nerfstudio.cameras.cameras.Cameras used by the hot path: a batch of pinhole cameras."""

        def __init__(self, camera_to_worlds, fx, fy, cx, cy, width, height, metadata=None):
            c2w = torch.as_tensor(camera_to_worlds, dtype=torch.float32)
            if c2w.dim() == 2:
                c2w = c2w[None]
            n = c2w.shape[0]

            def col(v, dt):
                v = torch.as_tensor(v, dtype=dt).reshape(-1, 1)
                return v.expand(n, 1).clone() if v.shape[0] == 1 else v
            self.camera_to_worlds = c2w[:, :3, :4].contiguous()
            self.fx, self.fy, self.cx, self.cy = (col(v, torch.float32) for v in (fx, fy, cx, cy))
            self.width, self.height = col(width, torch.int64), col(height, torch.int64)
            self.metadata = metadata

        @property
        def shape(self):
            return self.camera_to_worlds.shape[:1]

        def __len__(self):
            return self.camera_to_worlds.shape[0]

        def __getitem__(self, i):
            if isinstance(i, int):
                i = slice(i, i + 1)
            return Cameras(self.camera_to_worlds[i], self.fx[i], self.fy[i], self.cx[i], self.cy[i], self.width[i],
                           self.height[i], self.metadata)

        def to(self, device):
            return self     # camera glue is host math in this implementation (gaussctrl_amd/camera.py)

        def rescale_output_resolution(self, s):
            if s != 1:
                raise NotImplementedError("camera rescaling (splatfacto resolution schedule) ended long before step 30000")


@dataclass
class OptimizerSpec:
    """Adam groups of /root/reference/gaussctrl/gc_config.py:58-87 (name -> lr, eps, schedule)."""
    lr: float
    eps: float = 1e-15
    lr_final: float | None = None
    max_steps: int | None = None


PARAM_GROUPS = {
    "xyz": OptimizerSpec(1.6e-4, 1e-15, 1.6e-6, 30000),
    "features_dc": OptimizerSpec(0.0025),
    "features_rest": OptimizerSpec(0.0025 / 20),
    "opacity": OptimizerSpec(0.05),
    "scaling": OptimizerSpec(0.005),
    "rotation": OptimizerSpec(0.001),
    "camera_opt": OptimizerSpec(1e-3, 1e-15, 5e-5, 30000),
}


def exp_decay_lr(step: int, lr_init: float, lr_final: float, max_steps: int) -> float:
    """nerfstudio ExponentialDecayScheduler (warmup_steps = 0, the reference's settings at gc_config.py:59-66,83-86) [recall
    nerfstudio 1.0.0 engine/schedulers.py]: log-linear interpolation lr_init -> lr_final over max_steps, constant afterwards.
    A GaussCtrl run starts from a step-30000 splatfacto checkpoint (gc_trainer.py:75), i.e. at t = 1."""
    t = min(max(step / max_steps, 0.0), 1.0)
    return math.exp(math.log(lr_init) * (1.0 - t) + math.log(lr_final) * t)
