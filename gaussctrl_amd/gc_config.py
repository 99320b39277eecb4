"""Method registration: drop-in for /root/reference/gaussctrl/gc_config.py:40-92.

`gaussctrl_method` is a nerfstudio MethodSpecification when nerfstudio is importable (entry point
`nerfstudio.method_configs: gaussctrl = gaussctrl_amd.gc_config:gaussctrl_method`, same method name as the
reference's pyproject.toml:38-39); otherwise a plain description with the same numbers so the optimiser setup
(7 Adam groups, gc_config.py:58-87) can be reproduced without nerfstudio."""
from __future__ import annotations

from .gc_model import GaussCtrlModelConfig
from .gc_pipeline import GaussCtrlDataManagerConfig, GaussCtrlPipelineConfig
from .ns_compat import HAVE_NERFSTUDIO, PARAM_GROUPS

TRAINER = dict(method_name="gaussctrl", steps_per_eval_image=100, steps_per_eval_batch=100, steps_per_save=250,
               steps_per_eval_all_images=100000, max_num_iterations=1000, mixed_precision=False,
               gradient_accumulation_steps={"camera_opt": 100})          # gc_config.py:42-50

if HAVE_NERFSTUDIO:  # pragma: no cover
    from nerfstudio.plugins.types import MethodSpecification  # type: ignore
    gaussctrl_method = MethodSpecification(config=dict(trainer=TRAINER, pipeline=GaussCtrlPipelineConfig(), optimizers=PARAM_GROUPS),
                                           description="GaussCtrl")
else:
    gaussctrl_method = dict(config=dict(trainer=TRAINER, pipeline=GaussCtrlPipelineConfig(), optimizers=PARAM_GROUPS),
                            description="GaussCtrl")


def build_optimizers(model):
    """Adam per parameter group with the reference's learning rates / eps (gc_config.py:58-87)."""
    from .train_ops import FusedAdam
    groups = model.get_param_groups()
    return {name: FusedAdam(params, lr=PARAM_GROUPS[name].lr, eps=PARAM_GROUPS[name].eps) for name, params in groups.items()}
