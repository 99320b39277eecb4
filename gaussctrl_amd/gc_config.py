"""Method registration: drop-in for /root/reference/gaussctrl/gc_config.py:40-92.

`gaussctrl_method` is a nerfstudio `MethodSpecification` whose `.config` is a real trainer-config object -- a
`GaussCtrlTrainerConfig(TrainerConfig)` holding the pipeline / datamanager / model config tree, the seven Adam groups with their
`AdamOptimizerConfig` / `ExponentialDecaySchedulerConfig`, the viewer config -- when nerfstudio is importable (entry point
`nerfstudio.method_configs: gaussctrl = gaussctrl_amd.gc_config:gaussctrl_method`, the reference's pyproject.toml:38-39).
Without nerfstudio it is the same tree built from the stand-in dataclasses, so the reference's numbers stay testable:
tests/test_plugin_config.py diff-checks every value below against tests/golden/ref_config.json, which
tests/golden/make_config_golden.py extracts from the reference sources."""
from __future__ import annotations

from dataclasses import dataclass

from .gc_model import GaussCtrlModelConfig
from .gc_pipeline import GaussCtrlDataManagerConfig, GaussCtrlPipelineConfig
from .gc_trainer import GaussCtrlTrainerConfig
from .ns_compat import HAVE_NERFSTUDIO, PARAM_GROUPS, exp_decay_lr

# gc_config.py:42-50
TRAINER = dict(method_name="gaussctrl", steps_per_eval_image=100, steps_per_eval_batch=0, steps_per_save=250,
               max_num_iterations=1000, steps_per_eval_all_images=1000, save_only_latest_checkpoint=True,
               mixed_precision=False, gradient_accumulation_steps={"camera_opt": 100})
VIEWER = dict(num_rays_per_chunk=1 << 15)          # gc_config.py:88
VIS = "viewer"                                     # gc_config.py:89


def optimizer_table() -> dict:
    """{group: {"optimizer": {lr, eps}, "scheduler": {lr_final, max_steps} | None}} -- gc_config.py:58-87"""
    return {name: {"optimizer": {"lr": s.lr, "eps": s.eps},
                   "scheduler": None if s.lr_final is None else {"lr_final": s.lr_final, "max_steps": s.max_steps}}
            for name, s in PARAM_GROUPS.items()}


def _pipeline_config():
    if HAVE_NERFSTUDIO:  # pragma: no cover
        from .gc_datamanager import GaussCtrlDataManager
        from .gc_dataparser import GaussCtrlDataParserConfig
        return GaussCtrlPipelineConfig(
            datamanager=GaussCtrlDataManagerConfig(_target=GaussCtrlDataManager, dataparser=GaussCtrlDataParserConfig(load_3D_points=True)),
            model=GaussCtrlModelConfig())
    return GaussCtrlPipelineConfig(datamanager=GaussCtrlDataManagerConfig(), model=GaussCtrlModelConfig())


if HAVE_NERFSTUDIO:  # pragma: no cover - executed with nerfstudio (or tests/fake_nerfstudio) on the path
    from nerfstudio.configs.base_config import ViewerConfig  # type: ignore
    from nerfstudio.engine.optimizers import AdamOptimizerConfig  # type: ignore
    from nerfstudio.engine.schedulers import ExponentialDecaySchedulerConfig  # type: ignore
    from nerfstudio.plugins.types import MethodSpecification  # type: ignore

    def _optimizers():
        return {name: {"optimizer": AdamOptimizerConfig(lr=s.lr, eps=s.eps),
                       "scheduler": None if s.lr_final is None else ExponentialDecaySchedulerConfig(lr_final=s.lr_final, max_steps=s.max_steps)}
                for name, s in PARAM_GROUPS.items()}

    gaussctrl_method = MethodSpecification(
        config=GaussCtrlTrainerConfig(**TRAINER, pipeline=_pipeline_config(), optimizers=_optimizers(),
                                      viewer=ViewerConfig(**VIEWER), vis=VIS),
        description="GaussCtrl")
else:
    @dataclass
    class MethodSpecification:
        """nerfstudio.plugins.types.MethodSpecification stand-in: plugin discovery reads `.config.method_name`."""
        config: GaussCtrlTrainerConfig
        description: str

    gaussctrl_method = MethodSpecification(
        config=GaussCtrlTrainerConfig(**TRAINER, pipeline=_pipeline_config(), optimizers=optimizer_table(), viewer=dict(VIEWER), vis=VIS),
        description="GaussCtrl")


def scheduled_lr(group: str, step: int, table: dict | None = None) -> float:
    """Learning rate of a parameter group at a trainer step: constant, or ExponentialDecayScheduler(lr_final, max_steps) for xyz
    and camera_opt (gc_config.py:59-66,83-86).  A run resumed from the step-30000 checkpoint sits at lr_final."""
    t = (table or optimizer_table())[group]
    lr = t["optimizer"]["lr"]
    sch = t["scheduler"]
    return lr if sch is None else exp_decay_lr(step, lr, sch["lr_final"], sch["max_steps"])


def build_optimizers(model, table: dict | None = None, step: int = 30000):
    """Adam per parameter group with the reference's learning rates / eps / schedules (gc_config.py:58-87) on the fused HIP Adam."""
    from .train_ops import FusedAdam
    table = table or optimizer_table()
    groups = model.get_param_groups()
    return {name: FusedAdam(params, lr=scheduled_lr(name, step, table), eps=table[name]["optimizer"]["eps"])
            for name, params in groups.items() if name in table}
