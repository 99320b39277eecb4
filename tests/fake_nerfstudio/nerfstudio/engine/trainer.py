from dataclasses import dataclass, field
from typing import Dict

from nerfstudio.configs.base_config import InstantiateConfig, ViewerConfig
from nerfstudio.engine.callbacks import TrainingCallbackAttributes


@dataclass
class TrainerConfig(InstantiateConfig):
    _target: type = field(default_factory=lambda: Trainer)
    method_name: str = None
    steps_per_save: int = 1000
    steps_per_eval_batch: int = 500
    steps_per_eval_image: int = 500
    steps_per_eval_all_images: int = 25000
    max_num_iterations: int = 1000000
    mixed_precision: bool = False
    save_only_latest_checkpoint: bool = True
    gradient_accumulation_steps: Dict[str, int] = field(default_factory=dict)
    pipeline: object = None
    optimizers: Dict[str, dict] = field(default_factory=dict)
    viewer: ViewerConfig = field(default_factory=ViewerConfig)
    vis: str = "wandb"

    def setup(self, local_rank=0, world_size=1, **kw):
        return self._target(self, local_rank=local_rank, world_size=world_size)


class Trainer:
    def __init__(self, config, local_rank=0, world_size=1):
        self.config, self.local_rank, self.world_size = config, local_rank, world_size
        self.device = "cpu"
        self._start_step = 0
        self.grad_scaler = None
        self.trained_steps = []

    def setup(self, test_mode="val"):
        self.pipeline = self.config.pipeline.setup(device=self.device, test_mode=test_mode, world_size=self.world_size,
                                                   local_rank=self.local_rank, grad_scaler=self.grad_scaler)
        self.optimizers = dict(self.config.optimizers)
        self._start_step = 30000           # _load_checkpoint()
        self.callbacks = self.pipeline.get_training_callbacks(TrainingCallbackAttributes(self.optimizers, self.grad_scaler, self.pipeline))

    def train(self):
        for step in range(self._start_step, self._start_step + self.config.max_num_iterations):
            self.trained_steps.append(step)
