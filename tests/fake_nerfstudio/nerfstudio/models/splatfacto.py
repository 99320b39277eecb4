from dataclasses import dataclass, field

import torch
from torch import nn

from nerfstudio.configs.base_config import InstantiateConfig
from nerfstudio.engine.callbacks import TrainingCallback, TrainingCallbackLocation


@dataclass
class SplatfactoModelConfig(InstantiateConfig):
    _target: type = field(default_factory=lambda: SplatfactoModel)
    sh_degree: int = 3
    sh_degree_interval: int = 1000
    background_color: str = "random"
    ssim_lambda: float = 0.2
    num_random: int = 64


class SplatfactoModel(nn.Module):
    def __init__(self, config, scene_box=None, num_train_data=0, device="cpu", **kwargs):
        super().__init__()
        self.config = config
        n = config.num_random
        k = (config.sh_degree + 1) ** 2 - 1
        self.means = nn.Parameter(torch.zeros(n, 3)); self.scales = nn.Parameter(torch.zeros(n, 3))
        self.quats = nn.Parameter(torch.zeros(n, 4)); self.opacities = nn.Parameter(torch.zeros(n, 1))
        self.features_dc = nn.Parameter(torch.zeros(n, 3)); self.features_rest = nn.Parameter(torch.zeros(n, k, 3))
        self.step = 0
        self.crop_box = None
        self.background_color = torch.zeros(3)

    def get_param_groups(self):
        return {"xyz": [self.means], "features_dc": [self.features_dc], "features_rest": [self.features_rest],
                "opacity": [self.opacities], "scaling": [self.scales], "rotation": [self.quats]}

    def step_cb(self, step):
        self.step = step

    def get_training_callbacks(self, attrs):
        return [TrainingCallback([TrainingCallbackLocation.BEFORE_TRAIN_ITERATION], self.step_cb)]
