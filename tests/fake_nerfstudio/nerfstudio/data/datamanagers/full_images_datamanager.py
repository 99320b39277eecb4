from dataclasses import dataclass, field

import numpy as np
import torch

from nerfstudio.cameras.cameras import Cameras
from nerfstudio.configs.base_config import InstantiateConfig


@dataclass
class FullImageDatamanagerConfig(InstantiateConfig):
    _target: type = field(default_factory=lambda: FullImageDatamanager)
    dataparser: object = None
    num_fake_views: int = 57


class _Dataset:
    def __init__(self, n):
        c2w = np.tile(np.eye(4, dtype=np.float32)[:3], (n, 1, 1))
        self.cameras = Cameras(c2w, 100.0, 100.0, 16.0, 16.0, 32, 32)
        self.n = n

    def __len__(self):
        return self.n


class FullImageDatamanager:
    def __init__(self, config, device="cpu", test_mode="val", world_size=1, local_rank=0, **kwargs):
        self.config, self.device = config, device
        self.train_dataset = _Dataset(config.num_fake_views)
        self.cached_train = [{"image": torch.zeros(32, 32, 3), "image_idx": i} for i in range(config.num_fake_views)]

    def get_training_callbacks(self, attrs):
        return []
