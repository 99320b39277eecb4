from dataclasses import dataclass, field

from torch import nn

from nerfstudio.configs.base_config import InstantiateConfig


@dataclass
class VanillaPipelineConfig(InstantiateConfig):
    _target: type = field(default_factory=lambda: VanillaPipeline)
    datamanager: object = None
    model: object = None


class VanillaPipeline(nn.Module):
    def __init__(self, config, device, test_mode="val", world_size=1, local_rank=0, grad_scaler=None):
        super().__init__()
        self.config, self.test_mode = config, test_mode
        self.datamanager = config.datamanager.setup(device=device, test_mode=test_mode, world_size=world_size, local_rank=local_rank)
        self._model = config.model.setup(scene_box=None, num_train_data=len(self.datamanager.train_dataset), device=device)
        self.world_size = world_size

    @property
    def model(self):
        return self._model

    @property
    def device(self):
        """read-only, like nerfstudio's Pipeline.device: assigning to it raises AttributeError"""
        return getattr(self._model, "device", None)

    def get_training_callbacks(self, attrs):
        return self.datamanager.get_training_callbacks(attrs) + self.model.get_training_callbacks(attrs)
