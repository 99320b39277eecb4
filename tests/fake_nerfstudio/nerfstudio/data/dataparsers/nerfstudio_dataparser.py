from dataclasses import dataclass, field

from nerfstudio.configs.base_config import InstantiateConfig


@dataclass
class NerfstudioDataParserConfig(InstantiateConfig):
    _target: type = field(default_factory=lambda: Nerfstudio)
    load_3D_points: bool = False


class Nerfstudio:
    def __init__(self, config):
        self.config = config
