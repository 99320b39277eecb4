from dataclasses import dataclass, field


@dataclass
class InstantiateConfig:
    _target: type = None

    def setup(self, **kwargs):
        return self._target(self, **kwargs)


@dataclass
class ViewerConfig:
    num_rays_per_chunk: int = 32768
    quit_on_train_completion: bool = False
