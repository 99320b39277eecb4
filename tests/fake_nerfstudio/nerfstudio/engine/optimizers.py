from dataclasses import dataclass


@dataclass
class AdamOptimizerConfig:
    lr: float = 0.0005
    eps: float = 1e-08
    weight_decay: float = 0
