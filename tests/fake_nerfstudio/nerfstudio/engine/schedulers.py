from dataclasses import dataclass


@dataclass
class ExponentialDecaySchedulerConfig:
    lr_pre_warmup: float = 1e-8
    lr_final: float = None
    warmup_steps: int = 0
    max_steps: int = 100000
