from dataclasses import dataclass


@dataclass
class MethodSpecification:
    config: object
    description: str
