"""Dataparser config of the method (/root/reference/gaussctrl/gc_config.py:54 uses `GaussCtrlDataParserConfig(load_3D_points=True)`).

The reference's gc_dataparser_ns.py is a near-copy of nerfstudio's Nerfstudio dataparser (transforms.json -> Cameras, ply seed
points) plus optional mid-result directories; parsing is host I/O and out of scope.  Under nerfstudio the stock
parser is subclassed so the config tree of `ns-train gaussctrl` resolves; the mid-result cache (depth_npy / z_0 / mask_npy /
unedited, gc_dataparser_ns.py:408-420) is read and written by gaussctrl_amd.midcache instead."""
from __future__ import annotations

from dataclasses import dataclass, field

from .ns_compat import HAVE_NERFSTUDIO

if HAVE_NERFSTUDIO:  # pragma: no cover
    from nerfstudio.data.dataparsers.nerfstudio_dataparser import Nerfstudio, NerfstudioDataParserConfig  # type: ignore

    @dataclass
    class GaussCtrlDataParserConfig(NerfstudioDataParserConfig):
        _target: type = field(default_factory=lambda: GaussCtrlDataParser)
        load_3D_points: bool = True

    class GaussCtrlDataParser(Nerfstudio):
        config: GaussCtrlDataParserConfig
else:
    @dataclass
    class GaussCtrlDataParserConfig:
        load_3D_points: bool = True
