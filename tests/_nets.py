"""Session-wide caches of the seeded SD1.5-shaped test weights (tests only).  Generating the 1.2 G parameters of the UNet + ControlNet
from their CPU generators takes 10-25 s on the GPU boxes and a dozen tests want the same tensors; preparing them for the device (layout,
LayerNorm fold algebra) a few seconds more.  Everything here is a pure function of (seed, dtype, fold level): sharing changes no result."""
import torch

DEV = "cuda:0"
_RAW, _PREP = {}, {}


def raw_sd15(rounded: bool = True):
    """(unet, controlnet) fp32 state dicts of oracle.sd15_torch (seeds 100 / 200); rounded: values rounded to bf16, kept in fp32 (what the
    full-geometry fixtures were generated with)"""
    from oracle import sd15_torch as sd
    if "raw" not in _RAW:
        _RAW["raw"] = (sd.make_unet_weights(sd.SD15, 100), sd.make_controlnet_weights(sd.SD15, 200))
    if not rounded:
        return _RAW["raw"]
    if "bf16" not in _RAW:
        _RAW["bf16"] = tuple({k: v.to(torch.bfloat16).float() for k, v in w.items()} for w in _RAW["raw"])
    return _RAW["bf16"]


def prepared(dt, fold_ln=False):
    """(unet, controlnet) device weight sets of gaussctrl_amd.sd.weights.prepare on the bf16-rounded raw weights; callers that add to them
    (add_fp8_convs / add_fp8_linears) must copy the dicts first"""
    from gaussctrl_amd.sd.weights import prepare
    key = (dt, fold_ln)
    if key not in _PREP:
        usd, csd = raw_sd15(True)
        _PREP[key] = (prepare(usd, dt, DEV, heads=8, fold_ln=fold_ln), prepare(csd, dt, DEV, heads=8, fold_ln=fold_ln))
    return _PREP[key]


def conv_weights():
    """the resnet 3x3 convolution weights of the rounded raw state dicts (what add_fp8_convs quantises)"""
    if "convs" not in _RAW:
        _RAW["convs"] = tuple({k: v for k, v in w.items() if k.endswith((".conv1.weight", ".conv2.weight"))} for w in raw_sd15(True))
    return _RAW["convs"]
