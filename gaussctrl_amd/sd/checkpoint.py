"""Loading the diffusion checkpoints the reference loads through diffusers (/root/reference/gaussctrl/gc_pipeline.py:97-102:
`StableDiffusionControlNetPipeline.from_pretrained(config.diffusion_ckpt, controlnet=ControlNetModel.from_pretrained(
"lllyasviel/sd-controlnet-depth"))`) WITHOUT diffusers: the diffusers on-disk layout is read directly
(`unet/`, `vae/`, `text_encoder/`, `tokenizer/` sub-folders with `diffusion_pytorch_model.safetensors` / `.bin`), state-dict keys
are diffusers' own (what gaussctrl_amd.sd.arch lists), the CLIP text tower comes from `transformers`.

A checkpoint argument is a local directory, or a hub id resolved through the local Hugging Face cache (`huggingface_hub`
with local_files_only: the build / GPU boxes have no network).  Nothing here falls back to random weights: a missing file raises."""
from __future__ import annotations

import os

import torch

from . import arch

_VAE_ATTN_OLD = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}      # pre-0.15 diffusers VAE key names


def resolve(ckpt: str) -> str:
    """local directory, else the local Hugging Face cache, else (unless HF_HUB_OFFLINE is set) a normal hub download -- what the
    reference's `from_pretrained` does on a machine with network access and an empty cache"""
    if os.path.isdir(ckpt):
        return ckpt
    try:
        from huggingface_hub import snapshot_download
    except Exception as e:  # noqa: BLE001
        raise FileNotFoundError(f"diffusion checkpoint '{ckpt}' is not a local directory and huggingface_hub is unavailable: {e}") from e
    try:
        return snapshot_download(ckpt, local_files_only=True)
    except Exception as e_local:  # noqa: BLE001
        if os.environ.get("HF_HUB_OFFLINE", "0") not in ("", "0"):
            raise FileNotFoundError(f"diffusion checkpoint '{ckpt}' is neither a local directory nor in the local Hugging Face cache "
                                    f"(HF_HUB_OFFLINE is set): {e_local}") from e_local
        try:
            return snapshot_download(ckpt)
        except Exception as e:  # noqa: BLE001
            raise FileNotFoundError(f"diffusion checkpoint '{ckpt}' is neither a local directory nor in the local Hugging Face cache, "
                                    f"and downloading it failed (no network?): {e}") from e


def _load_file(folder: str) -> dict:
    for name in ("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.fp16.safetensors", "diffusion_pytorch_model.bin"):
        p = os.path.join(folder, name)
        if os.path.exists(p):
            if p.endswith(".safetensors"):
                from safetensors.torch import load_file
                return load_file(p)
            return torch.load(p, map_location="cpu", weights_only=True)
    raise FileNotFoundError(f"no diffusion_pytorch_model.[safetensors|bin] under {folder}")


def _vae_keys(sd: dict) -> dict:
    out = {}
    for k, v in sd.items():
        parts = k.split(".")
        if "attentions" in parts:
            for old, new in _VAE_ATTN_OLD.items():
                if parts[-2] == old:
                    k = ".".join(parts[:-2] + [new, parts[-1]])
        if v.dim() == 4 and v.shape[-1] == 1 and ".attentions." in k:          # very old VAEs store the attention linears as 1x1 convs
            v = v[:, :, 0, 0]
        out[k] = v
    return out


def load_diffusion_weights(diffusion_ckpt: str, controlnet_ckpt: str = "lllyasviel/sd-controlnet-depth") -> dict:
    """-> {"unet", "controlnet", "vae_decoder", "vae_encoder"}: fp32/fp16 CPU state dicts with diffusers key names, shape-checked
    against the SD1.x / sd-controlnet-depth inventories of gaussctrl_amd.sd.arch."""
    root = resolve(diffusion_ckpt)
    unet = _load_file(os.path.join(root, "unet"))
    vae = _vae_keys(_load_file(os.path.join(root, "vae")))
    croot = resolve(controlnet_ckpt)
    cn = _load_file(croot if not os.path.isdir(os.path.join(croot, "controlnet")) else os.path.join(croot, "controlnet"))
    out = {"unet": {k: v for k, v in unet.items() if k in arch.unet_shapes()},
           "controlnet": {k: v for k, v in cn.items() if k in arch.controlnet_shapes()},
           "vae_decoder": {k: v for k, v in vae.items() if k in arch.vae_decoder_shapes()},
           "vae_encoder": {k: v for k, v in vae.items() if k in arch.vae_encoder_shapes()}}
    arch.check_state_dict(out["unet"], arch.unet_shapes())
    arch.check_state_dict(out["controlnet"], arch.controlnet_shapes())
    arch.check_state_dict(out["vae_decoder"], arch.vae_decoder_shapes())
    arch.check_state_dict(out["vae_encoder"], arch.vae_encoder_shapes())
    return out


def load_text_encoder(diffusion_ckpt: str, device="cpu"):
    """-> prompt -> [1,77,768] CLIP text embedding (the `encode_prompt` of the diffusers pipeline: last hidden state, max_length 77)."""
    from transformers import CLIPTextModel, CLIPTokenizer
    root = resolve(diffusion_ckpt)
    tok = CLIPTokenizer.from_pretrained(os.path.join(root, "tokenizer"))
    enc = CLIPTextModel.from_pretrained(os.path.join(root, "text_encoder")).to(device).eval()

    @torch.no_grad()
    def encode(prompt: str) -> torch.Tensor:
        ids = tok(prompt, padding="max_length", max_length=tok.model_max_length, truncation=True, return_tensors="pt").input_ids
        return enc(ids.to(device))[0].float()
    return encode
