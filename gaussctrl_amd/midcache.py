"""On-disk mid-result cache in the reference's layout, so scenes prepared by either implementation interchange.

The reference's dataparser picks these folders up when they exist under the scene directory
(/root/reference/gaussctrl/gc_dataparser_ns.py:408-420) and its dataset reads them back with np.load / PIL
(/root/reference/gaussctrl/gc_dataset.py:36-69,129-158); the render CLI writes depth the same way
(/root/reference/gaussctrl/gc_render.py:217-221,833-838).  Frames are numbered from 1:

    depth_npy/frame_%05d.npy   float32 [H, W, 1]    rendered depth
    z_0/frame_%05d.npy         float32 [1, 4, h, w] DDIM-inverted latent (h = H/8)
    mask_npy/frame_%05d.npy    bool / {0,1} [H, W]  LangSAM mask (only when a mask prompt was used)
    unedited/frame_%05d.jpg    uint8 RGB            render of the un-edited scene

plus the splatfacto checkpoint keys (means, scales, quats, features_dc, features_rest, opacities under `_model.`;
/root/reference/gaussctrl/gc_trainer.py:146-174 stores {"step", "pipeline": state_dict, ...}).
Host-side I/O only: nothing here is on the timed path.
"""
from __future__ import annotations

import os
from pathlib import Path

import numpy as np
import torch

FOLDERS = {"depth": "depth_npy", "z_0": "z_0", "mask": "mask_npy", "unedited": "unedited"}
CKPT_KEYS = ("means", "scales", "quats", "features_dc", "features_rest", "opacities")


def frame_name(idx: int, ext: str) -> str:
    return f"frame_{idx + 1:05d}.{ext}"


def _np(t):
    return t.detach().to("cpu", torch.float32).numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def save_view(root, idx: int, unedited_image=None, depth=None, z_0=None, mask=None, jpeg_quality: int = 95) -> None:
    """Write one view's mid results.  unedited_image [H,W,3] in [0,1]; depth [H,W] or [H,W,1]; z_0 [1,4,h,w] or [4,h,w]."""
    root = Path(root)
    if depth is not None:
        d = _np(depth).astype(np.float32)
        d = d[..., None] if d.ndim == 2 else d
        (root / FOLDERS["depth"]).mkdir(parents=True, exist_ok=True)
        np.save(root / FOLDERS["depth"] / frame_name(idx, "npy"), d)
    if z_0 is not None:
        z = _np(z_0).astype(np.float32)
        z = z[None] if z.ndim == 3 else z
        (root / FOLDERS["z_0"]).mkdir(parents=True, exist_ok=True)
        np.save(root / FOLDERS["z_0"] / frame_name(idx, "npy"), z)
    if mask is not None:
        m = _np(mask)
        (root / FOLDERS["mask"]).mkdir(parents=True, exist_ok=True)
        np.save(root / FOLDERS["mask"] / frame_name(idx, "npy"), (m > 0.5) if m.dtype != np.bool_ else m)
    if unedited_image is not None:
        from PIL import Image
        img = np.clip(_np(unedited_image) * 255.0 + 0.5, 0, 255).astype(np.uint8)
        (root / FOLDERS["unedited"]).mkdir(parents=True, exist_ok=True)
        Image.fromarray(img, "RGB").save(root / FOLDERS["unedited"] / frame_name(idx, "jpg"), quality=jpeg_quality)


def has_view(root, idx: int) -> bool:
    root = Path(root)
    return all((root / FOLDERS[k] / frame_name(idx, e)).exists() for k, e in (("depth", "npy"), ("z_0", "npy"), ("unedited", "jpg")))


def load_view(root, idx: int, device="cpu") -> dict:
    """Read one view back with the dataset's conventions (gc_dataset.py:36-69,129-158): depth_image [1,H,W] (the reference drops
    the channel and prepends a batch axis), z_0_image [1,4,h,w], mask_image as stored, unedited_image float32 [H,W,3] in [0,1]."""
    root = Path(root)
    out = {}
    p = root / FOLDERS["depth"] / frame_name(idx, "npy")
    if p.exists():
        out["depth_image"] = torch.from_numpy(np.load(p)[:, :, 0][None].astype(np.float32)).to(device)
    p = root / FOLDERS["z_0"] / frame_name(idx, "npy")
    if p.exists():
        out["z_0_image"] = torch.from_numpy(np.load(p).astype(np.float32)).to(device)
    p = root / FOLDERS["mask"] / frame_name(idx, "npy")
    if p.exists():
        out["mask_image"] = torch.from_numpy(np.load(p)).to(device)
    p = root / FOLDERS["unedited"] / frame_name(idx, "jpg")
    if p.exists():
        from PIL import Image
        out["unedited_image"] = torch.from_numpy(np.asarray(Image.open(p).convert("RGB"), dtype=np.uint8).astype(np.float32) / 255.0).to(device)
    return out


def save_checkpoint(path, step: int, model_params: dict, optimizers: dict | None = None) -> None:
    """nerfstudio-style checkpoint: {"step", "pipeline": {"_model.<key>": tensor}, "optimizers": {...}} (gc_trainer.py:146-174)."""
    missing = [k for k in CKPT_KEYS if k not in model_params]
    if missing:
        raise KeyError(f"checkpoint needs the splatfacto parameters {missing}")
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save({"step": int(step),
                "pipeline": {f"_model.{k}": model_params[k].detach().cpu() for k in CKPT_KEYS},
                "optimizers": {k: v.state_dict() for k, v in (optimizers or {}).items()}}, path)


def load_checkpoint(path, device="cpu") -> tuple:
    """-> (step, {key: tensor}) accepting both `_model.` and `module._model.` (DDP) prefixes."""
    ck = torch.load(path, map_location=device, weights_only=False)
    sd = ck["pipeline"]
    out = {}
    for k in CKPT_KEYS:
        for pre in ("_model.", "module._model.", "_model.gauss_params.", ""):
            if pre + k in sd:
                out[k] = sd[pre + k].to(device)
                break
        else:
            raise KeyError(f"checkpoint has no parameter {k}")
    return int(ck.get("step", 0)), out
