"""ns-gaussctrl-render (/root/reference/gaussctrl/gc_render.py:875-892 registers `entrypoint`).

The reference file is a copy of nerfstudio's ns-render (video / trajectory tooling, out of scope: SURVEY.md 2.1 #9);
its only hot-path touch is `pipeline.model.get_outputs_for_camera(...)` (gc_render.py:196-203,807) plus dumping depth as
.npy (gc_render.py:217-221,833-838).  This module keeps that caller -- `render_cameras` -- and a small command line around it:

    ns-gaussctrl-render dataset --load-config <config.yml> --output-path <dir>            (under nerfstudio: its eval_setup)
    ns-gaussctrl-render dataset --load-gaussians scene.npz --cameras cams.json --output-path <dir>   (stand-alone)

Outputs per frame i (1-based, the reference's mid-result layout): rgb/frame_%05d.npy + .ppm, depth_npy/frame_%05d.npy."""
from __future__ import annotations

import argparse
import json
import os

import numpy as np


def render_cameras(model, cameras, out_dir=None):
    """-> list of dicts(rgb [H,W,3], depth [H,W,1], accumulation [H,W,1]) as numpy; depth also saved as
    depth_npy/frame_%05d.npy when out_dir is given (the mid-result layout of gc_dataparser_ns.py:408-420)."""
    outs = []
    for i in range(len(cameras)):
        o = model.get_outputs_for_camera(cameras[i])
        o = {k: v.detach().cpu().numpy() for k, v in o.items() if v is not None}
        if out_dir is not None and "depth" in o:
            os.makedirs(os.path.join(out_dir, "depth_npy"), exist_ok=True)
            np.save(os.path.join(out_dir, "depth_npy", f"frame_{i + 1:05d}.npy"), o["depth"])
        outs.append(o)
    return outs


def _write_ppm(path, rgb):
    img = (np.clip(rgb, 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8)
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(img.tobytes())


def load_cameras(path):
    """cameras.json: {"frames": [{"camera_to_world": 3x4 | 4x4, "fx", "fy", "cx", "cy", "w", "h"}, ...]} -> Cameras"""
    from .ns_compat import Cameras
    fr = json.load(open(path))["frames"]
    g = lambda k: [float(f[k]) for f in fr]
    return Cameras(np.asarray([np.asarray(f["camera_to_world"], np.float32)[:3, :4] for f in fr]), g("fx"), g("fy"), g("cx"), g("cy"),
                   [int(f["w"]) for f in fr], [int(f["h"]) for f in fr])


def load_model(npz_path, device="cuda"):
    """scene.npz with the six splatfacto tensors (means scales quats opacities features_dc features_rest) -> GaussCtrlModel"""
    from .gc_model import GaussCtrlModel, GaussCtrlModelConfig
    from .ns_compat import HAVE_NERFSTUDIO
    z = np.load(npz_path)
    params = {k: z[k] for k in ("means", "scales", "quats", "opacities", "features_dc", "features_rest")}
    cfg = GaussCtrlModelConfig()
    cfg.sh_degree = int(round((params["features_rest"].shape[1] + 1) ** 0.5)) - 1
    if HAVE_NERFSTUDIO:
        raise SystemExit("under nerfstudio load the scene with --load-config (a splatfacto / gaussctrl training run)")
    return GaussCtrlModel(cfg, params=params, device=device)


def entrypoint(argv=None):
    ap = argparse.ArgumentParser(prog="ns-gaussctrl-render", description=__doc__.split("\n\n")[0])
    ap.add_argument("mode", choices=["dataset"], help="render every camera of the dataset / camera file")
    ap.add_argument("--load-config", help="nerfstudio run config (needs nerfstudio)")
    ap.add_argument("--load-gaussians", help="stand-alone: .npz with the six splatfacto tensors")
    ap.add_argument("--cameras", help="stand-alone: cameras.json")
    ap.add_argument("--output-path", required=True)
    a = ap.parse_args(argv)
    if a.load_config:
        from .ns_compat import HAVE_NERFSTUDIO
        if not HAVE_NERFSTUDIO:
            raise SystemExit("--load-config needs nerfstudio; use --load-gaussians / --cameras")
        from pathlib import Path
        from nerfstudio.utils.eval_utils import eval_setup                       # pragma: no cover
        _, pipeline, _, _ = eval_setup(Path(a.load_config), test_mode="inference")  # pragma: no cover
        model, cameras = pipeline.model, pipeline.datamanager.train_dataset.cameras   # pragma: no cover
    else:
        if not (a.load_gaussians and a.cameras):
            raise SystemExit("give --load-config, or --load-gaussians and --cameras")
        model, cameras = load_model(a.load_gaussians), load_cameras(a.cameras)
    outs = render_cameras(model, cameras, a.output_path)
    os.makedirs(os.path.join(a.output_path, "rgb"), exist_ok=True)
    for i, o in enumerate(outs):
        np.save(os.path.join(a.output_path, "rgb", f"frame_{i + 1:05d}.npy"), o["rgb"])
        _write_ppm(os.path.join(a.output_path, "rgb", f"frame_{i + 1:05d}.ppm"), o["rgb"])
    print(f"ns-gaussctrl-render: {len(outs)} frames -> {a.output_path}")
    return 0


if __name__ == "__main__":
    raise SystemExit(entrypoint())
