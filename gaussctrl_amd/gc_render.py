"""ns-gaussctrl-render stand-in (/root/reference/gaussctrl/gc_render.py:875-892 registers `entrypoint`).

The reference file is a copy of nerfstudio's ns-render (video / trajectory tooling, out of scope: SURVEY.md 2.1 #9);
its only hot-path touch is `pipeline.model.get_outputs_for_camera(...)` (gc_render.py:196-203,807) plus dumping depth as
.npy (gc_render.py:217-221,833-838).  This module keeps that caller: render a list of cameras to rgb / depth arrays."""
from __future__ import annotations

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


def entrypoint():  # pragma: no cover
    raise SystemExit("ns-gaussctrl-render needs nerfstudio's CLI (tyro / mediapy); use gaussctrl_amd.gc_render.render_cameras")
