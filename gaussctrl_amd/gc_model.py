"""GaussCtrlModel: drop-in for /root/reference/gaussctrl/gc_model.py (GaussCtrlModelConfig :39-50,
GaussCtrlModel.get_outputs :57-206, get_outputs_for_camera :208-221) on the HIP rasterizer.

Same contract: get_outputs(camera: Cameras[1]) -> {"rgb"[H,W,3], "depth"[H,W,1] | None,
"accumulation"[H,W,1]}; training -> random / configured background and no depth; eval -> rgb + depth +
accumulation from ONE fused compositing sweep (the reference sorts and rasterises twice, :174-202); returns a
background image when nothing is visible (:89-91,155-156); `self.xys` / `self.radii` are kept for the densification
callbacks and `self.xys.grad` is populated after backward (:159-160).  Parameter names / optimizer groups are
splatfacto's (`means, scales, quats, features_dc, features_rest, opacities`; groups xyz, features_dc, features_rest,
opacity, scaling, rotation -- /root/reference/gaussctrl/gc_config.py:58-87)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Union

import torch
from torch import nn

from . import gsplat_ops as ops
from .camera import camera_to_gsplat
from .ns_compat import HAVE_NERFSTUDIO, Cameras

if HAVE_NERFSTUDIO:  # the reference's bases (gc_model.py:39,52): SplatfactoModelConfig / SplatfactoModel
    from nerfstudio.models.splatfacto import SplatfactoModel as _ModelBase, SplatfactoModelConfig as _ModelConfigBase  # type: ignore
else:
    _ModelBase = nn.Module

    @dataclass
    class _ModelConfigBase:
        """the SplatfactoModelConfig fields get_outputs / the loss / the culling callback read [recall nerfstudio 1.0.0]"""
        sh_degree: int = 3
        sh_degree_interval: int = 1000
        background_color: str = "random"
        ssim_lambda: float = 0.2
        cull_alpha_thresh: float = 0.1
        cull_scale_thresh: float = 0.5
        refine_every: int = 100
        reset_alpha_every: int = 30
        stop_split_at: int = 15000
        continue_cull_post_densification: bool = True

        def setup(self, **kw):
            return self._target(self, **kw)


@dataclass
class GaussCtrlModelConfig(_ModelConfigBase):
    """gc_model.py:39-50: the four fields the reference adds (never read: it has no get_loss_dict override)."""
    _target: type = field(default_factory=lambda: GaussCtrlModel)
    use_lpips: bool = True
    use_l1: bool = True
    patch_size: int = 32
    lpips_loss_mult: float = 1.0


class GaussCtrlModel(_ModelBase):
    """Under nerfstudio: a SplatfactoModel whose get_outputs / get_outputs_for_camera / loss run on the HIP kernels (parameters,
    densification callbacks, checkpoint keys and param groups are splatfacto's own).  Stand-alone: the same methods on an
    nn.Module that holds the six splatfacto parameter tensors."""
    config: GaussCtrlModelConfig

    def __init__(self, config: GaussCtrlModelConfig, *args, params: Optional[Dict[str, torch.Tensor]] = None, num_points: int = 0,
                 device="cuda", **kwargs):
        if HAVE_NERFSTUDIO:
            super().__init__(config, *args, **kwargs)         # SplatfactoModel.populate_modules creates the parameters
            self._aux = ops.RenderAux()
            return
        super().__init__()
        self.config = config
        k = (config.sh_degree + 1) ** 2 - 1
        if params is None:
            params = {"means": torch.zeros(num_points, 3), "scales": torch.zeros(num_points, 3),
                      "quats": torch.zeros(num_points, 4), "opacities": torch.zeros(num_points, 1),
                      "features_dc": torch.zeros(num_points, 3), "features_rest": torch.zeros(num_points, k, 3)}
        for name in ("means", "scales", "quats", "features_dc", "features_rest", "opacities"):
            setattr(self, name, nn.Parameter(torch.as_tensor(params[name], dtype=torch.float32).to(device).contiguous()))
        self.step = 30000                        # a loaded splatfacto checkpoint starts here (gc_trainer.py:75)
        self.crop_box = None
        self.background_color = torch.zeros(3)
        self.xys = None
        self.radii = None
        self.last_size = None
        self._aux = ops.RenderAux()

    @property
    def device(self):
        return self.means.device

    @property
    def num_points(self):
        return self.means.shape[0]

    def set_crop(self, crop_box):
        self.crop_box = crop_box

    def get_param_groups(self) -> Dict[str, List[nn.Parameter]]:
        if HAVE_NERFSTUDIO:
            return super().get_param_groups()
        return {"xyz": [self.means], "features_dc": [self.features_dc], "features_rest": [self.features_rest],
                "opacity": [self.opacities], "scaling": [self.scales], "rotation": [self.quats]}

    def get_training_callbacks(self, training_callback_attributes):
        """SplatfactoModel's callbacks (step bookkeeping, after_train, refinement_after) under nerfstudio; stand-alone: the
        part of refinement_after that still acts after step 30000 (> stop_split_at = 15000): opacity / scale culling every
        `refine_every` steps [recall nerfstudio 1.0.0 splatfacto.py; SURVEY.md 3.5]."""
        if HAVE_NERFSTUDIO:
            return super().get_training_callbacks(training_callback_attributes)
        from .gc_trainer import CullCallback, StepCallback
        return [StepCallback(self), CullCallback(self, training_callback_attributes.optimizers)]

    # ------------------------------------------------------------------------------------ gc_model.py:57-206
    def get_outputs(self, camera: Cameras) -> Dict[str, Union[torch.Tensor, List]]:
        if not isinstance(camera, Cameras):
            print("Called get_outputs with not a camera")
            return {}
        assert camera.shape[0] == 1, "Only one camera at a time"
        dev = self.device
        if self.training:                                                       # :72-81
            bc = self.config.background_color
            background = (torch.rand(3, device=dev) if bc == "random" else torch.ones(3, device=dev) if bc == "white"
                          else torch.zeros(3, device=dev) if bc == "black" else self.background_color.to(dev))
            if bc == "random" and getattr(self, "background_override", None) is not None:
                background = self.background_override.to(dev)       # multi-GPU train_mode "parity": the colour rank 0 drew (GaussCtrlPipeline._sync_view)
        else:
            background = self.background_color.to(dev)
        W, H = int(camera.width.reshape(-1)[0]), int(camera.height.reshape(-1)[0])
        p = [self.means, self.scales, self.quats, self.opacities, self.features_dc, self.features_rest]
        if self.crop_box is not None and not self.training:                     # :88-93,123-136
            crop_ids = self.crop_box.within(self.means).squeeze()
            if crop_ids.sum() == 0:
                return {"rgb": background.repeat(H, W, 1)}
            p = [t[crop_ids] for t in p]
        cam = camera_to_gsplat(camera.camera_to_worlds[0].detach().cpu().numpy(), float(camera.fx.reshape(-1)[0]),
                               float(camera.fy.reshape(-1)[0]), float(camera.cx.reshape(-1)[0]),
                               float(camera.cy.reshape(-1)[0]), W, H)        # :97-121 on the host
        self.last_size = (H, W)
        # :162-169: SH of degree n (+0.5, clamp) when config.sh_degree > 0, else sigmoid(features_dc) (encoded as n = -1)
        n = min(self.step // self.config.sh_degree_interval, self.config.sh_degree) if self.config.sh_degree > 0 else -1
        aux = self._aux = ops.RenderAux()
        if self.training and getattr(self, "grad_into", None) is not None and self.crop_box is None:
            # the fused backward writes the six leaf gradients straight into the caller's buffers (dist.FlatGrads: ONE flat allocation that
            # RCCL reduces in place) and autograd gets None for them: GaussCtrlPipeline.train_iteration, train_mode "throughput"
            aux.grad_into, aux.grad_accumulate = self.grad_into, False
        rgb, alpha, depth = ops.render_view(*p, cam, background, not self.training, n, aux)
        self.xys, self.radii = aux.xys, aux.radii
        if aux.M == 0:                                                          # :155-156
            return {"rgb": background.repeat(H, W, 1)}
        depth_im = None if self.training else depth[..., None]
        return {"rgb": rgb, "depth": depth_im, "accumulation": alpha[..., None]}

    forward = get_outputs

    if HAVE_NERFSTUDIO:
        def cull_gaussians(self, *args, **kwargs):
            """SplatfactoModel.cull_gaussians + a record of the row mask it applied: train_mode "sharded" prunes its 1 / N optimizer-state
            slices with it (GaussCtrlPipeline._sharded_adam), as splatfacto's own remove_from_all_optim does for the replicated Adam."""
            culls = super().cull_gaussians(*args, **kwargs)
            if torch.is_tensor(culls):
                self._cull_keep = ~culls
            return culls

    @property
    def xys_grad(self):
        """gradient of the loss w.r.t. the projected centres (what splatfacto's after_train reads as xys.grad)."""
        return self._aux.xys_grad

    @torch.no_grad()
    def get_outputs_for_camera(self, camera: Cameras, obb_box=None) -> Dict[str, torch.Tensor]:    # :208-221
        assert camera is not None, "must provide camera to gaussian model"
        self.set_crop(obb_box)
        self.training = False
        outs = self.get_outputs(camera.to(self.device))
        self.training = True
        return outs

    @torch.no_grad()
    def get_outputs_for_cameras(self, cameras: List[Cameras], obb_box=None) -> List[Dict[str, torch.Tensor]]:
        """get_outputs_for_camera for a LIST of cameras of this scene through ONE batched launch set (ops.render_views: the parameter
        record is read once per batch, the sort / binning / compositing kernels run with the view as a grid dimension).  The reference has
        no such call -- render_reverse asks for one camera at a time (gc_pipeline.py:124-130); per view the outputs are bit-identical to
        get_outputs_for_camera.  Falls back to it when a crop box is set or the cameras differ in size."""
        cams = [c.to(self.device) for c in cameras]
        sizes = {(int(c.width.reshape(-1)[0]), int(c.height.reshape(-1)[0])) for c in cams}
        if obb_box is not None or len(cams) < 2 or len(sizes) != 1 or max(sizes.pop()) > 255 * 16:
            return [self.get_outputs_for_camera(c, obb_box) for c in cameras]
        self.set_crop(None)
        W, H = int(cams[0].width.reshape(-1)[0]), int(cams[0].height.reshape(-1)[0])
        gcams = [camera_to_gsplat(c.camera_to_worlds[0].detach().cpu().numpy(), float(c.fx.reshape(-1)[0]), float(c.fy.reshape(-1)[0]),
                                  float(c.cx.reshape(-1)[0]), float(c.cy.reshape(-1)[0]), W, H) for c in cams]
        background = self.background_color.to(self.device)
        n = min(self.step // self.config.sh_degree_interval, self.config.sh_degree) if self.config.sh_degree > 0 else -1
        aux = self._aux = ops.RenderAux()
        rgb, alpha, depth = ops.render_views(self.means, self.scales, self.quats, self.opacities, self.features_dc, self.features_rest,
                                             gcams, background, True, n, aux)
        self.last_size = (H, W)
        cnt = aux.M[0].cpu()
        outs = []
        for k in range(len(cams)):
            if int(cnt[k]) == 0:                                                # :155-156
                outs.append({"rgb": background.repeat(H, W, 1)})
            else:
                outs.append({"rgb": rgb[k], "depth": depth[k][..., None], "accumulation": alpha[k][..., None]})
        return outs

    # ------------------------------------------------------------------------------------ inherited splatfacto loss
    def get_metrics_dict(self, outputs, batch) -> Dict[str, torch.Tensor]:
        gt = batch["image"].to(self.device)
        mse = ((outputs["rgb"] - gt) ** 2).mean()
        return {"psnr": -10.0 * torch.log10(mse.clamp_min(1e-12)), "gaussian_count": torch.tensor(self.num_points)}

    def get_loss_dict(self, outputs, batch, metrics_dict=None) -> Dict[str, torch.Tensor]:
        """SplatfactoModel.get_loss_dict [recall, SURVEY 8a/A8]: (1-l)*L1 + l*(1-SSIM), l = 0.2."""
        from .train_ops import l1_ssim_loss
        gt = batch["image"].to(self.device)
        return {"main_loss": l1_ssim_loss(outputs["rgb"], gt, self.config.ssim_lambda)}
