"""Loss and optimiser of the 500-iteration splat optimisation on HIP kernels.

  l1_ssim_loss : SplatfactoModel.get_loss_dict's main loss, (1-l)*L1 + l*(1-SSIM) with l = 0.2 (inherited by the reference via
                 /root/reference/gaussctrl/gc_pipeline.py:284-285), forward value AND gradient w.r.t. the render in two launches;
  FusedAdam    : the Adam groups of /root/reference/gaussctrl/gc_config.py:58-87 (eps 1e-15) as one fused read-modify-write per
                 tensor (torch.optim.Adam semantics, interface-compatible: param_groups / step / zero_grad / state_dict).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


class _L1SSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, lam, valid):
        if not pred.is_cuda:
            raise L.GaussCtrlHipError("l1_ssim_loss needs GPU tensors (HIP path only; no CPU fallback)")
        lib = L.lib()
        H, W, Cc = pred.shape
        p = pred.detach().float().contiguous(); t = target.detach().float().contiguous()
        nbytes = lib.gc_l1_ssim_workspace_bytes(H, W, Cc)
        ws = torch.empty(nbytes // 4 + 1, dtype=torch.float32, device=p.device)
        sums = torch.empty(2, dtype=torch.float32, device=p.device)
        v = torch.empty_like(p)
        L.check(lib.gc_l1_ssim_fwd_bwd(L.ptr(p), L.ptr(t), H, W, Cc, L.f32(lam), L.f32(1.0), int(valid), L.ptr(sums), L.ptr(v), L.ptr(ws),
                                       C.c_size_t(nbytes), L.stream_ptr()), "gc_l1_ssim_fwd_bwd")
        ctx.save_for_backward(v)
        n = float(H * W * Cc)
        n_ssim = float((H - 10) * (W - 10) * Cc) if valid else n
        return (1.0 - lam) * sums[1] / n + lam * (1.0 - sums[0] / n_ssim)

    @staticmethod
    def backward(ctx, g):
        (v,) = ctx.saved_tensors
        return v * g, None, None, None


def l1_ssim_loss(pred, target, ssim_lambda: float = 0.2, valid_window: bool = True):
    """pred, target: [H,W,3] float32 on the GPU -> scalar loss tensor (differentiable w.r.t. pred).
    valid_window=True: SSIM as pytorch_msssim.SSIM(data_range=1, channel=3) computes it (unpadded 11x11 gaussian windows, mean over
    the (H-10) x (W-10) interior) -- the module splatfacto's get_loss_dict calls [recall nerfstudio 1.0.0 splatfacto.py]."""
    return _L1SSIM.apply(pred, target, float(ssim_lambda), bool(valid_window))


class _L1SSIMViews(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, lam, valid):
        if not pred.is_cuda:
            raise L.GaussCtrlHipError("l1_ssim_loss_views needs GPU tensors (HIP path only; no CPU fallback)")
        lib = L.lib()
        B, H, W, Cc = pred.shape
        p = pred.detach().float().contiguous(); t = target.detach().float().contiguous()
        nbytes = lib.gc_l1_ssim_views_workspace_bytes(B, H, W, Cc)
        ws = torch.empty(nbytes // 4 + 1, dtype=torch.float32, device=p.device)
        sums = torch.empty(B, 2, dtype=torch.float32, device=p.device)
        v = torch.empty_like(p)
        L.check(lib.gc_l1_ssim_fwd_bwd_views(B, L.ptr(p), L.ptr(t), H, W, Cc, L.f32(lam), L.f32(1.0), int(valid), L.ptr(sums), L.ptr(v), L.ptr(ws),
                                             C.c_size_t(nbytes), L.stream_ptr()), "gc_l1_ssim_fwd_bwd_views")
        ctx.save_for_backward(v)
        n = float(H * W * Cc)
        n_ssim = float((H - 10) * (W - 10) * Cc) if valid else n
        return (1.0 - lam) * sums[:, 1] / n + lam * (1.0 - sums[:, 0] / n_ssim)

    @staticmethod
    def backward(ctx, g):
        (v,) = ctx.saved_tensors
        return v * g.reshape(-1, 1, 1, 1), None, None, None


def l1_ssim_loss_views(pred, target, ssim_lambda: float = 0.2, valid_window: bool = True):
    """pred, target: [B,H,W,3] float32 -> [B] losses (each view's own l1_ssim_loss), one pair of launches for the batch."""
    return _L1SSIMViews.apply(pred, target, float(ssim_lambda), bool(valid_window))


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-15):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        lib = L.lib()
        st = L.stream_ptr()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise L.GaussCtrlHipError("FusedAdam needs contiguous float32 GPU parameters")
                state = self.state[p]
                if not state:
                    state["step"] = 0
                    state["exp_avg"] = torch.zeros_like(p)
                    state["exp_avg_sq"] = torch.zeros_like(p)
                state["step"] += 1
                g = p.grad.contiguous()
                L.check(lib.gc_adam_step(L.ptr(p), L.ptr(g), L.ptr(state["exp_avg"]), L.ptr(state["exp_avg_sq"]), L.i64(p.numel()),
                                         L.f32(group["lr"]), L.f32(b1), L.f32(b2), L.f32(group["eps"]), L.i32(state["step"]), st),
                        "gc_adam_step")
