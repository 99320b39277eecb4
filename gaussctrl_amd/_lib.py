"""ctypes loader of libgaussctrl_hip.so (the C-ABI of include/gaussctrl_hip.h).

The product path has NO fallback: if the shared library is missing or a symbol cannot be resolved
this module raises.  (The CPU oracle under oracle/ is test infrastructure and is never imported
from here.)
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GC_HIP_LIB") or os.path.join(_HERE, "libgaussctrl_hip.so")   # GC_HIP_LIB: kernel-experiment builds only

# every symbol include/gaussctrl_hip.h declares (tests/test_abi.py checks the header against this list)
SYMBOLS = [
    "gc_last_error_string", "gc_abi_version",
    "gc_project_gaussians_fwd", "gc_project_gaussians_bwd", "gc_sh_fwd", "gc_sh_bwd",
    "gc_raster_scan_workspace_bytes", "gc_raster_scan_tiles", "gc_raster_read_count",
    "gc_raster_map_intersects", "gc_raster_pad_intersects", "gc_raster_sort_workspace_bytes",
    "gc_raster_sort_intersects", "gc_raster_tile_bins",
    "gc_raster_depth_order_workspace_bytes", "gc_raster_depth_order", "gc_raster_bin_workspace_bytes", "gc_raster_bin_tiles", "gc_raster_bin_tiles_dev", "gc_raster_bin_tiles_boxes", "gc_rasterize_fwd", "gc_rasterize_bwd", "gc_rasterize_bwd_clamped",
    "gc_project_sh_fwd", "gc_project_sh_fwd_boxes", "gc_project_sh_bwd", "gc_project_sh_bwd_accumulate", "gc_raster_finalize", "gc_raster_finalize_into",
    "gc_l1_ssim_workspace_bytes", "gc_l1_ssim_fwd_bwd", "gc_adam_step",
    # batched views (round 5)
    "gc_project_sh_fwd_views", "gc_project_sh_bwd_views", "gc_raster_scan_tiles_views", "gc_raster_depth_order_views_workspace_bytes",
    "gc_raster_depth_order_views", "gc_raster_bin_views_workspace_bytes", "gc_raster_bin_tiles_views", "gc_rasterize_fwd_views",
    "gc_rasterize_bwd_views", "gc_l1_ssim_views_workspace_bytes", "gc_l1_ssim_fwd_bwd_views",
    # gather-free depth order (round 6)
    "gc_raster_order_boxes_views_workspace_bytes", "gc_raster_order_boxes_views", "gc_raster_bin_sorted_views",
    "gc_dn_gemm", "gc_dn_gemm_workspace_bytes", "gc_dn_gemm_row_stat_slots", "gc_dn_gemm_chan_parts_layout", "gc_dn_groupnorm_apply_parts", "gc_dn_groupnorm_apply_parts_fp8", "gc_dn_groupnorm_coef_parts", "gc_dn_concat_parts_layout", "gc_dn_concat_add_parts", "gc_dn_attention", "gc_dn_groupnorm", "gc_dn_groupnorm_workspace_bytes", "gc_dn_groupnorm_apply", "gc_dn_groupnorm_apply_fp8", "gc_dn_group_stats", "gc_dn_layernorm", "gc_dn_layernorm_fp8", "gc_dn_concat_add", "gc_dn_axpby",
    "gc_dn_cast_f32", "gc_dn_softmax_rows", "gc_dn_attention_workspace_bytes", "gc_dn_transformer_tail", "gc_dn_transformer_tail_layout", "gc_dn_transformer_head", "gc_dn_groupnorm_coef", "gc_dn_cfg_ddim_step", "gc_dn_depth_to_disparity", "gc_dn_mask_composite",
]

_lib = None


class GaussCtrlHipError(RuntimeError):
    pass


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GaussCtrlHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path.")
        l = C.CDLL(LIB_PATH)
        for s in SYMBOLS:
            if not hasattr(l, s):
                raise GaussCtrlHipError(f"libgaussctrl_hip.so does not export {s}")
        l.gc_last_error_string.restype = C.c_char_p
        for s in SYMBOLS:
            if s.endswith("_bytes"):
                getattr(l, s).restype = C.c_size_t
                continue
            if s not in ("gc_last_error_string",):
                getattr(l, s).restype = C.c_int
        _lib = l
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().gc_last_error_string().decode("utf-8", "replace")
        raise GaussCtrlHipError(f"{what} failed (code {rc}): {msg}")


def ptr(t):
    """device pointer of a torch tensor (or NULL for None) as c_void_p."""
    if t is None:
        return C.c_void_p(0)
    return C.c_void_p(t.data_ptr())


def host_floats(values):
    """small host float array (camera matrices) -> ctypes float array."""
    vals = [float(v) for v in values]
    return (C.c_float * len(vals))(*vals)


def stream_ptr():
    import torch
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)       # ~0.3 us instead of ~8 us through torch.cuda.current_stream()
    if raw is not None:
        return C.c_void_p(raw(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


i64 = C.c_int64
i32 = C.c_int
f32 = C.c_float
