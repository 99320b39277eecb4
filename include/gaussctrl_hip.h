/*
 * gaussctrl_hip.h -- C ABI of libgaussctrl_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the hot path of GaussCtrl's multi-view edit loop.  Every entry point takes
 * plain device pointers + sizes + a hipStream_t (as void*); no torch types cross this boundary.
 *
 * Conventions (all entry points):
 *   - returns 0 on success, a negative GC_E* code otherwise; never throws, never exits; text of
 *     the last error of the calling thread: gc_last_error_string().
 *   - the caller owns every buffer (including workspaces); the library allocates nothing, frees
 *     nothing and keeps no pointer after return.  Launches go only to `stream`; no hidden
 *     device-wide synchronisation (gc_raster_read_count is the single documented blocking call).
 *   - re-entrant: all state is in the arguments (kernel-selection overrides are descriptor fields, not environment
 *     variables; the only process-wide state is a per-(kernel, device) "attribute already set" bit, updated atomically).
 *   - all float tensors are contiguous float32 unless a name says bf16.
 *
 * Each block cites the reference interface it replaces (paths under /root/reference/).
 */
#ifndef GAUSSCTRL_HIP_H
#define GAUSSCTRL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GC_OK 0
#define GC_EINVAL (-1)   /* bad argument */
#define GC_ELAUNCH (-2)  /* HIP launch / runtime error */
#define GC_ENOSPC (-3)   /* workspace or capacity too small */

const char *gc_last_error_string(void);
int gc_abi_version(void);

/* ===================================================================================== */
/* Part A -- 3D-Gaussian rasterizer (replaces gsplat 0.1.3 as called by gaussctrl/gc_model.py) */
/* ===================================================================================== */

/* gsplat.project_gaussians forward -- call site gaussctrl/gc_model.py:140-154.
 * means3d[N,3] scales[N,3] quats[N,4](wxyz); viewmat[>=12] (row-major 3x4), projmat[16] (row-major,
 * full projection P*V) and cam_origin[3] are HOST pointers (28+3 floats travel as kernel arguments) -> cov3d[N,6] xys[N,2] depths[N] radii[N]i32 conics[N,3] num_tiles_hit[N]i32.
 * Culled Gaussians get radii = num_tiles_hit = 0 and zeroed float outputs. */
int gc_project_gaussians_fwd(int64_t N, const float *means3d, const float *scales, float glob_scale,
                             const float *quats, const float *viewmat, const float *projmat,
                             float fx, float fy, float cx, float cy, int img_h, int img_w,
                             int tiles_x, int tiles_y, float clip_thresh,
                             float *cov3d, float *xys, float *depths, int32_t *radii, float *conics,
                             int32_t *num_tiles_hit, void *stream);

/* autograd backward of the above (fired by gaussctrl/gc_trainer.py:275).
 * v_xy[N,2] v_depth[N](may be NULL) v_conic[N,3] -> v_mean3d[N,3] v_scale[N,3] v_quat[N,4]. */
int gc_project_gaussians_bwd(int64_t N, const float *means3d, const float *scales, float glob_scale,
                             const float *quats, const float *viewmat, const float *projmat,
                             float fx, float fy, float cx, float cy, int img_h, int img_w,
                             const int32_t *radii, const float *conics,
                             const float *v_xy, const float *v_depth, const float *v_conic,
                             float *v_mean3d, float *v_scale, float *v_quat, void *stream);

/* gsplat.sh.spherical_harmonics forward/backward -- call site gaussctrl/gc_model.py:166.
 * coeffs[N,K,3], K=(degree+1)^2 stored bases, only the first (degrees_to_use+1)^2 are used. */
int gc_sh_fwd(int64_t N, int degree, int degrees_to_use, const float *viewdirs, const float *coeffs,
              float *colors, void *stream);
int gc_sh_bwd(int64_t N, int degree, int degrees_to_use, const float *viewdirs, const float *v_colors,
              float *v_coeffs, void *stream);

/* Tile binning (inside gsplat.rasterize_gaussians: cumsum -> map_gaussian_to_intersects -> sort ->
 * get_tile_bin_edges; call sites gaussctrl/gc_model.py:174-186,191-202).
 *
 * gc_raster_scan_tiles: cum_tiles_hit[N] = inclusive scan of num_tiles_hit; *count_dev (device
 * int32[1]) = M = total intersections.  workspace >= gc_raster_scan_workspace_bytes(N). */
size_t gc_raster_scan_workspace_bytes(int64_t N);
int gc_raster_scan_tiles(int64_t N, const int32_t *num_tiles_hit, int32_t *cum_tiles_hit,
                         int32_t *count_dev, void *workspace, size_t workspace_bytes, void *stream);
/* blocking 4-byte read-back of a device counter on `stream` (the one host sync the reference also
 * has: `.item()` on the cumsum total). */
int gc_raster_read_count(const int32_t *count_dev, int32_t *count_host, void *stream);

/* isect_ids[M_cap] (int64: tile_id<<32 | depth bits), gaussian_ids[M_cap]; entries in [M, M_cap)
 * are padded with key = tiles<<32 (sorts last).  Writes past M_cap are dropped (GC_ENOSPC is not
 * detectable without a sync; compare *count_dev with M_cap after the frame). */
int gc_raster_map_intersects(int64_t N, int64_t M_cap, const float *xys, const float *depths,
                             const int32_t *radii, const int32_t *cum_tiles_hit, int tiles_x, int tiles_y,
                             int64_t *isect_ids, int32_t *gaussian_ids, void *stream);

/* sync-free variant: fill [*count_dev, M_cap) with the padding key tiles<<32 so a fixed-size sort
 * over M_cap entries leaves the real intersections first. */
int gc_raster_pad_intersects(int64_t M_cap, const int32_t *count_dev, int num_tiles, int64_t *isect_ids,
                             int32_t *gaussian_ids, void *stream);

/* stable sort of (isect_ids, gaussian_ids) by key over M entries; ties keep emission order
 * (= ascending Gaussian id).  workspace >= gc_raster_sort_workspace_bytes(M, tiles). */
size_t gc_raster_sort_workspace_bytes(int64_t M, int num_tiles);
int gc_raster_sort_intersects(int64_t M, int num_tiles, const int64_t *isect_ids, const int32_t *gaussian_ids,
                              int64_t *isect_ids_sorted, int32_t *gaussian_ids_sorted,
                              void *workspace, size_t workspace_bytes, void *stream);

/* tile_bins[T,2] = [start,end) of each tile in the sorted list (zeros for empty tiles). */
int gc_raster_tile_bins(int64_t M, int num_tiles, const int64_t *isect_ids_sorted, int32_t *tile_bins,
                        void *stream);

/* Two-level binning (the product path; same outputs as scan+map+sort+tile_bins above, ~4x fewer bytes moved).
 * Phase 1, no host sync: depth_order[N] = Gaussian ids sorted by (depth bits, id) with culled ones last;
 * cum_sorted[N] = inclusive scan of num_tiles_hit in that order; *count_dev = M (read it with gc_raster_read_count). */
size_t gc_raster_depth_order_workspace_bytes(int64_t N);
int gc_raster_depth_order(int64_t N, const float *depths, const int32_t *radii, const int32_t *num_tiles_hit,
                          int32_t *depth_order, int32_t *cum_sorted, int32_t *count_dev,
                          void *workspace, size_t workspace_bytes, void *stream);
/* Phase 2: emit (tile, id) in depth order, stable radix passes over the tile id only, tile bins.
 * gaussian_ids_sorted[M], tile_bins[T,2]; isect_ids_sorted[M] (tile<<32 | depth bits) optional (NULL skips it).
 * Replaces gsplat's map_gaussian_to_intersects + 64-bit key sort + get_tile_bin_edges. */
size_t gc_raster_bin_workspace_bytes(int64_t M);
int gc_raster_bin_tiles(int64_t N, int64_t M, const int32_t *depth_order, const int32_t *cum_sorted,
                        const float *xys, const float *depths, const int32_t *radii, int tiles_x, int tiles_y,
                        int32_t *gaussian_ids_sorted, int32_t *tile_bins, int64_t *isect_ids_sorted,
                        void *workspace, size_t workspace_bytes, void *stream);

/* Sync-free phase 2 (the "device-side counter + capacity check" convention of the boundary): the intersection count is
 * read on the device from count_dev (as written by gc_raster_depth_order); buffers are sized for the caller's capacity
 * M_cap; *overflow_dev = 1 when the frame needed more than M_cap intersections (re-run with a larger capacity).
 * No host round trip anywhere in the frame. */
int gc_raster_bin_tiles_dev(int64_t N, int64_t M_cap, const int32_t *count_dev, int32_t *overflow_dev,
                            const int32_t *depth_order, const int32_t *cum_sorted, const float *xys, const float *depths,
                            const int32_t *radii, int tiles_x, int tiles_y, int32_t *gaussian_ids_sorted, int32_t *tile_bins,
                            int64_t *isect_ids_sorted, void *workspace, size_t workspace_bytes, void *stream);
/* Phase 2 on the TIGHT tile boxes gc_project_sh_fwd_boxes wrote (packed min_x | max_x << 8 | min_y << 16 | max_y << 24, exclusive maxima,
 * 0 = no tile) instead of the boxes recomputed from xys / radii: fewer (tile, Gaussian) pairs, bit-identical images.  count_dev NULL:
 * M is the exact count (as gc_raster_bin_tiles); else the sync-free form (M = capacity, overflow_dev required). */
int gc_raster_bin_tiles_boxes(int64_t N, int64_t M, const int32_t *count_dev, int32_t *overflow_dev,
                              const int32_t *depth_order, const int32_t *cum_sorted, const uint32_t *tile_boxes,
                              const float *depths, int tiles_x, int tiles_y, int32_t *gaussian_ids_sorted, int32_t *tile_bins,
                              int64_t *isect_ids_sorted, void *workspace, size_t workspace_bytes, void *stream);


/* gsplat.rasterize_gaussians forward: RGB (+ optional extra channel, used for the reference's
 * second "depth" pass gc_model.py:191-202, composited in the same sweep) + final_Ts + final_index.
 * colors[N,3], opacities[N], extra[N] or NULL, background[3] (device) ->
 * out_img[H,W,3], out_extra[H,W] or NULL, final_Ts[H,W], final_index[H,W]. */
int gc_rasterize_fwd(int img_h, int img_w, int tiles_x, int tiles_y,
                     const int32_t *gaussian_ids_sorted, const int32_t *tile_bins,
                     const float *xys, const float *conics, const float *colors, const float *opacities,
                     const float *extra, const float *background,
                     float *out_img, float *out_extra, float *final_Ts, int32_t *final_index, void *stream);

/* backward of the compositing.  v_out[H,W,3], v_out_alpha[H,W] or NULL.  The four gradient
 * buffers must be zero-filled by the caller (results are accumulated atomically). */
int gc_rasterize_bwd(int img_h, int img_w, int tiles_x, int tiles_y, int64_t N,
                     const int32_t *gaussian_ids_sorted, const int32_t *tile_bins,
                     const float *xys, const float *conics, const float *colors, const float *opacities,
                     const float *background, const float *final_Ts, const int32_t *final_index,
                     const float *v_out, const float *v_out_alpha,
                     float *v_xy, float *v_conic, float *v_colors, float *v_opacity, void *stream);
/* The same with the backward of get_outputs' `rgb = clamp(rgb, max=1)` (gaussctrl/gc_model.py:188) folded into the pixel load:
 * v_out is taken as 0 where pre_clamp[H,W,3] (the un-clamped composited image) exceeds 1. */
int gc_rasterize_bwd_clamped(int img_h, int img_w, int tiles_x, int tiles_y, int64_t N,
                             const int32_t *gaussian_ids_sorted, const int32_t *tile_bins,
                             const float *xys, const float *conics, const float *colors, const float *opacities,
                             const float *background, const float *final_Ts, const int32_t *final_index,
                             const float *v_out, const float *v_out_alpha, const float *pre_clamp,
                             float *v_xy, float *v_conic, float *v_colors, float *v_opacity, void *stream);

/* Fused per-Gaussian front end of GaussCtrlModel.get_outputs (gaussctrl/gc_model.py:138-169,181):
 * exp(scales), quat normalisation, projection, view directions, SH(+0.5, clamp min 0), sigmoid(opacity)
 * in ONE pass over the 59-float parameter record.
 * means[N,3] log_scales[N,3] quats[N,4] opacity_logits[N] features_dc[N,3] features_rest[N,K-1,3]
 * cam_origin[3] (host floats) -> xys depths radii conics num_tiles_hit rgbs[N,3] opac[N].
 * degrees_to_use = -1 selects the config.sh_degree == 0 colour of gc_model.py:169: rgbs = sigmoid(features_dc). */
int gc_project_sh_fwd(int64_t N, const float *means, const float *log_scales, const float *quats,
                      const float *opacity_logits, const float *features_dc, const float *features_rest,
                      int sh_degree, int degrees_to_use, const float *viewmat, const float *projmat,
                      const float *cam_origin, float fx, float fy, float cx, float cy, int img_h, int img_w,
                      int tiles_x, int tiles_y, float clip_thresh,
                      float *xys, float *depths, int32_t *radii, float *conics, int32_t *num_tiles_hit,
                      float *rgbs, float *opac, void *stream);
/* The same with tight tile boxes (fused product path): a pixel can only pass alpha >= 1/255 inside the ellipse sigma <= ln(255 opacity);
 * its bounding box, intersected with gsplat's 3-sigma box, is what tile_boxes[N] / num_tiles_hit describe (31 % fewer pairs on the
 * synthetic scenes, bit-identical images and gradients; the gsplat-shaped gc_project_gaussians_fwd keeps the reference's box).
 * tiles_x, tiles_y <= 255. */
int gc_project_sh_fwd_boxes(int64_t N, const float *means, const float *log_scales, const float *quats,
                            const float *opacity_logits, const float *features_dc, const float *features_rest,
                            int sh_degree, int degrees_to_use, const float *viewmat, const float *projmat,
                            const float *cam_origin, float fx, float fy, float cx, float cy, int img_h, int img_w,
                            int tiles_x, int tiles_y, float clip_thresh, float *xys, float *depths, int32_t *radii,
                            float *conics, int32_t *num_tiles_hit, float *rgbs, float *opac, uint32_t *tile_boxes, void *stream);


/* Fused backward: v_xy,v_conic,v_rgbs,v_opac -> gradients of the six leaf tensors.  rgbs[N,3] = the colours gc_project_sh_fwd
 * produced (the clamp mask is rgbs > 0, the sigmoid mode's derivative rgbs (1 - rgbs): the SH record is not read again). */
int gc_project_sh_bwd(int64_t N, const float *means, const float *log_scales, const float *quats,
                      const float *opacity_logits, const float *rgbs,
                      int sh_degree, int degrees_to_use, const float *viewmat, const float *projmat,
                      const float *cam_origin, float fx, float fy, float cx, float cy, int img_h, int img_w,
                      const int32_t *radii, const float *conics,
                      const float *v_xy, const float *v_conic, const float *v_rgbs, const float *v_opac,
                      float *v_means, float *v_log_scales, float *v_quats, float *v_opacity_logits,
                      float *v_features_dc, float *v_features_rest, void *stream);
/* The same with the six outputs ACCUMULATED into (+=): gradient accumulation over the views of a batch inside the kernel (what
 * autograd otherwise does with one read-add-write pass per tensor and view). */
int gc_project_sh_bwd_accumulate(int64_t N, const float *means, const float *log_scales, const float *quats,
                      const float *opacity_logits, const float *rgbs,
                      int sh_degree, int degrees_to_use, const float *viewmat, const float *projmat,
                      const float *cam_origin, float fx, float fy, float cx, float cy, int img_h, int img_w,
                      const int32_t *radii, const float *conics,
                      const float *v_xy, const float *v_conic, const float *v_rgbs, const float *v_opac,
                      float *v_means, float *v_log_scales, float *v_quats, float *v_opacity_logits,
                      float *v_features_dc, float *v_features_rest, void *stream);

/* Epilogue of get_outputs (gc_model.py:188,197-204): rgb=min(rgb,1); alpha=1-final_T;
 * depth = alpha>0 ? depth/alpha : 1000.  In place on out_img / out_extra; writes alpha[H,W]. */
int gc_raster_finalize(int64_t num_pixels, float *out_img, float *out_extra, const float *final_Ts,
                       float *alpha, void *stream);
/* Out-of-place form for a differentiable render: img_raw (as gc_rasterize_fwd wrote it) is kept for gc_rasterize_bwd_clamped,
 * the clamped image goes to img_clamped (same epilogue otherwise; reference: gaussctrl/gc_model.py:188,197-204). */
int gc_raster_finalize_into(int64_t num_pixels, const float *img_raw, float *img_clamped, float *out_extra,
                            const float *final_Ts, float *alpha, void *stream);

/* Loss + optimiser of the splat optimisation that follows the edit (reference:
 * SplatfactoModel.get_loss_dict inherited via gaussctrl/gc_pipeline.py:284-285, Adam groups gaussctrl/gc_config.py:58-87,
 * iteration gaussctrl/gc_trainer.py:257-301). */
size_t gc_l1_ssim_workspace_bytes(int H, int W, int C);
/* (1-lambda)*mean|pred-target| + lambda*(1 - mean SSIM_11x11(pred,target)) on float32 [H,W,C]: loss_sums (device float[2]) =
 * {sum of the SSIM map, sum |pred-target|}; v_pred = grad_scale * d loss / d pred.
 * valid_window = 1: unpadded windows, SSIM mean over (H-10)(W-10)C (pytorch_msssim.SSIM as splatfacto calls it); 0: zero-padded
 * windows, mean over HWC. */
int gc_l1_ssim_fwd_bwd(const float *pred, const float *target, int H, int W, int C, float lambda_, float grad_scale,
                       int valid_window, float *loss_sums, float *v_pred, void *workspace, size_t workspace_bytes, void *stream);
/* torch.optim.Adam step (no weight decay / amsgrad) on one flat float32 tensor; step is the 1-based iteration count. */
int gc_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr, float beta1,
                 float beta2, float eps, int step, void *stream);

/* ------------------------------------------------------------------------------------------------------------------------------
 * Batched views (round 5): C cameras of ONE scene per set of launches.  The reference renders its views one at a time
 * (gaussctrl/gc_pipeline.py:124-130 render_reverse, gaussctrl/gc_trainer.py:186-201 one view per training step); every stage below is
 * the single-view entry point above with a leading view dimension -- grid.y / grid.z = view for the sort and compositing kernels, a loop
 * over the views INSIDE the per-Gaussian kernels so that the 236-byte parameter record is read once per batch, not once per view.
 * Layout: per-view arrays are [C][N][..] / [C][M_cap] / [C][T][2] / [C][H][W][..]; all views share H, W and the tile grid; always the
 * sync-free form (count_dev[C] / overflow_dev[C] stay on the device, M_cap = per-view capacity).  Results per view are bit-identical to
 * the single-view entry points (same device code), gradients sum over the views in view order.
 * ------------------------------------------------------------------------------------------------------------------------------ */
#define GC_VIEW_CAM_FLOATS 35   /* per view, HOST floats: viewmat[12] | projmat[16] | cam_origin[3] | fx fy cx cy */

/* gc_project_sh_fwd[_boxes] for C views (launches of <= 8 views; the cameras travel as kernel arguments).  opac[N] is written once (it does
 * not depend on the camera).  tile_boxes [C][N] (NULL: gsplat's boxes in num_tiles_hit); depth_pairs [C][N][2] u32 or NULL = (depth bits, or
 * 0xFFFFFFFF when culled, Gaussian id): the input pairs of gc_raster_depth_order_views, saving its key-building pass. */
int gc_project_sh_fwd_views(int64_t N, int C, const float *means, const float *log_scales, const float *quats,
                            const float *opacity_logits, const float *features_dc, const float *features_rest,
                            int sh_degree, int degrees_to_use, const float *cams, int img_h, int img_w,
                            int tiles_x, int tiles_y, float clip_thresh, float *xys, float *depths, int32_t *radii,
                            float *conics, int32_t *num_tiles_hit, float *rgbs, float *opac, uint32_t *tile_boxes,
                            uint32_t *depth_pairs, void *stream);
/* gc_project_sh_bwd for C views: the six leaf gradients are the sum over the views (view order), written (accumulate = 0) or added to the
 * buffers' contents (accumulate = 1) once per batch. */
int gc_project_sh_bwd_views(int64_t N, int C, int accumulate, const float *means, const float *log_scales, const float *quats,
                            const float *opacity_logits, const float *rgbs, int sh_degree, int degrees_to_use, const float *cams,
                            int img_h, int img_w, const int32_t *radii, const float *conics, const float *v_xy, const float *v_conic,
                            const float *v_rgbs, const float *v_opac, float *v_means, float *v_log_scales, float *v_quats,
                            float *v_opacity_logits, float *v_features_dc, float *v_features_rest, void *stream);
/* gc_raster_scan_tiles for C views, the input optionally gathered through `order` first (in[order[i]]; NULL: plain).  The scan scratch of
 * view c starts ws_view_stride bytes after that of view c - 1 (>= gc_raster_scan_workspace_bytes(N), multiple of 4; ignored for C = 1). */
int gc_raster_scan_tiles_views(int64_t N, int C, const int32_t *num_tiles_hit, const int32_t *order, int32_t *cum_tiles_hit,
                               int32_t *count_dev, void *workspace, size_t workspace_bytes, int64_t ws_view_stride, void *stream);
size_t gc_raster_depth_order_views_workspace_bytes(int64_t N, int C);
int gc_raster_depth_order_views(int64_t N, int C, const float *depths, const int32_t *radii, const uint32_t *depth_pairs,
                                const int32_t *num_tiles_hit, int32_t *depth_order, int32_t *cum_sorted, int32_t *count_dev,
                                void *workspace, size_t workspace_bytes, void *stream);
size_t gc_raster_bin_views_workspace_bytes(int64_t M_cap, int C);
int gc_raster_bin_tiles_views(int64_t N, int C, int64_t M_cap, const int32_t *count_dev, int32_t *overflow_dev,
                              const int32_t *depth_order, const int32_t *cum_sorted, const uint32_t *tile_boxes, const float *depths,
                              int tiles_x, int tiles_y, int32_t *gaussian_ids_sorted, int32_t *tile_bins, int64_t *isect_ids_sorted,
                              void *workspace, size_t workspace_bytes, void *stream);
/* Round 6: the same two phases WITHOUT the per-Gaussian gathers (num_tiles_hit[order[j]] in the scan, tile_boxes[order[j]] in the emission: two
 * 128-byte line fetches per Gaussian, 2/3 of the bytes the depth order fetched).  The packed tight box rides through the radix passes as a third
 * word of the item, culled Gaussians are dropped by the first pass; the scan and the emission read sorted arrays sequentially.
 * gc_raster_order_boxes_views: depth_pairs [C][N][2] + tile_boxes [C][N] (gc_project_sh_fwd_views) -> depth_order [C][N] = ids of the VISIBLE
 * Gaussians in (depth bits, id) order (entries past visible_dev[c] unspecified), boxes_sorted [C][N], cum_sorted [C][N], count_dev [C] = M,
 * visible_dev [C].  gc_raster_bin_sorted_views: phase 2 on those (workspace: gc_raster_bin_views_workspace_bytes).  gaussian_ids_sorted and
 * tile_bins are bit-identical to gc_raster_depth_order_views + gc_raster_bin_tiles_views. */
size_t gc_raster_order_boxes_views_workspace_bytes(int64_t N, int C);
int gc_raster_order_boxes_views(int64_t N, int C, const uint32_t *depth_pairs, const uint32_t *tile_boxes, int32_t *depth_order,
                                uint32_t *boxes_sorted, int32_t *cum_sorted, int32_t *count_dev, int32_t *visible_dev, void *workspace,
                                size_t workspace_bytes, void *stream);
int gc_raster_bin_sorted_views(int64_t N, int C, int64_t M_cap, const int32_t *count_dev, int32_t *overflow_dev, const int32_t *visible_dev,
                               const int32_t *depth_order, const uint32_t *boxes_sorted, const int32_t *cum_sorted, int tiles_x, int tiles_y,
                               int32_t *gaussian_ids_sorted, int32_t *tile_bins, void *workspace, size_t workspace_bytes, void *stream);
/* compositing of C views in one launch.  opacities [N] (shared_opacities = 1) or [C][N]; background [3] (shared_background = 1) or [C][3]. */
int gc_rasterize_fwd_views(int C, int64_t N, int64_t M_cap, int shared_opacities, int shared_background, int img_h, int img_w,
                           int tiles_x, int tiles_y, const int32_t *gaussian_ids_sorted, const int32_t *tile_bins, const float *xys,
                           const float *conics, const float *colors, const float *opacities, const float *extra,
                           const float *background, float *out_img, float *out_extra, float *final_Ts, int32_t *final_index,
                           void *stream);
/* v_xy [C][N][2] v_conic [C][N][3] v_colors [C][N][3] v_opacity [C][N] must be ZERO on entry; pre_clamp [C][H][W][3] or NULL. */
int gc_rasterize_bwd_views(int C, int64_t N, int64_t M_cap, int shared_opacities, int shared_background, int img_h, int img_w,
                           int tiles_x, int tiles_y, const int32_t *gaussian_ids_sorted, const int32_t *tile_bins, const float *xys,
                           const float *conics, const float *colors, const float *opacities, const float *background,
                           const float *final_Ts, const int32_t *final_index, const float *v_out, const float *v_out_alpha,
                           const float *pre_clamp, float *v_xy, float *v_conic, float *v_colors, float *v_opacity, void *stream);
/* gc_l1_ssim_fwd_bwd for B image pairs [B][H][W][C]; loss_out [B][2] (every view's own sums). */
size_t gc_l1_ssim_views_workspace_bytes(int B, int H, int W, int C);
int gc_l1_ssim_fwd_bwd_views(int B, const float *pred, const float *target, int H, int W, int C, float lambda_, float grad_scale,
                             int valid_window, float *loss_out, float *v_pred, void *workspace, size_t workspace_bytes, void *stream);


/* ===================================================================================== */
/* Part B -- ControlNet + UNet denoise step (replaces the diffusers / cuBLAS / cuDNN calls behind  */
/* gaussctrl/gc_pipeline.py:142-145,209-219 and the attention processor gaussctrl/utils.py:25-133) */
/* ===================================================================================== */
/* Activations are channels-last ("tokens x channels") 2-byte tensors; dtype 0 = bf16, 1 = f16;    */
/* accumulation and statistics are fp32.                                                          */

/* GEMM / implicit-GEMM 3x3 convolution with fused epilogue (torch.nn.Linear / Conv2d of the UNet,  */
/* ControlNet and VAE).  out[m][n] = (act(acc + bias[n] + rowvec[m / rows_per_batch][n])) * out_scale */
/*                                   + residual[m][n];  acc = sum_k Act[m][k] * W[n][k].            */
typedef struct gc_gemm_desc {
    int dtype;                 /* 0 bf16, 1 f16 */
    int mode;                  /* 0: Act = A[M][lda] (Linear, 1x1 conv);  1: 3x3 conv, pad 1, on NHWC A[B][Hi][Wi][Cin] */
    int64_t M, N, K;           /* mode 1: M = B*Ho*Wo, K = 9*Cin ordered (tap, cin) */
    const void *A;
    int64_t lda;
    int B, Hi, Wi, Cin, Ho, Wo, stride, upsample;   /* upsample=1: nearest x2 of the input fused into the loader */
    int pad_lo;                /* top/left zero padding: 1 (Conv2d padding=1) or 0 (VAE encoder Downsample2D: pad (0,1,0,1), stride 2) */
    const void *W;             /* [N][K] */
    const float *bias;         /* [N] or NULL */
    const float *rowvec;       /* [M / rows_per_batch][ld_rowvec] or NULL (time-embedding add of ResnetBlock2D) */
    int64_t ld_rowvec;
    int64_t rows_per_batch;
    const void *residual;      /* [M][ldr] or NULL */
    int64_t ldr;
    float out_scale;
    int act;                   /* 0 none, 1 SiLU, 2 image post-process clamp(x/2+0.5, 0, 1) */
    int geglu;                 /* 1: W rows permuted in 16-blocks [x|gate]; out[m][n/2] = x * gelu(gate), ldc = N/2 */
    void *out;                 /* [M][ldc] (NULL to skip) */
    int64_t ldc;
    int out_f32;               /* 1: `out` is float32 */
    void *out_t;               /* optional transposed copy [M / rows_per_batch][N][ldt] (V operand of gc_dn_attention) */
    int64_t ldt;
    int64_t t_batch_stride;
    int64_t t_col0;            /* fused QKV projection: output columns >= t_col0 are written ONLY to out_t (column n - t_col0); 0 = all columns to both */
    void *workspace;           /* >= gc_dn_gemm_workspace_bytes(desc) bytes (split-K fp32 accumulator for small-M problems); */
    size_t workspace_bytes;    /* NULL / too small: the problem runs unsplit */
    const void *zeros;         /* >= 16 bytes of device zeros: enables the LDS-DMA kernel (padding / out-of-range lanes fetch from it); NULL: register-staged kernel */
    /* LayerNorm folded into the GEMM (mode 0): A holds the UN-normalised rows x, W = W0 * diag(gamma), bias = b0 + W0 beta and */
    /* out = rstd[m] * (acc - mean[m] * ln_colsum[n]) + bias[n], (mean, rstd) from sum_slots ln_row_stats[slot][m] = (sum_k x, sum_k x^2). */
    const float *ln_row_stats; /* [ln_row_stat_slots][M][2] or NULL: what the GEMM that produced A left in its out_row_stats */
    int ln_row_stat_slots;     /* gc_dn_gemm_row_stat_slots(producer's descriptor) */
    const float *ln_colsum;    /* [N] = sum_k W[n][k] (of the rounded weights) */
    float ln_eps;
    /* statistics of the STORED output (NULL: not wanted): */
    float *out_row_stats;      /* [gc_dn_gemm_row_stat_slots(desc)][M][2]: per row, (sum, sum of squares) over one column slab per slot; */
                               /* plain stores (no atomics, no zero-init needed) -> ln_row_stats of a following GEMM */
    float *out_group_stats;    /* [M / rows_per_batch][gn_groups][2]: per (batch, GroupNorm group of N / gn_groups channels) sums, ADDED with */
    int gn_groups;             /* float atomics (caller zero-fills) -> gc_dn_groupnorm_apply; needs rows_per_batch % 16 == 0, gn_groups <= 32 */
    /* fp8 path (BASELINE configs[3]): A and W hold OCP e4m3 bytes (K, lda, Cin count fp8 elements; K % 128 == 0, conv: Cin % 128 == 0), */
    /* multiplied on the block-scaled MFMA; real value = stored * 2^(scale byte - 127) (E8M0): one byte per weight row, one for all of A. */
    /* Linears take the GEGLU / transposed-V epilogues of the 2-byte path; long-K problems on part-filled grids (16 x 16-map convs) run in  */
    /* k-slices when `workspace` holds gc_dn_gemm_workspace_bytes(desc) bytes; out_chan_parts: fast convs and the k-sliced problems.         */
    int fp8;
    const void *w_scale;       /* [N] E8M0 bytes */
    int a_scale;               /* E8M0 byte of the activation tensor */
    int kernel_variant;        /* 0 = automatic.  Overrides for tests / experiments: bits 0-2 force the 8-wave kernel's m-tiles per wave (2,3,4); */
                               /* 0x10 4-wave kernel only; 0x20 force the 8-wave kernel; 0x40 no k-slices for part-filled conv grids; 0x80 slice 8x8-map convs too. */
                               /* fp8 problems (k_gemm8q) honour bits 0-2 (a forced tile height also disables their k-slices) and 0x40 */
                               /* 0x1000: fast 3x3 convs walk K tap-outer (rounds 1-5) instead of tap-inner (round 6 default: same products, other summation order) */
    float *out_chan_parts;     /* NULL or [M / rows_per_batch][nslab][gn_groups][2][2]: partial (sum, sum of squares) of the stored output per row slab,  */
                               /* GroupNorm group (N / gn_groups channels) and half (1: rest of a group straddling two column tiles); layout:          */
                               /* gc_dn_gemm_chan_parts_layout.  PLAIN stores -- no atomics, no zero-init -> gc_dn_groupnorm_apply_parts /              */
                               /* gc_dn_groupnorm_coef_parts: the statistics pass of the GroupNorm that follows this conv / linear for free           */
    int64_t plan_rows;         /* 0 = plan for M.  > 0 (the rows ONE frame contributes: tokens, or Ho * Wo): kernel family and split-K are planned as if */
                               /* M were plan_rows, so every output row is accumulated in the same order whatever else shares the batch */
                               /* (batch-invariant results: a view's latents do not depend on its chunk-mates or on the rank count) */
    int out_fp8;               /* fp8 linears only: 0 = output in `dtype`; else an E8M0 byte: `out` receives e4m3 BYTES [M][ldc] of value * 2^(127 - out_fp8), */
                               /* saturated to +-448 -- the activation operand of a following fp8 GEMM with a_scale = out_fp8 (GEGLU hidden ->    */
                               /* FF down projection: the feed-forward of /root/reference/gaussctrl/utils.py:121-131's blocks stays in e4m3)       */
    /* Weight SETS and the softmax-heads epilogue (round 5: the text cross-attention of a transformer block folded into two GEMMs -- scores =
     * LN(h) (K_text Wq)^T -> softmax per head -> P (Wo V_text^T)^T + b + h; diffusers BasicTransformerBlock.attn2 behind
     * /root/reference/gaussctrl/gc_pipeline.py:209-219).  w_set_rows > 0: rows [s w_set_rows, (s + 1) w_set_rows) use weight matrix s = W + s w_set_stride
     * elements, bias + s N, ln_colsum + s N (one set per CFG half: the text differs); only on the lean LayerNorm-fold kernels (a K % 64 == 0 linear with
     * ln_row_stats or out_row_stats), w_set_rows a multiple of 64, 128, 192 or 256 (the library picks a row tile that divides it).
     * softmax_keys > 0 (with ln_row_stats; N % 80 == 0): every 80-column block of a row is one head's scores against <= 80 keys, of which the first
     * softmax_keys are real; `out` receives softmax over them (scores pre-scaled by log2(e) / sqrt(D)), zeros in the padding columns. */
    int64_t w_set_rows, w_set_stride;
    int softmax_keys;
} gc_gemm_desc;
size_t gc_dn_gemm_workspace_bytes(const gc_gemm_desc *desc);
int gc_dn_gemm_row_stat_slots(const gc_gemm_desc *desc);   /* column slabs per row this problem writes to out_row_stats (with desc->workspace set) */
/* Layout of out_chan_parts for this problem (call with desc->workspace and desc->gn_groups set, as for the launch): *rows_per_slab rows per
 * slab (slabs = the kernel's row tiles, counted over all M rows; a tile that straddles two batches contributes a slab to each), *nslab slab
 * slots per batch, *col_tile = the kernel's column tile (a group that straddles two column tiles has its rest in half 1).
 * *rows_per_slab = 0: the kernel this problem selects cannot produce them (GEGLU / transposed / fp32 / fp8 outputs, upsample-fused convs,
 * rows_per_batch < 256 or not a multiple of 32, 4-wave kernel) -- run the stand-alone GroupNorm instead. */
int gc_dn_gemm_chan_parts_layout(const gc_gemm_desc *desc, int64_t *rows_per_slab, int *nslab, int *col_tile);
int gc_dn_gemm(const gc_gemm_desc *desc, void *stream);

/* Fused multi-K/V-set attention = CrossViewAttnProcessor core, gaussctrl/utils.py:86-117 (+ compute_attn :25-37). */
/* O[b] = sum_s set_weight[s] * softmax(scale * Q[b] K[kv(b,s)]^T) V[kv(b,s)],                                   */
/* kv(b,s) = b if set_kind[s] == -1; b / frames_per_half if -2 (text K/V shared by a CFG half);                  */
/*           reference frame r = set_kind[s] >= 0 of b's half: row (b / frames_per_half) * ref_frames_per_half + r */
/*           of the reference bank Kref / Vtref (NULL: the bank is K / Vt itself, ref_frames_per_half =           */
/*           frames_per_half, i.e. the references are the first frames of each half as in gc_pipeline.py:206-207). */
typedef struct gc_attn_desc {
    int dtype;
    int batch, heads, head_dim;      /* head_dim in {8,16,32,40,64,80,160} */
    int Lq, Lk;
    int frames_per_half;             /* video_length = B // unet_chunk_size (utils.py:94) */
    int nsets;
    int set_kind[5];
    float set_weight[5];
    float scale;
    const void *Q; int64_t ldq, q_batch_stride;     /* [B][Lq][ldq], head h at column h*head_dim */
    const void *K; int64_t ldk, k_batch_stride;     /* [Bk][Lk][ldk] */
    const void *Vt; int64_t ldvt, vt_batch_stride;  /* [Bk][heads*head_dim][ldvt], token-contiguous, zero padded */
    void *O; int64_t ldo, o_batch_stride;
    const void *Kref; int64_t kref_batch_stride;    /* optional cached reference K / V^T (same ldk / ldvt) */
    const void *Vtref; int64_t vtref_batch_stride;
    int ref_frames_per_half;
    int q_prescaled;                 /* Q is already multiplied by scale*log2(e) (folded into the Q projection weights): `scale` is ignored */
    int kernel_variant;              /* 0 = automatic.  head_dim 40 then runs k_attn5 (key-split 8-wave kernel, csrc/dn_attn5.hip) when
                                        Lk % 64 == 0 and Lq % 256 == 0 and nsets * ceil(Lk / 64) >= 4, else k_attn4; head_dim 80 k_attn3; every
                                        other head_dim and short key streams the online-softmax kernel k_attn.  A/B and test overrides:
                                        bit 0: k_attn for every shape; bit 1: head_dim 40 on the 16x16x32 kernel (k_attn3); bit 2: k_attn4 with
                                        8 waves; bit 3: k_attn4 with 64 queries per wave (25 % slower: DESIGN.md 7.1); bit 4: k_attn4 instead of
                                        k_attn5; bits 5 / 6: k_attn5 with an LDS ring of 4 / 8 tiles instead of 6; bits 8..: timing ablations of
                                        instrumented builds (ignored by the product build) */
    void *workspace;                 /* optional, >= gc_dn_attention_workspace_bytes(desc) (0 unless head_dim == 160 and nsets > 1; ignored when 0): lets small grids with several K/V sets run */
    size_t workspace_bytes;          /* one workgroup per (query block, set) + a fixed-order fp32 combine; NULL: one launch as before */
} gc_attn_desc;
size_t gc_dn_attention_workspace_bytes(const gc_attn_desc *desc);
int gc_dn_attention(const gc_attn_desc *desc, void *stream);

/* The tail of a level-0 transformer block (C = 320, 8 heads) in ONE launch: attn1.to_out + residual, LayerNorm, attn2 (text cross-
 * attention: to_q, softmax over <= 96 text tokens, to_out + residual), LayerNorm, GEGLU feed-forward + residual, proj_out + the block's
 * input.  Replaces nine launches of the per-op path (diffusers BasicTransformerBlock.forward / Transformer2DModel.forward behind the
 * reference's gc_pipeline.py:224-227); rows stay in registers, the weights arrive as a host-prepared linear stream of 1 KB MFMA operand
 * blocks (gaussctrl_amd/sd/weights.py::tail_streams documents the layout; gc_dn_transformer_tail_layout returns its sizes). */
typedef struct gc_ttail_desc {
    int dtype;
    int channels, heads;             /* 320, 8 */
    int64_t M;                       /* token rows = frames * rows_per_frame */
    int64_t rows_per_frame;          /* % 128 == 0 */
    int frames_per_half;             /* frame b reads the text K / V^T of CFG half b / frames_per_half (<= 2 halves) */
    int text_len;                    /* valid text tokens (77), <= 96 */
    float ln_eps;
    const void *attn_out;            /* [M][320] output of the self-attention (before to_out) */
    const void *resid;               /* [M][320] residual stream entering the block (proj_in output) */
    const void *x_in;                /* [M][320] input of the Transformer2DModel (proj_out residual) */
    void *out;                       /* [M][320] */
    const void *w_a;                 /* stream segment A: attn1.to_out, attn2.to_q */
    const void *w_kv;                /* [halves][blocks_kv KB] text K / V^T per CFG half */
    const void *w_b;                 /* stream segment B: attn2.to_out, feed-forward, proj_out */
    const float *params;             /* biases / LayerNorm affine in lane order */
    int stop_after;                  /* 0; tests: 1..5 = `out` receives the intermediate after that stage */
    int resid_fragment_layout;       /* 1: `resid` was written by gc_dn_transformer_head with h_fragment_layout = 1 */
    int64_t in_rows;                 /* > 0: attn_out / resid / x_in hold rows [0, in_rows) only and output row m reads input row m % in_rows -- the
                                      * CFG-shared prefix: both CFG halves continue from ONE copy of the shared rows (in_rows % 128 == 0, M % in_rows == 0) */
} gc_ttail_desc;
int gc_dn_transformer_tail(const gc_ttail_desc *desc, void *stream);
void gc_dn_transformer_tail_layout(int64_t *blocks_a, int64_t *blocks_kv, int64_t *blocks_b, int64_t *param_floats);

/* The head of a level-0 transformer block (C = 320) in ONE launch: GroupNorm apply (coefficients from gc_dn_groupnorm_coef), proj_in,
 * LayerNorm1 and the attn1 Q / K / V projections (Transformer2DModel.forward + BasicTransformerBlock.forward up to the attention
 * processor call, behind the reference's gc_pipeline.py:224-227).  Rows stay in registers as in gc_dn_transformer_tail; the weights are one
 * operand stream (gaussctrl_amd/sd/weights.py::head_stream). */
typedef struct gc_thead_desc {
    int dtype;
    int channels;                    /* 320 */
    int64_t M;                       /* token rows = frames * rows_per_frame */
    int64_t rows_per_frame;          /* % 128 == 0 */
    float ln_eps;
    const void *x;                   /* [M][320] input of the Transformer2DModel (before its GroupNorm) */
    const float *gn_coef;            /* [frames][320][2] */
    void *h;                         /* [M][320] proj_in output (the block's residual stream) */
    void *qk;                        /* [M][640] = Q | K (Q scaled on the host as for gc_attn_desc.q_prescaled) */
    void *vt;                        /* [frames][320][ldvt] V transposed, token-contiguous */
    int64_t ldvt, vt_batch_stride;
    const void *w;                   /* operand stream: proj_in, to_q, to_k, to_v */
    const float *params;             /* proj_in bias, LayerNorm1 gamma / beta in lane order */
    int h_fragment_layout;           /* 1: h is written as MFMA fragments ([M / 32][20][64 lanes][8]), readable only by gc_dn_transformer_tail */
} gc_thead_desc;
int gc_dn_transformer_head(const gc_thead_desc *desc, void *stream);

/* GroupNorm(G groups, eps)(+SiLU) on [B][HW][C]; stats_ws: gc_dn_groupnorm_workspace_bytes(B, HW, C) bytes of scratch. */
size_t gc_dn_groupnorm_workspace_bytes(int64_t B, int64_t HW, int C);
int gc_dn_groupnorm(int dtype, const void *x, void *y, int64_t B, int64_t HW, int C, int G, const float *gamma,
                    const float *beta, float eps, int act, float *stats_ws, void *stream);
/* The first two passes of gc_dn_groupnorm only: coef[B][C][2] = (a, d) with GroupNorm(x)[b, :, c] = x * a + d.  Input of gc_dn_transformer_head. */
int gc_dn_groupnorm_coef(int dtype, const void *x, int64_t B, int64_t HW, int C, int G, const float *gamma, const float *beta, float eps,
                         float *stats_ws, float *coef, void *stream);
/* GroupNorm(+SiLU) from producer-side statistics: group_stats[B][G][2] = per (batch, group) (sum, sum of squares) over the HW pixels and
 * the C / G channels of the group, accumulated by the kernel that wrote x (gc_gemm_desc.out_group_stats, gc_dn_concat_add).
 * One launch instead of three. */
int gc_dn_groupnorm_apply(int dtype, const void *x, void *y, int64_t B, int64_t HW, int C, int G, const float *gamma,
                          const float *beta, float eps, int act, const float *group_stats, void *stream);
/* The same with an OCP fp8 (e4m3) output for the fp8 convolution path: y8[B][HW][C_padded] bytes (C_padded % 128 == 0, padding = 0),
 * stored = y * 2^(127 - a_scale) saturated to +-448 (a_scale = E8M0 byte handed to gc_gemm_desc.a_scale). */
int gc_dn_groupnorm_apply_fp8(int dtype, const void *x, void *y8, int64_t B, int64_t HW, int C, int C_padded, int G, const float *gamma,
                              const float *beta, float eps, int act, const float *group_stats, int a_scale, void *stream);
/* per-(batch, GroupNorm group) (sum, sum of squares) of x[B][HW][C], ADDED into the caller-zeroed group_stats[B][G][2] (one streaming launch). */
int gc_dn_group_stats(int dtype, const void *x, int64_t B, int64_t HW, int C, int G, float *group_stats, void *stream);
/* LayerNorm over C on [M][C]. */
int gc_dn_layernorm(int dtype, const void *x, void *y, int64_t M, int C, const float *gamma, const float *beta,
                    float eps, void *stream);
/* LayerNorm over C on [M][C] with an OCP fp8 (e4m3) output y8[M][C] (bytes; C % 16 == 0): stored = y * 2^(127 - a_scale), saturated to
 * +-448 -- the activation operand of the fp8 linears that consume a LayerNorm (Q | K | V, attn2.to_q, the GEGLU projection). */
int gc_dn_layernorm_fp8(int dtype, const void *x, void *y8, int64_t M, int C, const float *gamma, const float *beta,
                        float eps, int a_scale, void *stream);
/* out[M][C1+C2] = [a | b (+ c)] : skip concat of the up blocks with the ControlNet residual add folded in.
 * group_stats (optional, caller-zeroed [M / rows_per_batch][gn_groups][2]): per (batch, GroupNorm group) (sum, sum of squares) of `out`. */
int gc_dn_concat_add(int dtype, const void *a, int C1, const void *b, const void *c, int C2, void *out, int64_t M,
                     int64_t rows_per_batch, float *group_stats, int gn_groups, void *stream);

/* GroupNorm(+SiLU) whose statistics pass was done by the PRODUCER of x: `parts` [B][nslab][G][2][2] are partial (sum, sum of squares)
 * per row slab, group and half as left by gc_gemm_desc.out_chan_parts (slab_mode 0: slabs = the GEMM's row tiles of rows_per_slab rows
 * counted over all B * HW rows) or gc_dn_concat_add_parts (slab_mode 1: slabs restart at every batch); col_tile from the same layout
 * query.  ONE launch instead of three (gc_dn_groupnorm): the few KB of partials, the moments and the coefficients are formed in the apply
 * kernel's prologue.  Semantics = torch.nn.GroupNorm of diffusers' ResnetBlock2D / Transformer2DModel reached from
 * /root/reference/gaussctrl/gc_pipeline.py:142-145,209-219. */
int gc_dn_groupnorm_apply_parts(int dtype, const void *x, void *y, int64_t B, int64_t HW, int C, int G, const float *gamma, const float *beta,
                                float eps, int act, const float *parts, int64_t rows_per_slab, int nslab, int slab_mode, int col_tile, void *stream);
/* The same with an OCP fp8 (e4m3) output y8[B][HW][C_padded] (bytes, C_padded % 128 == 0, padding channels zero; stored = value *
 * 2^(127 - a_scale), saturated to +-448): GroupNorm(+SiLU) in ONE launch as the input of an fp8 convolution (gc_dn_groupnorm_apply_fp8
 * with the producer's partials instead of a statistics pass). */
int gc_dn_groupnorm_apply_parts_fp8(int dtype, const void *x, void *y8, int64_t B, int64_t HW, int C, int C_padded, int G, const float *gamma,
                                    const float *beta, float eps, int act, const float *parts, int64_t rows_per_slab, int nslab, int slab_mode,
                                    int col_tile, int a_scale, void *stream);
/* the coefficients alone: coef [B][C][2], y = x * coef[.][c][0] + coef[.][c][1] (input of gc_dn_transformer_head) */
int gc_dn_groupnorm_coef_parts(int64_t B, int64_t HW, int C, int G, const float *gamma, const float *beta, float eps, const float *parts,
                               int64_t rows_per_slab, int nslab, int slab_mode, int col_tile, float *coef, void *stream);
/* gc_dn_concat_add that also leaves the group partials of its output: parts [M / rows_per_batch][nslab][gn_groups][2][2], slab_mode 1,
 * layout from gc_dn_concat_parts_layout (*rows_per_slab = 0: channel slices would cut groups -- use gc_dn_concat_add) */
int gc_dn_concat_parts_layout(int64_t rows_per_batch, int C, int gn_groups, int64_t *rows_per_slab, int *nslab, int *col_tile);
int gc_dn_concat_add_parts(int dtype, const void *a, int C1, const void *b, const void *c, int C2, void *out, int64_t M, int64_t rows_per_batch,
                           int gn_groups, float *parts, void *stream);
/* out = act(a*sa + b*sb) over n elements (b may be NULL). */
int gc_dn_axpby(int dtype, const void *a, float sa, const void *b, float sb, int act, void *out, int64_t n, void *stream);
/* float32 -> dtype with optional SiLU (time-embedding vectors). */
int gc_dn_cast_f32(int dtype, const float *a, int act, void *out, int64_t n, void *stream);
/* in-place row softmax(scale * s) on [M][ld] (VAE mid-block attention). */
int gc_dn_softmax_rows(int dtype, void *s, int64_t M, int64_t N, int64_t ld, float scale, void *stream);
/* CFG combine + DDIM / inverse-DDIM step (eta = 0) + re-pack of the next UNet input (pipeline `cat([latents]*2)`):
 * eps float32 [cfg ? 2f : f][HW][ld_eps], latents float32 [f][HW][4] (updated in place),
 * xin dtype [nrep*f][HW][8].  alpha_t / alpha_prev are the two alphas_cumprod of the step. */
/* depth2disparity_torch (gaussctrl/gc_pipeline.py:258-266): out[HW][8] (channels 0..2 = 1/(d+1e-5) / max, 3..7 = 0).
 * max_ws: device uint32[1] scratch. */
int gc_dn_depth_to_disparity(int dtype, const float *depth, int64_t HW, void *out, unsigned *max_ws, void *stream);
/* mask compositing of edit_images (gaussctrl/gc_pipeline.py:226-234): out[HW][3] f32 = edited*mask + unedited*(1-mask);
 * edited f32 [HW][ld_e] (channels-last, 3 used), unedited f32 [HW][3], mask f32 [HW] or NULL (copy). */
int gc_dn_mask_composite(const float *edited, int ld_e, const float *unedited, const float *mask, float *out, int64_t HW,
                         void *stream);
int gc_dn_cfg_ddim_step(int dtype, const float *eps, int ld_eps, int64_t frames, int64_t HW, float guidance, int cfg,
                        float alpha_t, float alpha_prev, float *latents, void *xin, int nrep, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GAUSSCTRL_HIP_H */
