// raster_composite.hip -- per-tile alpha compositing (forward + backward) for gfx950.
//
// Replaces gsplat 0.1.3's rasterize_forward / rasterize_backward reached from
// /root/reference/gaussctrl/gc_model.py:174-186,191-202 and the autograd fired by gc_trainer.py:275.
// Semantics: SURVEY.md Appendix A.4 / A.5 (alpha cap 0.999, cull alpha < 1/255 or sigma < 0, stop
// when T*(1-alpha) <= 1e-4, pixel (j,i) sampled at (j,i)).
//
// CDNA4 mapping: one 16x16 tile = one 256-lane workgroup = 4 wave64s, each wave owning a 16x4 pixel
// strip.  The tile's depth-sorted splat list is staged through LDS in batches of 256 records
// (48 B each, two ds_read_b128 + one ds_read_b64 per record, all lanes reading the same address ->
// LDS broadcast, no bank conflicts).  The RGB pass and the reference's second "depth" pass are one
// sweep (extra channel).  Backward replays the list back-to-front starting at the workgroup's
// largest final_index (not at the end of the tile list), reduces the nine per-splat partials over
// the 64 lanes with DPP row operations and issues one hardware float atomic per value per wave.
#include "common.h"

namespace {

constexpr int TILE = 16;
constexpr int BLOCK = TILE * TILE;
constexpr float ALPHA_CAP = 0.999f;
constexpr float ALPHA_MIN = 1.f / 255.f;
constexpr float T_STOP = 1e-4f;

struct SplatA { float x, y, opac, cxx; };
struct SplatB { float cxy, cyy, r, g; };
struct SplatC { float b, e; };

template <bool HAS_EXTRA>
__global__ __launch_bounds__(BLOCK) void k_rasterize_fwd(int H, int W, int tiles_x,
                                                         const int32_t *__restrict__ ids_sorted,
                                                         const int32_t *__restrict__ tile_bins,
                                                         const float *__restrict__ xys, const float *__restrict__ conics,
                                                         const float *__restrict__ colors, const float *__restrict__ opacities,
                                                         const float *__restrict__ extra, const float *__restrict__ background,
                                                         float *__restrict__ out_img, float *__restrict__ out_extra,
                                                         float *__restrict__ final_Ts, int32_t *__restrict__ final_index)
{
    __shared__ SplatA sA[BLOCK];
    __shared__ SplatB sB[BLOCK];
    __shared__ SplatC sC[BLOCK];
    const int tile = blockIdx.y * tiles_x + blockIdx.x;
    const int tid = threadIdx.x;
    const int j = blockIdx.x * TILE + (tid & 15);
    const int i = blockIdx.y * TILE + (tid >> 4);
    const bool inside = (i < H) && (j < W);
    const float px = (float)j, py = (float)i;
    const int start = tile_bins[2 * tile], end = tile_bins[2 * tile + 1];
    bool done = !inside;
    float T = 1.f, r = 0.f, g = 0.f, b = 0.f, e = 0.f;
    int last = 0;
    for (int bs = start; bs < end; bs += BLOCK) {
        if (__syncthreads_and(done)) break;
        const int idx = bs + tid;
        if (idx < end) {
            const int gid = ids_sorted[idx];
            const float2 xy = *reinterpret_cast<const float2 *>(xys + 2 * gid);
            const float c0 = conics[3 * gid], c1 = conics[3 * gid + 1], c2 = conics[3 * gid + 2];
            sA[tid] = {xy.x, xy.y, opacities[gid], c0};
            sB[tid] = {c1, c2, colors[3 * gid], colors[3 * gid + 1]};
            sC[tid] = {colors[3 * gid + 2], HAS_EXTRA ? extra[gid] : 0.f};
        }
        __syncthreads();
        const int n = min(BLOCK, end - bs);
        if (!__all(done)) {   // a finished wave only helps staging
            for (int t = 0; t < n && !done; ++t) {
                const SplatA a = sA[t];
                const SplatB bb = sB[t];
                const float dx = a.x - px, dy = a.y - py;
                const float sigma = 0.5f * (a.cxx * dx * dx + bb.cyy * dy * dy) + bb.cxy * dx * dy;
                const float alpha = fminf(ALPHA_CAP, a.opac * __expf(-sigma));
                if (sigma < 0.f || alpha < ALPHA_MIN) continue;
                const float next_T = T * (1.f - alpha);
                if (next_T <= T_STOP) { done = true; break; }
                const float vis = alpha * T;
                const SplatC cc = sC[t];
                r += bb.r * vis; g += bb.g * vis; b += cc.b * vis;
                if (HAS_EXTRA) e += cc.e * vis;
                T = next_T;
                last = bs + t;
            }
        }
    }
    if (inside) {
        const int pix = i * W + j;
        final_Ts[pix] = T;
        final_index[pix] = last;
        out_img[3 * pix] = r + T * background[0];
        out_img[3 * pix + 1] = g + T * background[1];
        out_img[3 * pix + 2] = b + T * background[2];
        if (HAS_EXTRA) out_extra[pix] = e;
    }
}

// ---- wave64 sum via DPP; the total lands in lane 63 and is broadcast with readlane -------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov0(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}

__device__ __forceinline__ float wave_sum(float v)
{
    v += dpp_mov0<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov0<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov0<0x141, 0xF>(v);   // row_half_mirror
    v += dpp_mov0<0x140, 0xF>(v);   // row_mirror  -> every lane holds its 16-lane row sum
    v += dpp_mov0<0x142, 0xA>(v);   // row_bcast:15 into rows 1,3
    v += dpp_mov0<0x143, 0xC>(v);   // row_bcast:31 into rows 2,3 -> lane 63 = total
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__global__ __launch_bounds__(BLOCK) void k_rasterize_bwd(int H, int W, int tiles_x,
                                                         const int32_t *__restrict__ ids_sorted,
                                                         const int32_t *__restrict__ tile_bins,
                                                         const float *__restrict__ xys, const float *__restrict__ conics,
                                                         const float *__restrict__ colors, const float *__restrict__ opacities,
                                                         const float *__restrict__ background,
                                                         const float *__restrict__ final_Ts, const int32_t *__restrict__ final_index,
                                                         const float *__restrict__ v_out, const float *__restrict__ v_out_alpha,
                                                         float *__restrict__ v_xy, float *__restrict__ v_conic,
                                                         float *__restrict__ v_colors, float *__restrict__ v_opacity)
{
    __shared__ SplatA sA[BLOCK];
    __shared__ SplatB sB[BLOCK];
    __shared__ float sBlue[BLOCK];
    __shared__ int sId[BLOCK];
    __shared__ int sMax[4];
    const int tile = blockIdx.y * tiles_x + blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int j = blockIdx.x * TILE + (tid & 15);
    const int i = blockIdx.y * TILE + (tid >> 4);
    const bool inside = (i < H) && (j < W);
    const float px = (float)j, py = (float)i;
    const int start = tile_bins[2 * tile], end = tile_bins[2 * tile + 1];
    if (end <= start) return;
    const int pix = inside ? i * W + j : 0;
    const float T_final = inside ? final_Ts[pix] : 1.f;
    float T = T_final;
    const int bin_final = inside ? final_index[pix] : -1;
    float vo0 = 0.f, vo1 = 0.f, vo2 = 0.f, voa = 0.f;
    if (inside) {
        vo0 = v_out[3 * pix]; vo1 = v_out[3 * pix + 1]; vo2 = v_out[3 * pix + 2];
        if (v_out_alpha) voa = v_out_alpha[pix];
    }
    const float bgdot = background[0] * vo0 + background[1] * vo1 + background[2] * vo2;
    float S0 = 0.f, S1 = 0.f, S2 = 0.f;
    // wave / workgroup maxima of final_index: nothing beyond them was composited
    int wmax = bin_final;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wmax = max(wmax, __shfl_xor(wmax, d, 64));
    if (lane == 0) sMax[wid] = wmax;
    __syncthreads();
    const int kmax = max(max(sMax[0], sMax[1]), max(sMax[2], sMax[3]));
    if (kmax < start) return;
    for (int batch_end = kmax; batch_end >= start; batch_end -= BLOCK) {
        __syncthreads();
        const int idx = batch_end - tid;
        if (idx >= start) {
            const int gid = ids_sorted[idx];
            const float2 xy = *reinterpret_cast<const float2 *>(xys + 2 * gid);
            sId[tid] = gid;
            sA[tid] = {xy.x, xy.y, opacities[gid], conics[3 * gid]};
            sB[tid] = {conics[3 * gid + 1], conics[3 * gid + 2], colors[3 * gid], colors[3 * gid + 1]};
            sBlue[tid] = colors[3 * gid + 2];
        }
        __syncthreads();
        const int n = min(BLOCK, batch_end - start + 1);
        for (int t = max(0, batch_end - wmax); t < n; ++t) {   // wave-uniform bounds
            const int k = batch_end - t;
            const SplatA a = sA[t];
            const SplatB bb = sB[t];
            const float dx = a.x - px, dy = a.y - py;
            const float sigma = 0.5f * (a.cxx * dx * dx + bb.cyy * dy * dy) + bb.cxy * dx * dy;
            const float vis = __expf(-sigma);
            const float araw = a.opac * vis;
            const float alpha = fminf(ALPHA_CAP, araw);
            const bool valid = (k <= bin_final) && !(sigma < 0.f || alpha < ALPHA_MIN);
            if (!__any(valid)) continue;
            float g_r = 0.f, g_g = 0.f, g_b = 0.f, g_cxx = 0.f, g_cxy = 0.f, g_cyy = 0.f, g_x = 0.f, g_y = 0.f, g_o = 0.f;
            if (valid) {
                const float ra = 1.f / (1.f - alpha);
                T *= ra;
                const float fac = alpha * T;
                g_r = fac * vo0; g_g = fac * vo1; g_b = fac * vo2;
                const float cb = sBlue[t];
                float v_alpha = (bb.r * T - S0 * ra) * vo0 + (bb.g * T - S1 * ra) * vo1 + (cb * T - S2 * ra) * vo2;
                v_alpha += T_final * ra * voa;
                v_alpha += -T_final * ra * bgdot;
                S0 += bb.r * fac; S1 += bb.g * fac; S2 += cb * fac;
                if (!(araw > ALPHA_CAP)) {
                    const float v_sigma = -a.opac * vis * v_alpha;
                    g_cxx = 0.5f * v_sigma * dx * dx;
                    g_cxy = v_sigma * dx * dy;
                    g_cyy = 0.5f * v_sigma * dy * dy;
                    g_x = v_sigma * (a.cxx * dx + bb.cxy * dy);
                    g_y = v_sigma * (bb.cxy * dx + bb.cyy * dy);
                    g_o = vis * v_alpha;
                }
            }
            g_r = wave_sum(g_r); g_g = wave_sum(g_g); g_b = wave_sum(g_b);
            g_cxx = wave_sum(g_cxx); g_cxy = wave_sum(g_cxy); g_cyy = wave_sum(g_cyy);
            g_x = wave_sum(g_x); g_y = wave_sum(g_y); g_o = wave_sum(g_o);
            if (lane == 0) {
                const int gid = sId[t];
                unsafeAtomicAdd(v_colors + 3 * gid, g_r);
                unsafeAtomicAdd(v_colors + 3 * gid + 1, g_g);
                unsafeAtomicAdd(v_colors + 3 * gid + 2, g_b);
                unsafeAtomicAdd(v_conic + 3 * gid, g_cxx);
                unsafeAtomicAdd(v_conic + 3 * gid + 1, g_cxy);
                unsafeAtomicAdd(v_conic + 3 * gid + 2, g_cyy);
                unsafeAtomicAdd(v_xy + 2 * gid, g_x);
                unsafeAtomicAdd(v_xy + 2 * gid + 1, g_y);
                unsafeAtomicAdd(v_opacity + gid, g_o);
            }
        }
    }
}

}  // namespace

extern "C" {

int gc_rasterize_fwd(int img_h, int img_w, int tiles_x, int tiles_y, const int32_t *gaussian_ids_sorted,
                     const int32_t *tile_bins, const float *xys, const float *conics, const float *colors,
                     const float *opacities, const float *extra, const float *background, float *out_img,
                     float *out_extra, float *final_Ts, int32_t *final_index, void *stream)
{
    GC_REQUIRE(img_h > 0 && img_w > 0 && tiles_x == (img_w + TILE - 1) / TILE && tiles_y == (img_h + TILE - 1) / TILE,
               "tile bounds do not match the image size");
    GC_REQUIRE((extra == nullptr) == (out_extra == nullptr), "extra and out_extra must be given together");
    dim3 grid(tiles_x, tiles_y), block(BLOCK);
    if (extra)
        hipLaunchKernelGGL(k_rasterize_fwd<true>, grid, block, 0, gc::S(stream), img_h, img_w, tiles_x, gaussian_ids_sorted,
                           tile_bins, xys, conics, colors, opacities, extra, background, out_img, out_extra, final_Ts, final_index);
    else
        hipLaunchKernelGGL(k_rasterize_fwd<false>, grid, block, 0, gc::S(stream), img_h, img_w, tiles_x, gaussian_ids_sorted,
                           tile_bins, xys, conics, colors, opacities, extra, background, out_img, out_extra, final_Ts, final_index);
    return gc::check_launch("gc_rasterize_fwd");
}

int gc_rasterize_bwd(int img_h, int img_w, int tiles_x, int tiles_y, int64_t N, const int32_t *gaussian_ids_sorted,
                     const int32_t *tile_bins, const float *xys, const float *conics, const float *colors,
                     const float *opacities, const float *background, const float *final_Ts, const int32_t *final_index,
                     const float *v_out, const float *v_out_alpha, float *v_xy, float *v_conic, float *v_colors,
                     float *v_opacity, void *stream)
{
    (void)N;
    GC_REQUIRE(img_h > 0 && img_w > 0 && tiles_x == (img_w + TILE - 1) / TILE && tiles_y == (img_h + TILE - 1) / TILE,
               "tile bounds do not match the image size");
    dim3 grid(tiles_x, tiles_y), block(BLOCK);
    hipLaunchKernelGGL(k_rasterize_bwd, grid, block, 0, gc::S(stream), img_h, img_w, tiles_x, gaussian_ids_sorted, tile_bins,
                       xys, conics, colors, opacities, background, final_Ts, final_index, v_out, v_out_alpha, v_xy, v_conic,
                       v_colors, v_opacity);
    return gc::check_launch("gc_rasterize_bwd");
}

}  // extern "C"
