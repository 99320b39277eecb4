// raster_composite.hip -- per-tile alpha compositing (forward + backward) for gfx950.
//
// Replaces gsplat 0.1.3's rasterize_forward / rasterize_backward reached from
// /root/reference/gaussctrl/gc_model.py:174-186,191-202 and the autograd fired by gc_trainer.py:275.
// Semantics: SURVEY.md Appendix A.4 / A.5 (alpha cap 0.999, cull alpha < 1/255 or sigma < 0, stop
// when T*(1-alpha) <= 1e-4, pixel (j,i) sampled at (j,i)).
//
// CDNA4 mapping: one 16x16 tile = one 256-lane workgroup = 4 wave64s, each wave owning an 8x8 pixel
// block (a compact block is touched by fewer splats than a 16x4 strip: fwd 318 -> 293 us, bwd 515 -> 479 us at
// N = 4 M).  The tile's depth-sorted splat list is staged through LDS in batches of 256 records
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

// Batched views (round 5: gc_rasterize_fwd_views / gc_rasterize_bwd_views): blockIdx.z = view (camera); per-view arrays are [C][...] with
// the strides below (elements).  A single-view launch has gridDim.z == 1: every offset is 0.
struct CV {
    int64_t n;       // per-Gaussian arrays of a view: xys / 2, conics / 3, colors / 3, extra, v_* ([C][N])
    int64_t n_op;    // opacities: N when per view, 0 when one array serves every view (sigmoid(opacity) does not depend on the camera)
    int64_t m;       // gaussian_ids_sorted ([C][M_cap])
    int tiles;       // tile_bins ([C][T][2])
    int bg;          // background: 3 when per view, 0 when shared
};

struct SplatA { float x, y, opac, cxx; };
struct SplatB { float cxy, cyy, r, g; };
struct SplatC { float b, e; };

// ---- block culling ---------------------------------------------------------------------------------------------------------
// gsplat bins a Gaussian into every tile of the BOX around a circle of 3 sqrt(lambda_max); most (tile, Gaussian) pairs of an anisotropic
// or faint Gaussian never reach alpha >= 1/255 anywhere in the tile, and of the rest few touch all four 8x8 blocks.  When a batch is
// staged, the lane that loads a record also evaluates -- exactly, the form is convex -- the minimum of sigma over each block's
// rectangle of pixel centres and keeps a 4-bit mask "block w can reach alpha >= 1/255" (threshold sigma <= ln(255 opacity), with a
// margin far above the rounding of either side).  A wave then walks only the set bits of its block's ballot (scalar loop: s_ff1 +
// s_andn2), so culled pairs cost no vector work at all.  The per-pixel test is unchanged: results are bit-identical to the unculled loop.
__device__ __forceinline__ float edge_min(float a, float b, float c, float rc, float e, float lo, float hi)
{
    // min over v in [lo, hi] of 0.5 (a e^2 + c v^2) + b e v   (c > 0, rc = 1-ulp reciprocal of c): v* = clamp(-b e / c).  Evaluating at a
    // v that is off by delta overestimates the minimum by c delta^2 / 2 ~ 1e-14 sigma -- twelve orders below the margin kept on tau
    const float v = fminf(fmaxf(-b * e * rc, lo), hi);
    return 0.5f * (a * e * e + c * v * v) + b * e * v;
}

__device__ __forceinline__ unsigned block_mask(const float x, const float y, const float opac, const float cxx, const float cxy,
                                               const float cyy, const float tile_x0, const float tile_y0)
{
#ifdef GC_NO_BLOCK_CULL                                                  // A/B builds only (tests compare culled vs unculled)
    return 0xFu;
#endif
    if (!(cxx > 0.f && cyy > 0.f)) return 0xFu;                       // degenerate conic: no culling
    const float tau = __logf(255.f * opac) * 1.001f + 0.01f;          // alpha >= 1/255  <=>  sigma <= ln(255 opacity)
    if (!(tau >= 0.f)) return tau < 0.f ? 0u : 0xFu;                  // NaN -> keep
    const float rxx = __builtin_amdgcn_rcpf(cxx), ryy = __builtin_amdgcn_rcpf(cyy);
    unsigned m = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {                                     // wave w owns the 8x8 block (w & 1, w >> 1) of the tile
        const float bx = tile_x0 + 8.f * (w & 1), by = tile_y0 + 8.f * (w >> 1);
        const float dx0 = x - (bx + 7.f), dx1 = x - bx, dy0 = y - (by + 7.f), dy1 = y - by;      // d = splat - pixel over the block
        float smin;
        if (dx0 <= 0.f && dx1 >= 0.f && dy0 <= 0.f && dy1 >= 0.f) smin = 0.f;
        else {
            smin = fminf(fminf(edge_min(cxx, cxy, cyy, ryy, dx0, dy0, dy1), edge_min(cxx, cxy, cyy, ryy, dx1, dy0, dy1)),
                         fminf(edge_min(cyy, cxy, cxx, rxx, dy0, dx0, dx1), edge_min(cyy, cxy, cxx, rxx, dy1, dx0, dx1)));
        }
        if (!(smin > tau)) m |= 1u << w;
    }
    return m;
}

template <bool HAS_EXTRA>
__global__ __launch_bounds__(BLOCK) void k_rasterize_fwd(int H, int W, int tiles_x,
                                                         const int32_t *__restrict__ ids_sorted,
                                                         const int32_t *__restrict__ tile_bins,
                                                         const float *__restrict__ xys, const float *__restrict__ conics,
                                                         const float *__restrict__ colors, const float *__restrict__ opacities,
                                                         const float *__restrict__ extra, const float *__restrict__ background,
                                                         float *__restrict__ out_img, float *__restrict__ out_extra,
                                                         float *__restrict__ final_Ts, int32_t *__restrict__ final_index, CV cv)
{
    {
        const int64_t v = blockIdx.z, hw = (int64_t)H * W;
        ids_sorted += v * cv.m; tile_bins += v * 2 * cv.tiles; xys += v * 2 * cv.n; conics += v * 3 * cv.n; colors += v * 3 * cv.n;
        opacities += v * cv.n_op; background += v * cv.bg;
        if (HAS_EXTRA) { extra += v * cv.n; out_extra += v * hw; }
        out_img += v * 3 * hw; final_Ts += v * hw; final_index += v * hw;
    }
    __shared__ SplatA sA[BLOCK];
    __shared__ SplatB sB[BLOCK];
    __shared__ SplatC sC[BLOCK];
    __shared__ unsigned char sMask[BLOCK];
    const int tile = blockIdx.y * tiles_x + blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int j = blockIdx.x * TILE + 8 * (wid & 1) + (lane & 7);       // wave = one 8x8 pixel block of the tile
    const int i = blockIdx.y * TILE + 8 * (wid >> 1) + (lane >> 3);
    const bool inside = (i < H) && (j < W);
    const float px = (float)j, py = (float)i;
    const float tx0 = (float)(blockIdx.x * TILE), ty0 = (float)(blockIdx.y * TILE);
    const int start = tile_bins[2 * tile], end = tile_bins[2 * tile + 1];
    bool done = !inside;
    float T = 1.f, r = 0.f, g = 0.f, b = 0.f, e = 0.f;
    int last = 0;
    for (int bs = start; bs < end; bs += BLOCK) {
        if (__syncthreads_and(done)) break;
        const int idx = bs + tid;
        unsigned mk = 0;
        if (idx < end) {
            const int gid = ids_sorted[idx];
            const float2 xy = *reinterpret_cast<const float2 *>(xys + 2 * gid);
            const float c0 = conics[3 * gid], c1 = conics[3 * gid + 1], c2 = conics[3 * gid + 2];
            const float op = opacities[gid];
            sA[tid] = {xy.x, xy.y, op, c0};
            sB[tid] = {c1, c2, colors[3 * gid], colors[3 * gid + 1]};
            sC[tid] = {colors[3 * gid + 2], HAS_EXTRA ? extra[gid] : 0.f};
            mk = block_mask(xy.x, xy.y, op, c0, c1, c2, tx0, ty0);
        }
        sMask[tid] = (unsigned char)mk;
        __syncthreads();
        const int n = min(BLOCK, end - bs);
        bool wave_done = __all(done);                                  // a finished wave only helps staging
        for (int c = 0; c * 64 < n && !wave_done; ++c) {
            unsigned long long bal = __ballot((sMask[c * 64 + lane] >> wid) & 1);
            if (!bal) continue;
            // scalar walk over the splats that can touch this block; branch-free body (selects), the next record is read from LDS
            // while the current one is evaluated
            int t = c * 64 + __builtin_ctzll(bal);
            bal &= bal - 1;
            SplatA a = sA[t];
            SplatB bb = sB[t];
            SplatC cc = sC[t];
            for (;;) {
                const int tn = c * 64 + (bal ? __builtin_ctzll(bal) : 0);
                const bool more = bal != 0;
                bal &= bal - 1;
                const SplatA an = sA[tn];
                const SplatB bn = sB[tn];
                const SplatC cn = sC[tn];
                const float dx = a.x - px, dy = a.y - py;
                const float sigma = 0.5f * (a.cxx * dx * dx + bb.cyy * dy * dy) + bb.cxy * dx * dy;
                const float alpha = fminf(ALPHA_CAP, a.opac * __expf(-sigma));
                const bool hit = !done && !(sigma < 0.f || alpha < ALPHA_MIN);
                const float next_T = T * (1.f - alpha);
                const bool stop = hit && next_T <= T_STOP;
                const bool add = hit && !stop;
                const float vis = add ? alpha * T : 0.f;
                r += bb.r * vis; g += bb.g * vis; b += cc.b * vis;
                if (HAS_EXTRA) e += cc.e * vis;
                T = add ? next_T : T;
                last = add ? bs + t : last;
                done |= stop;
                if (!more) break;
                if (__all(done)) { wave_done = true; break; }
                t = tn; a = an; bb = bn; cc = cn;
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

// ---- reduction of the nine per-splat partials ------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v)      // v + v[dpp lane], all rows / banks
{
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float row_sum(float v)      // every lane ends with the sum over its row of 16 lanes
{
    v = dpp_add<0xB1>(v);     // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);     // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);    // row_half_mirror
    v = dpp_add<0x140>(v);    // row_mirror
    return v;
}
__device__ __forceinline__ float xor_rows_sum(float v)  // sum over the four rows, lane-wise (lane l: lanes l%16 + 16 k)
{
    {
        const unsigned x = __float_as_uint(v);
        const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    {
        const unsigned x = __float_as_uint(v);
        const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    return v;
}

// Backward.  Per (block, splat) with at least one contributing pixel: nine partials are summed over the rows with DPP (36 adds), lane c
// of every row picks value c, two lane-swap steps add the four rows, and lanes 0..8 add their value into a per-batch LDS accumulator
// (ONE ds_add_f32 instruction).  After a batch the staging lane of each splat flushes nine hardware float atomics -- once per
// (tile, splat) instead of once per (block, splat), issued by 256 lanes at a time.
__global__ __launch_bounds__(BLOCK) void k_rasterize_bwd(int H, int W, int tiles_x,
                                                         const int32_t *__restrict__ ids_sorted,
                                                         const int32_t *__restrict__ tile_bins,
                                                         const float *__restrict__ xys, const float *__restrict__ conics,
                                                         const float *__restrict__ colors, const float *__restrict__ opacities,
                                                         const float *__restrict__ background,
                                                         const float *__restrict__ final_Ts, const int32_t *__restrict__ final_index,
                                                         const float *__restrict__ v_out, const float *__restrict__ v_out_alpha,
                                                         const float *__restrict__ pre_clamp,
                                                         float *__restrict__ v_xy, float *__restrict__ v_conic,
                                                         float *__restrict__ v_colors, float *__restrict__ v_opacity, CV cv)
{
    {
        const int64_t v = blockIdx.z, hw = (int64_t)H * W;
        ids_sorted += v * cv.m; tile_bins += v * 2 * cv.tiles; xys += v * 2 * cv.n; conics += v * 3 * cv.n; colors += v * 3 * cv.n;
        opacities += v * cv.n_op; background += v * cv.bg;
        final_Ts += v * hw; final_index += v * hw; v_out += v * 3 * hw;
        if (v_out_alpha) v_out_alpha += v * hw;
        if (pre_clamp) pre_clamp += v * 3 * hw;
        v_xy += v * 2 * cv.n; v_conic += v * 3 * cv.n; v_colors += v * 3 * cv.n; v_opacity += v * cv.n;
    }
    __shared__ SplatA sA[BLOCK];
    __shared__ SplatB sB[BLOCK];
    __shared__ float sBlue[BLOCK];
    __shared__ float sG[BLOCK * 9];
    __shared__ unsigned char sMask[BLOCK];
    __shared__ int sMax[4];
    const int tile = blockIdx.y * tiles_x + blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int j = blockIdx.x * TILE + 8 * (wid & 1) + (lane & 7);       // wave = one 8x8 pixel block of the tile
    const int i = blockIdx.y * TILE + 8 * (wid >> 1) + (lane >> 3);
    const bool inside = (i < H) && (j < W);
    const float px = (float)j, py = (float)i;
    const float tx0 = (float)(blockIdx.x * TILE), ty0 = (float)(blockIdx.y * TILE);
    const int start = tile_bins[2 * tile], end = tile_bins[2 * tile + 1];
    if (end <= start) return;
    const int pix = inside ? i * W + j : 0;
    const float T_final = inside ? final_Ts[pix] : 1.f;
    float T = T_final;
    const int bin_final = inside ? final_index[pix] : -1;
    float vo0 = 0.f, vo1 = 0.f, vo2 = 0.f, voa = 0.f;
    if (inside) {
        vo0 = v_out[3 * pix]; vo1 = v_out[3 * pix + 1]; vo2 = v_out[3 * pix + 2];
        if (pre_clamp) {      // backward of rgb = min(rgb, 1) (gc_model.py:188): the gradient passes where the un-clamped value is <= 1
            vo0 = pre_clamp[3 * pix] <= 1.f ? vo0 : 0.f; vo1 = pre_clamp[3 * pix + 1] <= 1.f ? vo1 : 0.f; vo2 = pre_clamp[3 * pix + 2] <= 1.f ? vo2 : 0.f;
        }
        if (v_out_alpha) voa = v_out_alpha[pix];
    }
    const float bgdot = background[0] * vo0 + background[1] * vo1 + background[2] * vo2;
    float S0 = 0.f, S1 = 0.f, S2 = 0.f;
    // wave / workgroup maxima of final_index: nothing beyond them was composited
    int wmax = bin_final;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wmax = max(wmax, __shfl_xor(wmax, d, 64));
    wmax = __builtin_amdgcn_readfirstlane(wmax);                       // wave-uniform: keeps the splat walk below in scalar registers
    if (lane == 0) sMax[wid] = wmax;
#pragma unroll
    for (int q = 0; q < 9; ++q) sG[q * BLOCK + tid] = 0.f;
    __syncthreads();
    const int kmax = __builtin_amdgcn_readfirstlane(max(max(sMax[0], sMax[1]), max(sMax[2], sMax[3])));
    if (kmax < start) return;
    const int col = lane & 15;
    for (int batch_end = kmax; batch_end >= start; batch_end -= BLOCK) {
        const int idx = batch_end - tid;
        int gid = -1;
        unsigned mk = 0;
        if (idx >= start) {
            gid = ids_sorted[idx];
            const float2 xy = *reinterpret_cast<const float2 *>(xys + 2 * gid);
            const float op = opacities[gid], c0 = conics[3 * gid], c1 = conics[3 * gid + 1], c2 = conics[3 * gid + 2];
            sA[tid] = {xy.x, xy.y, op, c0};
            sB[tid] = {c1, c2, colors[3 * gid], colors[3 * gid + 1]};
            sBlue[tid] = colors[3 * gid + 2];
            mk = block_mask(xy.x, xy.y, op, c0, c1, c2, tx0, ty0);
        }
        sMask[tid] = (unsigned char)mk;
        __syncthreads();
        const int n = min(BLOCK, batch_end - start + 1);
        const int t0 = max(0, batch_end - wmax);                       // wave-uniform: splats behind every pixel's last one
        for (int c = t0 >> 6; c * 64 < n; ++c) {
            unsigned long long bal = __ballot((sMask[c * 64 + lane] >> wid) & 1);
            if (c * 64 < t0) bal &= ~0ull << (t0 - c * 64);
            while (bal) {
                const int t = c * 64 + __builtin_ctzll(bal);
                bal &= bal - 1;
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
                    const float ra = __builtin_amdgcn_rcpf(1.f - alpha);      // 1-ulp reciprocal: alpha <= 0.999, the IEEE division sequence is 10 instructions
                    T *= ra;
                    const float fac = alpha * T;
                    g_r = fac * vo0; g_g = fac * vo1; g_b = fac * vo2;
                    const float cb = sBlue[t];
                    float v_alpha = (bb.r * T - S0 * ra) * vo0 + (bb.g * T - S1 * ra) * vo1 + (cb * T - S2 * ra) * vo2;
                    v_alpha += T_final * ra * voa;
                    v_alpha += -T_final * ra * bgdot;
                    S0 += bb.r * fac; S1 += bb.g * fac; S2 += cb * fac;
                    // alpha clamped at the cap passes no gradient to sigma / opacity (select instead of a nested exec-mask branch)
                    const float va = araw > ALPHA_CAP ? 0.f : vis * v_alpha;
                    const float v_sigma = -a.opac * va;
                    g_cxx = 0.5f * v_sigma * dx * dx;
                    g_cxy = v_sigma * dx * dy;
                    g_cyy = 0.5f * v_sigma * dy * dy;
                    g_x = v_sigma * (a.cxx * dx + bb.cxy * dy);
                    g_y = v_sigma * (bb.cxy * dx + bb.cyy * dy);
                    g_o = va;
                }
                g_r = row_sum(g_r); g_g = row_sum(g_g); g_b = row_sum(g_b);
                g_cxx = row_sum(g_cxx); g_cxy = row_sum(g_cxy); g_cyy = row_sum(g_cyy);
                g_x = row_sum(g_x); g_y = row_sum(g_y); g_o = row_sum(g_o);
                float x = g_r;                                          // lane column c keeps value c
                x = col == 1 ? g_g : x; x = col == 2 ? g_b : x; x = col == 3 ? g_cxx : x; x = col == 4 ? g_cxy : x;
                x = col == 5 ? g_cyy : x; x = col == 6 ? g_x : x; x = col == 7 ? g_y : x; x = col == 8 ? g_o : x;
                x = xor_rows_sum(x);
                if (lane < 9) __hip_atomic_fetch_add(&sG[lane * BLOCK + t], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        __syncthreads();
        if (gid >= 0) {                                                 // flush: nine atomics per (tile, splat) that was touched at all
            float v[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) { v[q] = sG[q * BLOCK + tid]; sG[q * BLOCK + tid] = 0.f; }
            bool any = false;
#pragma unroll
            for (int q = 0; q < 9; ++q) any |= v[q] != 0.f;
            if (any) {
                unsafeAtomicAdd(v_colors + 3 * gid, v[0]);
                unsafeAtomicAdd(v_colors + 3 * gid + 1, v[1]);
                unsafeAtomicAdd(v_colors + 3 * gid + 2, v[2]);
                unsafeAtomicAdd(v_conic + 3 * gid, v[3]);
                unsafeAtomicAdd(v_conic + 3 * gid + 1, v[4]);
                unsafeAtomicAdd(v_conic + 3 * gid + 2, v[5]);
                unsafeAtomicAdd(v_xy + 2 * gid, v[6]);
                unsafeAtomicAdd(v_xy + 2 * gid + 1, v[7]);
                unsafeAtomicAdd(v_opacity + gid, v[8]);
            }
        }
        // the next staging pass overwrites sA / sB / sMask: every wave has left the loop (barrier above); sG slots are private to
        // their staging lane until the next barrier
    }
}

}  // namespace

extern "C" {

static int rasterize_fwd_impl(const char *what, int C, int64_t N, int64_t M_cap, int shared_opacities, int shared_background, int img_h, int img_w,
                              int tiles_x, int tiles_y, const int32_t *gaussian_ids_sorted,
                              const int32_t *tile_bins, const float *xys, const float *conics, const float *colors,
                              const float *opacities, const float *extra, const float *background, float *out_img,
                              float *out_extra, float *final_Ts, int32_t *final_index, void *stream)
{
    if (!(img_h > 0 && img_w > 0 && tiles_x == (img_w + TILE - 1) / TILE && tiles_y == (img_h + TILE - 1) / TILE)) {
        gc::set_error("%s: tile bounds do not match the image size", what); return GC_EINVAL;
    }
    if ((extra == nullptr) != (out_extra == nullptr)) { gc::set_error("%s: extra and out_extra must be given together", what); return GC_EINVAL; }
    CV cv; cv.n = N; cv.n_op = shared_opacities ? 0 : N; cv.m = M_cap; cv.tiles = tiles_x * tiles_y; cv.bg = shared_background ? 0 : 3;
    dim3 grid(tiles_x, tiles_y, C), block(BLOCK);
    if (extra)
        hipLaunchKernelGGL(k_rasterize_fwd<true>, grid, block, 0, gc::S(stream), img_h, img_w, tiles_x, gaussian_ids_sorted,
                           tile_bins, xys, conics, colors, opacities, extra, background, out_img, out_extra, final_Ts, final_index, cv);
    else
        hipLaunchKernelGGL(k_rasterize_fwd<false>, grid, block, 0, gc::S(stream), img_h, img_w, tiles_x, gaussian_ids_sorted,
                           tile_bins, xys, conics, colors, opacities, extra, background, out_img, out_extra, final_Ts, final_index, cv);
    return gc::check_launch(what);
}

int gc_rasterize_fwd(int img_h, int img_w, int tiles_x, int tiles_y, const int32_t *gaussian_ids_sorted,
                     const int32_t *tile_bins, const float *xys, const float *conics, const float *colors,
                     const float *opacities, const float *extra, const float *background, float *out_img,
                     float *out_extra, float *final_Ts, int32_t *final_index, void *stream)
{
    return rasterize_fwd_impl("gc_rasterize_fwd", 1, 0, 0, 1, 1, img_h, img_w, tiles_x, tiles_y, gaussian_ids_sorted, tile_bins, xys, conics, colors,
                              opacities, extra, background, out_img, out_extra, final_Ts, final_index, stream);
}

/* C views in one launch (grid.z = view): gaussian_ids_sorted [C][M_cap], tile_bins [C][T][2], xys / conics / colors / extra [C][N][..],
 * opacities [N] (shared_opacities = 1) or [C][N], background [3] (shared_background = 1) or [C][3]; outputs [C][H][W][..]. */
int gc_rasterize_fwd_views(int C, int64_t N, int64_t M_cap, int shared_opacities, int shared_background, int img_h, int img_w, int tiles_x,
                           int tiles_y, const int32_t *gaussian_ids_sorted, const int32_t *tile_bins, const float *xys, const float *conics,
                           const float *colors, const float *opacities, const float *extra, const float *background, float *out_img,
                           float *out_extra, float *final_Ts, int32_t *final_index, void *stream)
{
    GC_REQUIRE(C >= 1 && C <= 65535 && N >= 0 && M_cap >= 0, "bad arguments");
    return rasterize_fwd_impl("gc_rasterize_fwd_views", C, N, M_cap, shared_opacities, shared_background, img_h, img_w, tiles_x, tiles_y,
                              gaussian_ids_sorted, tile_bins, xys, conics, colors, opacities, extra, background, out_img, out_extra, final_Ts,
                              final_index, stream);
}

static int rasterize_bwd_impl(const char *what, int C, int64_t N, int64_t M_cap, int shared_opacities, int shared_background, int img_h, int img_w,
                              int tiles_x, int tiles_y, const int32_t *gaussian_ids_sorted,
                              const int32_t *tile_bins, const float *xys, const float *conics, const float *colors,
                              const float *opacities, const float *background, const float *final_Ts, const int32_t *final_index,
                              const float *v_out, const float *v_out_alpha, const float *pre_clamp, float *v_xy, float *v_conic,
                              float *v_colors, float *v_opacity, void *stream)
{
    if (!(img_h > 0 && img_w > 0 && tiles_x == (img_w + TILE - 1) / TILE && tiles_y == (img_h + TILE - 1) / TILE)) {
        gc::set_error("%s: tile bounds do not match the image size", what); return GC_EINVAL;
    }
    CV cv; cv.n = N; cv.n_op = shared_opacities ? 0 : N; cv.m = M_cap; cv.tiles = tiles_x * tiles_y; cv.bg = shared_background ? 0 : 3;
    dim3 grid(tiles_x, tiles_y, C), block(BLOCK);
    hipLaunchKernelGGL(k_rasterize_bwd, grid, block, 0, gc::S(stream), img_h, img_w, tiles_x, gaussian_ids_sorted, tile_bins,
                       xys, conics, colors, opacities, background, final_Ts, final_index, v_out, v_out_alpha, pre_clamp, v_xy, v_conic,
                       v_colors, v_opacity, cv);
    return gc::check_launch(what);
}

int gc_rasterize_bwd(int img_h, int img_w, int tiles_x, int tiles_y, int64_t N, const int32_t *gaussian_ids_sorted,
                     const int32_t *tile_bins, const float *xys, const float *conics, const float *colors,
                     const float *opacities, const float *background, const float *final_Ts, const int32_t *final_index,
                     const float *v_out, const float *v_out_alpha, float *v_xy, float *v_conic, float *v_colors,
                     float *v_opacity, void *stream)
{
    return rasterize_bwd_impl("gc_rasterize_bwd", 1, N, 0, 1, 1, img_h, img_w, tiles_x, tiles_y, gaussian_ids_sorted, tile_bins, xys, conics, colors,
                              opacities, background, final_Ts, final_index, v_out, v_out_alpha, nullptr, v_xy, v_conic, v_colors, v_opacity, stream);
}

/* the same with the backward of get_outputs' rgb clamp folded into the pixel load: v_out is masked where pre_clamp > 1 */
int gc_rasterize_bwd_clamped(int img_h, int img_w, int tiles_x, int tiles_y, int64_t N, const int32_t *gaussian_ids_sorted,
                             const int32_t *tile_bins, const float *xys, const float *conics, const float *colors,
                             const float *opacities, const float *background, const float *final_Ts, const int32_t *final_index,
                             const float *v_out, const float *v_out_alpha, const float *pre_clamp, float *v_xy, float *v_conic,
                             float *v_colors, float *v_opacity, void *stream)
{
    GC_REQUIRE(pre_clamp, "pre_clamp image required (gc_rasterize_bwd is the form without the clamp)");
    return rasterize_bwd_impl("gc_rasterize_bwd_clamped", 1, N, 0, 1, 1, img_h, img_w, tiles_x, tiles_y, gaussian_ids_sorted, tile_bins, xys, conics,
                              colors, opacities, background, final_Ts, final_index, v_out, v_out_alpha, pre_clamp, v_xy, v_conic, v_colors, v_opacity,
                              stream);
}

/* C views in one launch: v_out [C][H][W][3], v_out_alpha [C][H][W] or NULL, pre_clamp [C][H][W][3] or NULL (clamp backward folded in);
 * v_xy [C][N][2], v_conic [C][N][3], v_colors [C][N][3], v_opacity [C][N] must be ZERO on entry (the kernel adds into them). */
int gc_rasterize_bwd_views(int C, int64_t N, int64_t M_cap, int shared_opacities, int shared_background, int img_h, int img_w, int tiles_x,
                           int tiles_y, const int32_t *gaussian_ids_sorted, const int32_t *tile_bins, const float *xys, const float *conics,
                           const float *colors, const float *opacities, const float *background, const float *final_Ts,
                           const int32_t *final_index, const float *v_out, const float *v_out_alpha, const float *pre_clamp, float *v_xy,
                           float *v_conic, float *v_colors, float *v_opacity, void *stream)
{
    GC_REQUIRE(C >= 1 && C <= 65535 && N >= 0 && M_cap >= 0, "bad arguments");
    return rasterize_bwd_impl("gc_rasterize_bwd_views", C, N, M_cap, shared_opacities, shared_background, img_h, img_w, tiles_x, tiles_y,
                              gaussian_ids_sorted, tile_bins, xys, conics, colors, opacities, background, final_Ts, final_index, v_out, v_out_alpha,
                              pre_clamp, v_xy, v_conic, v_colors, v_opacity, stream);
}

}  // extern "C"
