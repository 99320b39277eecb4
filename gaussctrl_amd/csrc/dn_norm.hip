// dn_norm.hip -- HBM-bound normalisation / elementwise kernels of the SD1.5 UNet / ControlNet / VAE on
// NHWC ("tokens x channels") bf16/f16 tensors (gfx950).  Replaces torch.nn.GroupNorm / LayerNorm / SiLU /
// cat / add and the scheduler + CFG arithmetic diffusers runs for
// /root/reference/gaussctrl/gc_pipeline.py:142-145,209-219.
//
// All kernels move 16 bytes per lane per access, statistics in fp32.
#include "dn_common.h"

namespace {
using namespace dn;

// ------------------------------------------------------------------------------------------ GroupNorm
// Three streaming kernels, no atomics, 32-bit index arithmetic:
//   k_gn_partial : per (batch, pixel slab <= 32 per image, channel slice) per-CHANNEL sums of (x - s_c) and (x - s_c)^2 with the
//                  per-channel shift s_c = x[b, pixel 0, c] (shifted sums: no catastrophic cancellation when |mean| >> std)
//   k_gn_finalize: per (batch, group): mean / rstd from the shifted channel sums -> per-channel scale a = rstd*gamma, d = beta - mean*a
//   k_gn_apply   : y = (x * a + d) [SiLU], 16 bytes per lane, coefficients from L1/L2
template <class T>
__global__ __launch_bounds__(256) void k_gn_partial(const unsigned short *__restrict__ x, int HW, int C, int nchb,
                                                    int pix_per_block, float *__restrict__ part)
{
    extern __shared__ float sp[];   // [lanes][nchb][16]
    const int b = blockIdx.z;
    const int lanes = 256 / nchb;
    const int tid = threadIdx.x;
    const int cch = tid % nchb, pl = tid / nchb;
    const int c0 = (blockIdx.y * nchb + cch) * 8;
    const unsigned short *xb = x + (size_t)b * HW * C;
    if (pl < lanes) {
        float sh[8], s1[8], s2[8];
        unpack8<T>(*reinterpret_cast<const uint4 *>(xb + c0), sh);
#pragma unroll
        for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
        const int p0 = blockIdx.x * pix_per_block;
        const int p1 = min(p0 + pix_per_block, HW);
        for (int p = p0 + pl; p < p1; p += lanes) {
            float f[8];
            unpack8<T>(*reinterpret_cast<const uint4 *>(xb + (size_t)p * C + c0), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = f[j] - sh[j]; s1[j] += d; s2[j] += d * d; }
        }
        float *o = sp + ((size_t)pl * nchb + cch) * 16;
#pragma unroll
        for (int j = 0; j < 8; ++j) { o[2 * j] = s1[j]; o[2 * j + 1] = s2[j]; }
    }
    __syncthreads();
    if (pl == 0) {
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.f;
        for (int l = 0; l < lanes; ++l) {
            const float *o = sp + ((size_t)l * nchb + cch) * 16;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] += o[j];
        }
        float *dst = part + (((size_t)b * gridDim.x + blockIdx.x) * C + c0) * 2;
#pragma unroll
        for (int j = 0; j < 16; j += 4) *reinterpret_cast<float4 *>(dst + j) = make_float4(acc[j], acc[j + 1], acc[j + 2], acc[j + 3]);
    }
}

// one workgroup per (batch, group): channel c of the group is handled by lanes l = c - g*cpg (+ 256 k)
template <class T>
__global__ __launch_bounds__(256) void k_gn_finalize(const unsigned short *__restrict__ x, int HW, int C, int G, int nslab,
                                                     const float *__restrict__ part, const float *__restrict__ gamma,
                                                     const float *__restrict__ beta, float eps, float *__restrict__ coef)
{
    __shared__ float red[3][4];
    const int b = blockIdx.x / G, gi = blockIdx.x % G, cpg = C / G, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // every thread: one (channel, slab-subset) share; totals per channel are combined with the channel's shift
    float t0 = 0.f, t1 = 0.f, t2 = 0.f;        // sum x, sum (x-s)^2 parts folded below
    float cs1 = 0.f, cs2 = 0.f, sh = 0.f;
    const int ci = tid % cpg, sub = tid / cpg, nsub = 256 / cpg;   // cpg <= 256
    if (sub < nsub) {
        const int c = gi * cpg + ci;
        sh = T::to_f(x[(size_t)b * HW * C + c]);
        for (int sl = sub; sl < nslab; sl += nsub) {
            const float2 v = *reinterpret_cast<const float2 *>(part + (((size_t)b * nslab + sl) * C + c) * 2);
            cs1 += v.x; cs2 += v.y;
        }
        // this share covers n_s pixels (unknown here) -> keep moments about the shift; pixel counts are added in closed form
        t0 = cs1;                       // sum (x - s_c) over the share
        t1 = cs2;                       // sum (x - s_c)^2
        t2 = (sub == 0) ? sh : 0.f;     // s_c counted once per channel
    }
    // group mean: mu = (sum_c [S1_c + HW * s_c]) / (HW * cpg)
    float a0 = wave_sum_f(t0), a2 = wave_sum_f(t2);
    if (lane == 0) { red[0][wid] = a0; red[2][wid] = a2; }
    __syncthreads();
    const float S1 = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    const float SS = red[2][0] + red[2][1] + red[2][2] + red[2][3];
    const float n = (float)HW * (float)cpg;
    const float mu = (S1 + (float)HW * SS) / n;
    // variance: sum_c [S2_c + 2 (s_c - mu) S1_c + HW (s_c - mu)^2]: the last term once per channel
    float q = 0.f;
    if (sub < nsub) {
        const float dlt = sh - mu;
        q = t1 + 2.f * dlt * t0 + ((sub == 0) ? (float)HW * dlt * dlt : 0.f);
    }
    float aq = wave_sum_f(q);
    if (lane == 0) red[1][wid] = aq;
    __syncthreads();
    const float var = fmaxf((red[1][0] + red[1][1] + red[1][2] + red[1][3]) / n, 0.f);
    const float rstd = rsqrtf(var + eps);
    for (int c = gi * cpg + tid; c < (gi + 1) * cpg; c += 256) {
        const float a = rstd * gamma[c];
        *reinterpret_cast<float2 *>(coef + ((size_t)b * C + c) * 2) = make_float2(a, beta[c] - mu * a);
    }
}

template <class T>
__global__ __launch_bounds__(256) void k_gn_apply(const unsigned short *__restrict__ x, unsigned short *__restrict__ y,
                                                  unsigned HW, unsigned C, const float *__restrict__ coef, int act, unsigned total_chunks)
{
    const unsigned nch = C / 8, per_img = HW * nch;
    for (unsigned q = blockIdx.x * 256u + threadIdx.x; q < total_chunks; q += gridDim.x * 256u) {
        const unsigned b = q / per_img;
        const unsigned c0 = (q % nch) * 8;
        float f[8];
        unpack8<T>(*reinterpret_cast<const uint4 *>(x + (size_t)q * 8), f);
        const float4 *cf = reinterpret_cast<const float4 *>(coef + ((size_t)b * C + c0) * 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 ab = cf[j];
            float v0 = f[2 * j] * ab.x + ab.y, v1 = f[2 * j + 1] * ab.z + ab.w;
            if (act) { v0 = silu(v0); v1 = silu(v1); }
            f[2 * j] = v0; f[2 * j + 1] = v1;
        }
        *reinterpret_cast<uint4 *>(y + (size_t)q * 8) = pack8<T>(f);
    }
}

// Small tensors (the 16x16 / 8x8 feature maps: a few MB, L2-resident): ONE launch, one workgroup per (batch, group).
// Pass 1: sums of (x - s) and (x - s)^2 about the group's first element s; pass 2 (re-read from L2): normalise, affine, SiLU.
// The three-kernel pipeline above is launch-latency bound there (3 x ~5 us for < 2 us of traffic).
template <class T>
__global__ __launch_bounds__(256) void k_gn_small(const unsigned short *__restrict__ x, unsigned short *__restrict__ y, int HW, int C,
                                                  int G, const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                                  int act)
{
    __shared__ float red[2][4];
    const int b = blockIdx.x / G, gi = blockIdx.x % G, cpg = C / G, hp = cpg / 2;     // hp: 4-byte pairs per pixel of this group
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const unsigned *xb = reinterpret_cast<const unsigned *>(x + ((size_t)b * HW * C + (size_t)gi * cpg));
    unsigned *yb = reinterpret_cast<unsigned *>(y + ((size_t)b * HW * C + (size_t)gi * cpg));
    const int n = HW * hp, rowp = C / 2;                                               // pairs in the group; pairs per pixel row
    const float sh = T::to_f((unsigned short)(xb[0] & 0xffff));
    float s1 = 0.f, s2 = 0.f;
    for (int i = tid; i < n; i += 256) {
        const int p = i / hp, c = i - p * hp;
        const unsigned u = xb[(size_t)p * rowp + c];
        const float a = T::to_f((unsigned short)(u & 0xffff)) - sh, d = T::to_f((unsigned short)(u >> 16)) - sh;
        s1 += a + d; s2 += a * a + d * d;
    }
    s1 = wave_sum_f(s1); s2 = wave_sum_f(s2);
    if (lane == 0) { red[0][wid] = s1; red[1][wid] = s2; }
    __syncthreads();
    const float cnt = (float)HW * (float)cpg;
    const float m1 = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / cnt;           // mean of (x - s)
    const float var = fmaxf((red[1][0] + red[1][1] + red[1][2] + red[1][3]) / cnt - m1 * m1, 0.f);
    const float mu = sh + m1, rstd = rsqrtf(var + eps);
    for (int i = tid; i < n; i += 256) {
        const int p = i / hp, c = i - p * hp;
        const unsigned u = xb[(size_t)p * rowp + c];
        const int ch = gi * cpg + 2 * c;
        float v0 = (T::to_f((unsigned short)(u & 0xffff)) - mu) * rstd * gamma[ch] + beta[ch];
        float v1 = (T::to_f((unsigned short)(u >> 16)) - mu) * rstd * gamma[ch + 1] + beta[ch + 1];
        if (act) { v0 = silu(v0); v1 = silu(v1); }
        yb[(size_t)p * rowp + c] = pack2<T>(v0, v1);
    }
}

// GroupNorm from per-(batch, group) sums that the PRODUCER of x accumulated (GEMM epilogue / k_concat_add_stats):
// stats[b][g] = (sum x, sum x^2) over the HW pixels and the C/G channels of group g.  ONE launch: every workgroup (blockIdx.y =
// batch) first turns the batch's G x 2 sums into per-channel coefficients a = rstd_g * gamma_c, d = beta_c - mean_g * a in LDS, then
// streams its share of the image: y = x * a + d [SiLU], 16 bytes per lane.  Replaces k_gn_partial + k_gn_finalize + k_gn_apply.
template <class T>
__global__ __launch_bounds__(256) void k_gn_apply_stats(const unsigned short *__restrict__ x, unsigned short *__restrict__ y, unsigned HW,
                                                        unsigned C, int G, const float *__restrict__ stats, const float *__restrict__ gamma,
                                                        const float *__restrict__ beta, float eps, int act)
{
    extern __shared__ float coef[];          // [C][2]
    const unsigned b = blockIdx.y, tid = threadIdx.x, cpg = C / G;
    const float *sb = stats + (size_t)b * G * 2;
    const float n = (float)HW * (float)cpg;
    for (unsigned c = tid; c < C; c += 256) {
        const unsigned g = c / cpg;
        const float mu = sb[2 * g] / n;
        const float rstd = rsqrtf(fmaxf(sb[2 * g + 1] / n - mu * mu, 0.f) + eps);
        const float a = rstd * gamma[c];
        coef[2 * c] = a; coef[2 * c + 1] = beta[c] - mu * a;
    }
    __syncthreads();
    const unsigned nch = C / 8, per_img = HW * nch;
    const unsigned short *xb = x + (size_t)b * per_img * 8;
    unsigned short *yb = y + (size_t)b * per_img * 8;
    for (unsigned q = blockIdx.x * 256u + tid; q < per_img; q += gridDim.x * 256u) {
        const unsigned c0 = (q % nch) * 8;
        float f[8];
        unpack8<T>(*reinterpret_cast<const uint4 *>(xb + (size_t)q * 8), f);
        const float4 *cf = reinterpret_cast<const float4 *>(coef + 2 * c0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 ab = cf[j];
            float v0 = f[2 * j] * ab.x + ab.y, v1 = f[2 * j + 1] * ab.z + ab.w;
            if (act) { v0 = silu(v0); v1 = silu(v1); }
            f[2 * j] = v0; f[2 * j + 1] = v1;
        }
        *reinterpret_cast<uint4 *>(yb + (size_t)q * 8) = pack8<T>(f);
    }
}

// GroupNorm whose statistics pass was done by the PRODUCER of x (gc_gemm_desc.out_chan_parts: the conv / linear epilogue, the split-K reduce
// kernel; gc_dn_concat_add_parts): parts[b][slab][g][half] = (sum x, sum x^2) over the rows of batch b inside row slab `slab` and over the
// channels of group g inside one column tile of the producer (half 1: the rest of a group that straddles two column tiles of `col_tile`
// channels).  Plain stores on the producer side -- no atomics, no zero-init; the few KB of a batch are added up HERE, in the prologue
// of the apply kernel (<= 8 independent 8-byte loads per thread), then the moments and the per-channel coefficients, all in LDS: ONE launch.
// slab_mode 0: slabs are the producer's row tiles of R rows counted over all B * HW rows (a tile that straddles two batches has a slab
// in each); slab_mode 1: slabs restart at every batch.  Raw (unshifted) fp32 sums: relative error of the variance ~ 1e-7 * mean^2 / var.
// (Tried first: per-CHANNEL partials, summed in this prologue -- every one of the ~500 workgroups re-read nslab x C x 8 bytes and re-added
// them in LDS: 64 us instead of 23 us at 64 x 64 x 320 -- and per-channel partials with a separate finalize launch: +1.8 % views/s.)
__device__ __forceinline__ int parts_count(unsigned b, unsigned HW, unsigned R, int mode)
{
    if (mode) return (int)((HW + R - 1) / R);
    const unsigned r0 = b * HW, r1 = r0 + HW - 1;          // B * HW < 2^31 (checked on the host)
    return (int)(r1 / R - r0 / R) + 1;
}

// group moments of batch b into LDS: gs[2 g] = mean_g, gs[2 g + 1] = rstd_g (ends with a barrier).  Thread = (entry e = (group, half) of a
// slab, slab subset): all of a thread's slab loads are in flight together, the subsets meet through plain LDS stores.  (LDS float atomics
// instead: 16 ds_add_f32 per thread cost 6.7 us of a 16 us kernel -- the LDS unit serialises them lane by lane.)
__device__ __forceinline__ void gparts_to_moments(float *gs, float2 *red /* [256] */, unsigned b, unsigned HW, unsigned C, unsigned G,
                                                  const float *__restrict__ parts, int nslab, unsigned R, int mode, unsigned col_tile, float eps)
{
    const unsigned tid = threadIdx.x, cpg = C / G;
    const unsigned E = 2 * G;                                   // entries per slab (G <= 128: E <= 256)
    const unsigned nsub = 256 / E, e = tid % E, sub = tid / E;
    const int ns = parts_count(b, HW, R, mode);
    const float2 *pb = reinterpret_cast<const float2 *>(parts) + (size_t)b * nslab * E + e;
    const unsigned g = e >> 1;
    // half 1 exists only for a group that straddles two column tiles of the producer
    const bool ok = sub < nsub && (!(e & 1) || (g * cpg) / col_tile != ((g + 1) * cpg - 1) / col_tile);
    float s1 = 0.f, s2 = 0.f;
    if (ok) {
        int sl = (int)sub;
        for (; sl + 3 * (int)nsub < ns; sl += 4 * (int)nsub) {
            const float2 v0 = pb[(size_t)sl * E], v1 = pb[(size_t)(sl + nsub) * E], v2 = pb[(size_t)(sl + 2 * nsub) * E], v3 = pb[(size_t)(sl + 3 * nsub) * E];
            s1 += (v0.x + v1.x) + (v2.x + v3.x); s2 += (v0.y + v1.y) + (v2.y + v3.y);
        }
        for (; sl < ns; sl += (int)nsub) { const float2 v = pb[(size_t)sl * E]; s1 += v.x; s2 += v.y; }
    }
    if (sub < nsub) red[sub * E + e] = make_float2(s1, s2);
    __syncthreads();
    const float n = (float)HW * (float)cpg;
    if (tid < G) {
        float t1 = 0.f, t2 = 0.f;
        for (unsigned k = 0; k < nsub; ++k) {
            const float2 a = red[k * E + 2 * tid], c = red[k * E + 2 * tid + 1];
            t1 += a.x + c.x; t2 += a.y + c.y;
        }
        const float mu = t1 / n;
        gs[2 * tid] = mu; gs[2 * tid + 1] = rsqrtf(fmaxf(t2 / n - mu * mu, 0.f) + eps);
    }
    __syncthreads();
}

// Apply pass with the statistics prologue.  A thread owns ONE 16-byte channel chunk and walks the pixels of its slab (slab plan of the
// statistics kernels): its 16 coefficients live in registers -- no coefficient table in the loop (the first version read a [C][2] LDS
// table per chunk: 64-byte lane stride = 16-way bank conflicts, 43 us against 12 us for the plain apply kernel at 64 x 64 x 320 / 640) --
// and gamma / beta and the first pixels of x are requested before the partials (nothing of that depends on the statistics).
template <class T, bool Q8 = false>
__global__ __launch_bounds__(256) void k_gn_apply_parts(const unsigned short *__restrict__ x, unsigned short *__restrict__ y, unsigned HW,
                                                        unsigned C, unsigned G, const float *__restrict__ parts, int nslab, unsigned R, int mode,
                                                        unsigned col_tile, const float *__restrict__ gamma, const float *__restrict__ beta, float eps, int act,
                                                        int nchb, int pix_per_block, unsigned Cp = 0, float qscale = 1.f)
{
    // Q8: e4m3 output y8[b][p][Cp] bytes (Cp = C rounded up to 128, padding channels written as zero by the thread of the last chunk),
    // stored = value * qscale saturated to +-448: the input of an fp8 convolution / linear (k_gn_apply_stats_fp8 with the producer's partials)
    __shared__ float gs[256];                // [G][2], G <= 128
    __shared__ float2 red[256];
    const unsigned b = blockIdx.z, tid = threadIdx.x, cpg = C / G;
    const int lanes = 256 / nchb;
    const int cch = tid % nchb, pl = tid / nchb;
    const unsigned c0 = (blockIdx.y * nchb + cch) * 8;
    const bool live = pl < lanes;
    const int p0 = blockIdx.x * pix_per_block, p1 = min(p0 + pix_per_block, (int)HW);
    const unsigned short *xb = x + (size_t)b * HW * C + c0;
    unsigned short *yb = y + (size_t)b * HW * C + c0;
    float4 gm[2], bt[2];
    uint4 pre[4];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        gm[k] = live ? *reinterpret_cast<const float4 *>(gamma + c0 + 4 * k) : make_float4(0.f, 0.f, 0.f, 0.f);
        bt[k] = live ? *reinterpret_cast<const float4 *>(beta + c0 + 4 * k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int p = p0 + pl + u * lanes;
        pre[u] = (live && p < p1) ? *reinterpret_cast<const uint4 *>(xb + (size_t)p * C) : make_uint4(0u, 0u, 0u, 0u);
    }
    gparts_to_moments(gs, red, b, HW, C, G, parts, nslab, R, mode, col_tile, eps);
    if (!live) return;
    float ca[8], cd[8];
    const float gmv[8] = {gm[0].x, gm[0].y, gm[0].z, gm[0].w, gm[1].x, gm[1].y, gm[1].z, gm[1].w};
    const float btv[8] = {bt[0].x, bt[0].y, bt[0].z, bt[0].w, bt[1].x, bt[1].y, bt[1].z, bt[1].w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const unsigned g = (c0 + j) / cpg;
        ca[j] = gs[2 * g + 1] * gmv[j];
        cd[j] = btv[j] - gs[2 * g] * ca[j];
    }
    auto one = [&](int p, const uint4 &raw) __attribute__((always_inline)) {
        float f[8];
        unpack8<T>(raw, f);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = f[j] * ca[j] + cd[j];
            f[j] = act ? silu(v) : v;
        }
        if constexpr (Q8) {
            unsigned char *o8 = reinterpret_cast<unsigned char *>(y) + ((size_t)b * HW + p) * Cp + c0;
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = fminf(fmaxf(f[j] * qscale, -448.f), 448.f);
            int w0 = 0, w1 = 0;
            w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], w0, false); w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w0, true);
            w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], w1, false); w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], w1, true);
            *reinterpret_cast<uint2 *>(o8) = make_uint2((unsigned)w0, (unsigned)w1);
            if (c0 + 8 == C)
                for (unsigned c = C; c < Cp; c += 8) *reinterpret_cast<uint2 *>(o8 + (c - c0)) = make_uint2(0u, 0u);
        } else {
            *reinterpret_cast<uint4 *>(yb + (size_t)p * C) = pack8<T>(f);
        }
    };
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int p = p0 + pl + u * lanes;
        if (p < p1) one(p, pre[u]);
    }
    for (int p = p0 + pl + 4 * lanes; p < p1; p += 4 * lanes) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int pp = p + u * lanes; v[u] = *reinterpret_cast<const uint4 *>(xb + (size_t)(pp < p1 ? pp : p) * C); }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int pp = p + u * lanes; if (pp < p1) one(pp, v[u]); }
    }
}

// the coefficients alone, [B][C][2] to global (input of the fused transformer head): one workgroup per batch
__global__ __launch_bounds__(256) void k_gn_coef_parts(unsigned HW, unsigned C, unsigned G, const float *__restrict__ parts, int nslab, unsigned R, int mode,
                                                       unsigned col_tile, const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                                       float *__restrict__ out)
{
    __shared__ float gs[256];
    __shared__ float2 red[256];
    const unsigned b = blockIdx.x, cpg = C / G;
    gparts_to_moments(gs, red, b, HW, C, G, parts, nslab, R, mode, col_tile, eps);
    for (unsigned c = threadIdx.x; c < C; c += 256) {
        const unsigned g = c / cpg;
        const float a = gs[2 * g + 1] * gamma[c];
        *reinterpret_cast<float2 *>(out + ((size_t)b * C + c) * 2) = make_float2(a, beta[c] - gs[2 * g] * a);
    }
}

// The same GroupNorm(+SiLU) with an OCP fp8 (e4m3) OUTPUT for the fp8 convolution path (k_gemm8q): y8[b][p][Cp] bytes, Cp = C rounded
// up to a multiple of 128 (one 3x3 tap per 128-byte k-tile), padding channels written as zero; stored value = y * qscale (a power of
// two: the tensor-wide E8M0 activation scale), saturated to +-448.
template <class T>
__global__ __launch_bounds__(256) void k_gn_apply_stats_fp8(const unsigned short *__restrict__ x, unsigned char *__restrict__ y8, unsigned HW,
                                                            unsigned C, unsigned Cp, int G, const float *__restrict__ stats,
                                                            const float *__restrict__ gamma, const float *__restrict__ beta, float eps, int act,
                                                            float qscale)
{
    extern __shared__ float coef[];          // [C][2]
    const unsigned b = blockIdx.y, tid = threadIdx.x, cpg = C / G;
    const float *sb = stats + (size_t)b * G * 2;
    const float n = (float)HW * (float)cpg;
    for (unsigned c = tid; c < C; c += 256) {
        const unsigned g = c / cpg;
        const float mu = sb[2 * g] / n;
        const float rstd = rsqrtf(fmaxf(sb[2 * g + 1] / n - mu * mu, 0.f) + eps);
        const float a = rstd * gamma[c];
        coef[2 * c] = a; coef[2 * c + 1] = beta[c] - mu * a;
    }
    __syncthreads();
    const unsigned nch = C / 8, nchp = Cp / 8, per_img = HW * nchp;
    const unsigned short *xb = x + (size_t)b * HW * C;
    unsigned char *yb = y8 + (size_t)b * HW * Cp;
    for (unsigned q = blockIdx.x * 256u + tid; q < per_img; q += gridDim.x * 256u) {
        const unsigned p = q / nchp, ch = q - p * nchp;
        uint2 o = make_uint2(0u, 0u);
        if (ch < nch) {
            const unsigned c0 = ch * 8;
            float f[8];
            unpack8<T>(*reinterpret_cast<const uint4 *>(xb + (size_t)p * C + c0), f);
            const float4 *cf = reinterpret_cast<const float4 *>(coef + 2 * c0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 ab = cf[j];
                float v0 = f[2 * j] * ab.x + ab.y, v1 = f[2 * j + 1] * ab.z + ab.w;
                if (act) { v0 = silu(v0); v1 = silu(v1); }
                f[2 * j] = fminf(fmaxf(v0 * qscale, -448.f), 448.f); f[2 * j + 1] = fminf(fmaxf(v1 * qscale, -448.f), 448.f);
            }
            int w0 = 0, w1 = 0;
            w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], w0, false); w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w0, true);
            w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], w1, false); w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], w1, true);
            o = make_uint2((unsigned)w0, (unsigned)w1);
        }
        *reinterpret_cast<uint2 *>(yb + (size_t)q * 8) = o;
    }
}

// out[b][p][C1+C2] = [a | b (+ c)] with the per-(batch, group) sums of the OUTPUT accumulated on the way (the input of the next
// GroupNorm): slab structure of k_gn_partial -- a thread owns one 16-byte channel chunk and walks the pixels of its slab, partial
// sums are combined over the block's pixel lanes in LDS, one atomic pair per channel per block.
template <class T>
__global__ __launch_bounds__(256) void k_concat_add_stats(const unsigned short *__restrict__ a, int C1, const unsigned short *__restrict__ bsrc,
                                                          const unsigned short *__restrict__ c, int C2, unsigned short *__restrict__ out,
                                                          int HW, int nchb, int pix_per_block, int G, float *__restrict__ stats)
{
    extern __shared__ float sp[];   // [lanes][nchb][16], then [G][2]
    const int b = blockIdx.z, C = C1 + C2;
    const int lanes = 256 / nchb;
    const int tid = threadIdx.x;
    const int cch = tid % nchb, pl = tid / nchb;
    const int c0 = (blockIdx.y * nchb + cch) * 8;
    const size_t row0 = (size_t)b * HW;
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    if (pl < lanes) {
        const int p0 = blockIdx.x * pix_per_block;
        const int p1 = min(p0 + pix_per_block, HW);
        for (int p = p0 + pl; p < p1; p += lanes) {
            const size_t m = row0 + p;
            uint4 v;
            float f[8];
            if (c0 < C1) { v = *reinterpret_cast<const uint4 *>(a + m * C1 + c0); unpack8<T>(v, f); }
            else {
                v = *reinterpret_cast<const uint4 *>(bsrc + m * C2 + (c0 - C1));
                unpack8<T>(v, f);
                if (c) {
                    float fc[8];
                    unpack8<T>(*reinterpret_cast<const uint4 *>(c + m * C2 + (c0 - C1)), fc);
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] += fc[j];
                    v = pack8<T>(f);
                    unpack8<T>(v, f);        // statistics of the values as stored
                }
            }
            if (out) *reinterpret_cast<uint4 *>(out + m * C + c0) = v;
#pragma unroll
            for (int j = 0; j < 8; ++j) { s1[j] += f[j]; s2[j] += f[j] * f[j]; }
        }
        float *o = sp + ((size_t)pl * nchb + cch) * 16;
#pragma unroll
        for (int j = 0; j < 8; ++j) { o[2 * j] = s1[j]; o[2 * j + 1] = s2[j]; }
    }
    __syncthreads();
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    if (pl == 0) {
        for (int l = 0; l < lanes; ++l) {
            const float *o = sp + ((size_t)l * nchb + cch) * 16;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] += o[j];
        }
    }
    __syncthreads();
    float *grp = sp;                 // [G][2] group sums of this block
    for (int i = tid; i < 2 * G; i += 256) grp[i] = 0.f;
    __syncthreads();
    if (pl == 0) {
        const int cpg = C / G;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int gi = (c0 + j) / cpg;
            __hip_atomic_fetch_add(grp + 2 * gi, acc[2 * j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(grp + 2 * gi + 1, acc[2 * j + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    for (int i = tid; i < 2 * G; i += 256) {
        const float v = grp[i];
        if (v != 0.f) unsafeAtomicAdd(stats + (size_t)b * G * 2 + i, v);
    }
}

// out[b][p][C1+C2] = [a | b (+ c)] + the per-CHANNEL partial sums of the output per pixel slab, plain stores (parts[b][slab][C][2],
// slab_mode 1 of gc_dn_groupnorm_apply_parts).  A thread owns one 16-byte channel chunk and walks the pixels of its slab FOUR at a time
// (independent loads in flight; the one-at-a-time walk of k_concat_add_stats over 128-pixel slabs ran 41 us against 9 us for the plain
// concat), slabs of 16 / 32 pixels so that the grid has several workgroups per CU.
template <class T>
__global__ __launch_bounds__(256) void k_concat_add_parts(const unsigned short *__restrict__ a, int C1, const unsigned short *__restrict__ bsrc,
                                                          const unsigned short *__restrict__ c, int C2, unsigned short *__restrict__ out,
                                                          int HW, int nchb, int pix_per_block, int cpg, float *__restrict__ parts)
{
    extern __shared__ float sp[];   // [lanes][nchb][16]
    const int b = blockIdx.z, C = C1 + C2;
    const int lanes = 256 / nchb;
    const int tid = threadIdx.x;
    const int cch = tid % nchb, pl = tid / nchb;
    const int c0 = (blockIdx.y * nchb + cch) * 8;
    const size_t row0 = (size_t)b * HW;
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    if (pl < lanes) {
        const int p0 = blockIdx.x * pix_per_block;
        const int p1 = min(p0 + pix_per_block, HW);
        const bool first = c0 < C1;
        const unsigned short *src = first ? a + c0 : bsrc + (c0 - C1);
        const unsigned short *src2 = (!first && c) ? c + (c0 - C1) : nullptr;
        const int ld = first ? C1 : C2;
        for (int p = p0 + pl; p < p1; p += 4 * lanes) {
            uint4 v[4], w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int pp = p + u * lanes;
                const size_t m = row0 + (pp < p1 ? pp : p);
                v[u] = *reinterpret_cast<const uint4 *>(src + m * ld);
                if (src2) w[u] = *reinterpret_cast<const uint4 *>(src2 + m * ld);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int pp = p + u * lanes;
                if (pp >= p1) break;
                float f[8];
                unpack8<T>(v[u], f);
                if (src2) {
                    float fc[8];
                    unpack8<T>(w[u], fc);
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] += fc[j];
                    v[u] = pack8<T>(f);
                    unpack8<T>(v[u], f);        // statistics of the values as stored
                }
                *reinterpret_cast<uint4 *>(out + (row0 + pp) * C + c0) = v[u];
#pragma unroll
                for (int j = 0; j < 8; ++j) { s1[j] += f[j]; s2[j] += f[j] * f[j]; }
            }
        }
        float *o = sp + ((size_t)pl * nchb + cch) * 16;
#pragma unroll
        for (int j = 0; j < 8; ++j) { o[2 * j] = s1[j]; o[2 * j + 1] = s2[j]; }
    }
    __syncthreads();
    if (pl == 0) {
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.f;
        for (int l = 0; l < lanes; ++l) {
            const float *o = sp + ((size_t)l * nchb + cch) * 16;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] += o[j];
        }
        float *o = sp + (size_t)cch * 16;          // lane 0's own slot: per-channel (sum, sum^2) of this slab for the group pass
#pragma unroll
        for (int j = 0; j < 16; ++j) o[j] = acc[j];
    }
    __syncthreads();
    // group g of this workgroup's channel slice = cpg consecutive channels (the slice holds whole groups: checked on the host) -> half 0
    const int gslice = nchb * 8 / cpg, G = C / cpg;
    if (tid < gslice) {
        float s1 = 0.f, s2 = 0.f;
        for (int cc = tid * cpg; cc < (tid + 1) * cpg; ++cc) { s1 += sp[2 * cc]; s2 += sp[2 * cc + 1]; }
        const int gg = blockIdx.y * gslice + tid;
        *reinterpret_cast<float2 *>(parts + (((((size_t)b * gridDim.x + blockIdx.x) * G + gg) * 2) * 2)) = make_float2(s1, s2);
    }
}

// ------------------------------------------------------------------------------------------ LayerNorm
// one wave64 per token row; exact two-pass in registers (C <= 64*8*4).
template <class T>
__global__ __launch_bounds__(256) void k_layernorm(const unsigned short *__restrict__ x, unsigned short *__restrict__ y,
                                                   int64_t M, int C, const float *__restrict__ gamma,
                                                   const float *__restrict__ beta, float eps)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nch = C / 8;
    float f[4][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
            unpack8<T>(*reinterpret_cast<const uint4 *>(x + row * C + ch * 8), f[i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += f[i][j];
        }
    }
    const float mean = wave_sum_f(s) / (float)C;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (lane + 64 * i < nch) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = f[i][j] - mean; v += d * d; }
        }
    const float rstd = rsqrtf(wave_sum_f(v) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (f[i][j] - mean) * rstd * gamma[ch * 8 + j] + beta[ch * 8 + j];
            *reinterpret_cast<uint4 *>(y + row * C + ch * 8) = pack8<T>(o);
        }
    }
}

// LayerNorm with an OCP fp8 (e4m3) output: same wave-per-row two-pass kernel, 8 bytes stored per 16-byte input chunk; stored value =
// y * qscale (power of two: the tensor-wide E8M0 activation scale of the fp8 GEMM that consumes it), saturated to +-448.
template <class T>
__global__ __launch_bounds__(256) void k_layernorm_fp8(const unsigned short *__restrict__ x, unsigned char *__restrict__ y8,
                                                       int64_t M, int C, const float *__restrict__ gamma,
                                                       const float *__restrict__ beta, float eps, float qscale)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nch = C / 8;
    float f[4][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
            unpack8<T>(*reinterpret_cast<const uint4 *>(x + row * C + ch * 8), f[i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += f[i][j];
        }
    }
    const float mean = wave_sum_f(s) / (float)C;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (lane + 64 * i < nch) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = f[i][j] - mean; v += d * d; }
        }
    const float rstd = rsqrtf(wave_sum_f(v) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                o[j] = fminf(fmaxf(((f[i][j] - mean) * rstd * gamma[ch * 8 + j] + beta[ch * 8 + j]) * qscale, -448.f), 448.f);
            int w0 = 0, w1 = 0;
            w0 = __builtin_amdgcn_cvt_pk_fp8_f32(o[0], o[1], w0, false); w0 = __builtin_amdgcn_cvt_pk_fp8_f32(o[2], o[3], w0, true);
            w1 = __builtin_amdgcn_cvt_pk_fp8_f32(o[4], o[5], w1, false); w1 = __builtin_amdgcn_cvt_pk_fp8_f32(o[6], o[7], w1, true);
            *reinterpret_cast<uint2 *>(y8 + row * C + ch * 8) = make_uint2((unsigned)w0, (unsigned)w1);
        }
    }
}

// ------------------------------------------------------------------------------------------ elementwise
// out[M, C1+C2] = [a[M,C1] | b[M,C2] (+ c[M,C2])]   (skip concat of the up blocks, with the ControlNet
// residual add of `down_block_res_samples` folded in)
template <class T>
__global__ __launch_bounds__(256) void k_concat_add(const unsigned short *__restrict__ a, int C1,
                                                    const unsigned short *__restrict__ b, const unsigned short *__restrict__ c,
                                                    int C2, unsigned short *__restrict__ out, unsigned total_chunks)
{
    const unsigned nch = (C1 + C2) / 8, n1 = C1 / 8;
    for (unsigned q = blockIdx.x * 256u + threadIdx.x; q < total_chunks; q += gridDim.x * 256u) {
        const size_t m = q / nch;
        const unsigned ch = q - (unsigned)m * nch;
        uint4 v;
        if (ch < n1) v = *reinterpret_cast<const uint4 *>(a + m * C1 + ch * 8);
        else {
            v = *reinterpret_cast<const uint4 *>(b + m * C2 + (ch - n1) * 8);
            if (c) {
                float fb[8], fc[8];
                unpack8<T>(v, fb);
                unpack8<T>(*reinterpret_cast<const uint4 *>(c + m * C2 + (ch - n1) * 8), fc);
#pragma unroll
                for (int j = 0; j < 8; ++j) fb[j] += fc[j];
                v = pack8<T>(fb);
            }
        }
        *reinterpret_cast<uint4 *>(out + m * (int64_t)(C1 + C2) + ch * 8) = v;
    }
}

// out = a*sa + b*sb (b may be null); act: 0 none, 1 silu
template <class T>
__global__ __launch_bounds__(256) void k_axpby(const unsigned short *__restrict__ a, float sa, const unsigned short *__restrict__ b,
                                               float sb, int act, unsigned short *__restrict__ out, int64_t total_chunks)
{
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total_chunks; q += (int64_t)gridDim.x * 256) {
        float fa[8], fb[8];
        unpack8<T>(*reinterpret_cast<const uint4 *>(a + q * 8), fa);
        if (b) unpack8<T>(*reinterpret_cast<const uint4 *>(b + q * 8), fb);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = fa[j] * sa + (b ? fb[j] * sb : 0.f);
            fa[j] = act ? silu(v) : v;
        }
        *reinterpret_cast<uint4 *>(out + q * 8) = pack8<T>(fa);
    }
}

// fp32 -> T with optional silu (time-embedding vectors)
template <class T>
__global__ __launch_bounds__(256) void k_cast_f32(const float *__restrict__ a, int act, unsigned short *__restrict__ out, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float v = a[i];
        out[i] = T::from_f(act ? silu(v) : v);
    }
}

// row softmax in place over [M, N] T (VAE mid-block attention scores), one workgroup per row
template <class T>
__global__ __launch_bounds__(256) void k_softmax_rows(unsigned short *__restrict__ s, int64_t N, int64_t ld, float scale)
{
    __shared__ float red[4];
    unsigned short *row = s + (int64_t)blockIdx.x * ld;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    float mx = -1e30f;
    for (int64_t i = tid; i < N; i += 256) mx = fmaxf(mx, T::to_f(row[i]) * scale);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
    if (lane == 0) red[wid] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int64_t i = tid; i < N; i += 256) sum += __expf(T::to_f(row[i]) * scale - mx);
    sum = wave_sum_f(sum);
    if (lane == 0) red[wid] = sum;
    __syncthreads();
    const float inv = 1.f / (red[0] + red[1] + red[2] + red[3]);
    for (int64_t i = tid; i < N; i += 256) row[i] = T::from_f(__expf(T::to_f(row[i]) * scale - mx) * inv);
}

// CFG combine + DDIM step fused with the re-packing of the next UNet input:
//   eps = eps_u + gs*(eps_c - eps_u); x0 = (x - sqrt(1-a_t) eps)/sqrt(a_t); x' = sqrt(a_p) x0 + sqrt(1-a_p) eps
// eps: fp32 [2f or f][HW][ldE] (conv_out, channels padded to ldE); lat: fp32 master [f][HW][4];
// xin: T [nrep*f][HW][8] (channels 4..7 stay zero) -- the `cat([latents]*2)` of the pipeline.
template <class T>
__global__ __launch_bounds__(256) void k_cfg_ddim(const float *__restrict__ eps, int ldE, int64_t f, int64_t HW, float gs,
                                                  int cfg, float c_x, float c_e, float *__restrict__ lat,
                                                  unsigned short *__restrict__ xin, int nrep)
{
    const int64_t n = f * HW;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float4 eu = *reinterpret_cast<const float4 *>(eps + i * ldE);
        float4 e = eu;
        if (cfg) {
            const float4 ec = *reinterpret_cast<const float4 *>(eps + (n + i) * ldE);
            e = make_float4(eu.x + gs * (ec.x - eu.x), eu.y + gs * (ec.y - eu.y), eu.z + gs * (ec.z - eu.z), eu.w + gs * (ec.w - eu.w));
        }
        float4 x = *reinterpret_cast<float4 *>(lat + i * 4);
        // x' = c_x * x + c_e * eps  with c_x = sqrt(a_p/a_t), c_e = sqrt(1-a_p) - sqrt(a_p (1-a_t)/a_t)
        x = make_float4(c_x * x.x + c_e * e.x, c_x * x.y + c_e * e.y, c_x * x.z + c_e * e.z, c_x * x.w + c_e * e.w);
        *reinterpret_cast<float4 *>(lat + i * 4) = x;
        const uint4 p = make_uint4(pack2<T>(x.x, x.y), pack2<T>(x.z, x.w), 0u, 0u);
        for (int r = 0; r < nrep; ++r) *reinterpret_cast<uint4 *>(xin + (r * n + i) * 8) = p;
    }
}

__global__ __launch_bounds__(256) void k_disp_max(const float *__restrict__ depth, int64_t n, unsigned *__restrict__ mx)
{
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, 1.f / (depth[i] + 1e-5f));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(mx, __float_as_uint(m));     // disparity > 0: uint order == float order
}

template <class T>
__global__ __launch_bounds__(256) void k_disp_write(const float *__restrict__ depth, int64_t n, const unsigned *__restrict__ mx,
                                                    unsigned short *__restrict__ out)
{
    const float inv = 1.f / __uint_as_float(*mx);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float d = (1.f / (depth[i] + 1e-5f)) * inv;
        const unsigned h = T::from_f(d);
        *reinterpret_cast<uint4 *>(out + i * 8) = make_uint4(h | (h << 16), h, 0u, 0u);
    }
}

__global__ __launch_bounds__(256) void k_mask_composite(const float *__restrict__ e, int ld_e, const float *__restrict__ u,
                                                        const float *__restrict__ m, float *__restrict__ out, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float w = m ? m[i] : 1.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) out[3 * i + c] = e[i * ld_e + c] * w + (m ? u[3 * i + c] * (1.f - w) : 0.f);
    }
}

// slab plan of the GroupNorm statistics pass: channel slices of <= 256 16-byte chunks, pixel slabs sized for ~1-2k workgroups
inline void gn_plan(int64_t B, int64_t HW, int C, int *nslab, int *ppb, int *ny, int *nchb)
{
    const int nch = C / 8;
    int y = 1;
    while (nch % y != 0 || nch / y > 256) ++y;
    *ny = y; *nchb = nch / y;
    const int lanes = 256 / *nchb;
    // at most 32 slabs per image (keeps the finalize pass tiny), each at least `lanes` pixels
    int p = (int)((HW + 31) / 32);
    if (p < lanes) p = lanes;
    (void)B;
    *ppb = p; *nslab = (int)((HW + p - 1) / p);
}

inline unsigned ew_grid(int64_t items) { return (unsigned)std::min<int64_t>((items + 255) / 256, 256 * 8); }

}  // namespace

#define DN_DISPATCH(dtype, CALL_BF16, CALL_F16)                 \
    do {                                                       \
        if ((dtype) == DT_BF16) { CALL_BF16; }                 \
        else if ((dtype) == DT_F16) { CALL_F16; }              \
        else { gc::set_error("%s: bad dtype", __func__); return GC_EINVAL; } \
    } while (0)

extern "C" {

size_t gc_dn_groupnorm_workspace_bytes(int64_t B, int64_t HW, int C)
{
    int nslab, ppb, ny, nchb;
    gn_plan(B, HW, C, &nslab, &ppb, &ny, &nchb);
    return sizeof(float) * 2 * (size_t)B * (size_t)C * (size_t)(nslab + 1);
}

int gc_dn_groupnorm(int dtype, const void *x, void *y, int64_t B, int64_t HW, int C, int G, const float *gamma,
                    const float *beta, float eps, int act, float *stats_ws, void *stream)
{
    GC_REQUIRE(C % 8 == 0 && C % G == 0 && stats_ws, "groupnorm: C must be a multiple of 8 and of G; workspace required");
    GC_REQUIRE(C / G <= 256 && B * HW * (C / 8) < (int64_t)1 << 31, "groupnorm: group too wide / tensor too large");
    hipStream_t s = gc::S(stream);
    // act bit 8: batch-invariant planning -- the one-launch form (other summation order) is chosen from the per-frame size only
    const int64_t Bsel = (act & 0x100) ? 1 : B;
    act &= 0xff;
    if ((C / G) % 2 == 0 && Bsel * HW * (int64_t)C <= ((int64_t)1 << 20) && Bsel * G >= 64) {      // <= 2 MB (the 8x8 maps): one launch, 5 vs 12 us
        DN_DISPATCH(dtype,
                    hipLaunchKernelGGL((k_gn_small<BF16>), dim3((unsigned)(B * G)), dim3(256), 0, s, (const unsigned short *)x,
                                       (unsigned short *)y, (int)HW, C, G, gamma, beta, eps, act),
                    hipLaunchKernelGGL((k_gn_small<F16>), dim3((unsigned)(B * G)), dim3(256), 0, s, (const unsigned short *)x,
                                       (unsigned short *)y, (int)HW, C, G, gamma, beta, eps, act));
        return gc::check_launch("gc_dn_groupnorm");
    }
    int nslab, ppb, ny, nchb;
    gn_plan(B, HW, C, &nslab, &ppb, &ny, &nchb);
    float *part = stats_ws, *coef = stats_ws + 2 * (size_t)B * C * nslab;
    const int lanes = 256 / nchb;
    dim3 grid((unsigned)nslab, ny, (unsigned)B);
    const size_t lds = sizeof(float) * 16 * (size_t)lanes * nchb;
    DN_DISPATCH(dtype,
                hipLaunchKernelGGL((k_gn_partial<BF16>), grid, dim3(256), lds, s, (const unsigned short *)x, (int)HW, C, nchb, ppb, part),
                hipLaunchKernelGGL((k_gn_partial<F16>), grid, dim3(256), lds, s, (const unsigned short *)x, (int)HW, C, nchb, ppb, part));
    DN_DISPATCH(dtype,
                hipLaunchKernelGGL((k_gn_finalize<BF16>), dim3((unsigned)(B * G)), dim3(256), 0, s, (const unsigned short *)x, (int)HW, C, G, nslab, part, gamma, beta, eps, coef),
                hipLaunchKernelGGL((k_gn_finalize<F16>), dim3((unsigned)(B * G)), dim3(256), 0, s, (const unsigned short *)x, (int)HW, C, G, nslab, part, gamma, beta, eps, coef));
    const int64_t chunks = B * HW * (C / 8);
    DN_DISPATCH(dtype,
                hipLaunchKernelGGL((k_gn_apply<BF16>), dim3(ew_grid(chunks)), dim3(256), 0, s, (const unsigned short *)x,
                                   (unsigned short *)y, (unsigned)HW, (unsigned)C, coef, act, (unsigned)chunks),
                hipLaunchKernelGGL((k_gn_apply<F16>), dim3(ew_grid(chunks)), dim3(256), 0, s, (const unsigned short *)x,
                                   (unsigned short *)y, (unsigned)HW, (unsigned)C, coef, act, (unsigned)chunks));
    return gc::check_launch("gc_dn_groupnorm");
}

int gc_dn_groupnorm_coef(int dtype, const void *x, int64_t B, int64_t HW, int C, int G, const float *gamma, const float *beta, float eps,
                         float *stats_ws, float *coef, void *stream)
{
    GC_REQUIRE(C % 8 == 0 && C % G == 0 && stats_ws && coef, "groupnorm_coef: C must be a multiple of 8 and of G; workspace and output required");
    GC_REQUIRE(C / G <= 256 && B * HW * (C / 8) < (int64_t)1 << 31, "groupnorm_coef: group too wide / tensor too large");
    hipStream_t s = gc::S(stream);
    int nslab, ppb, ny, nchb;
    gn_plan(B, HW, C, &nslab, &ppb, &ny, &nchb);
    float *part = stats_ws;
    const int lanes = 256 / nchb;
    dim3 grid((unsigned)nslab, ny, (unsigned)B);
    const size_t lds = sizeof(float) * 16 * (size_t)lanes * nchb;
    DN_DISPATCH(dtype,
                hipLaunchKernelGGL((k_gn_partial<BF16>), grid, dim3(256), lds, s, (const unsigned short *)x, (int)HW, C, nchb, ppb, part),
                hipLaunchKernelGGL((k_gn_partial<F16>), grid, dim3(256), lds, s, (const unsigned short *)x, (int)HW, C, nchb, ppb, part));
    DN_DISPATCH(dtype,
                hipLaunchKernelGGL((k_gn_finalize<BF16>), dim3((unsigned)(B * G)), dim3(256), 0, s, (const unsigned short *)x, (int)HW, C, G, nslab, part, gamma, beta, eps, coef),
                hipLaunchKernelGGL((k_gn_finalize<F16>), dim3((unsigned)(B * G)), dim3(256), 0, s, (const unsigned short *)x, (int)HW, C, G, nslab, part, gamma, beta, eps, coef));
    return gc::check_launch("gc_dn_groupnorm_coef");
}

int gc_dn_groupnorm_apply(int dtype, const void *x, void *y, int64_t B, int64_t HW, int C, int G, const float *gamma,
                          const float *beta, float eps, int act, const float *group_stats, void *stream)
{
    const float *chan_stats = group_stats;
    GC_REQUIRE(C % 8 == 0 && C % G == 0 && chan_stats && gamma && beta, "groupnorm_apply: C must be a multiple of 8 and of G; statistics required");
    GC_REQUIRE(HW * (int64_t)(C / 8) < (int64_t)1 << 31 && B <= 65535, "groupnorm_apply: tensor too large");
    const int64_t per_img = HW * (C / 8);
    // enough workgroups per image to fill the chip, few enough that the coefficient prologue (a few KB from L2 per workgroup) stays small
    int64_t gx = (per_img + 255) / 256;
    const int64_t cap = std::max<int64_t>(1, (256 * 4 + B - 1) / B);
    if (gx > cap) gx = cap;
    const size_t lds = sizeof(float) * 2 * (size_t)C;
    dim3 grid((unsigned)gx, (unsigned)B);
    DN_DISPATCH(dtype,
                hipLaunchKernelGGL((k_gn_apply_stats<BF16>), grid, dim3(256), lds, gc::S(stream), (const unsigned short *)x, (unsigned short *)y,
                                   (unsigned)HW, (unsigned)C, G, chan_stats, gamma, beta, eps, act),
                hipLaunchKernelGGL((k_gn_apply_stats<F16>), grid, dim3(256), lds, gc::S(stream), (const unsigned short *)x, (unsigned short *)y,
                                   (unsigned)HW, (unsigned)C, G, chan_stats, gamma, beta, eps, act));
    return gc::check_launch("gc_dn_groupnorm_apply");
}

int gc_dn_groupnorm_apply_fp8(int dtype, const void *x, void *y8, int64_t B, int64_t HW, int C, int C_padded, int G, const float *gamma,
                              const float *beta, float eps, int act, const float *group_stats, int a_scale, void *stream)
{
    GC_REQUIRE(C % 8 == 0 && C % G == 0 && group_stats && gamma && beta, "groupnorm_apply_fp8: C must be a multiple of 8 and of G; statistics required");
    GC_REQUIRE(C_padded >= C && C_padded % 128 == 0, "groupnorm_apply_fp8: padded channel count must be a multiple of 128");
    GC_REQUIRE(HW * (int64_t)(C_padded / 8) < (int64_t)1 << 31 && B <= 65535 && a_scale > 0 && a_scale < 255, "groupnorm_apply_fp8: bad size / scale");
    const int64_t per_img = HW * (C_padded / 8);
    int64_t gx = (per_img + 255) / 256;
    const int64_t cap = std::max<int64_t>(1, (256 * 4 + B - 1) / B);
    if (gx > cap) gx = cap;
    const size_t lds = sizeof(float) * 2 * (size_t)C;
    const float qscale = exp2f((float)(127 - a_scale));          // stored = value * 2^(127 - byte); the MFMA multiplies by 2^(byte - 127)
    dim3 grid((unsigned)gx, (unsigned)B);
    DN_DISPATCH(dtype,
                hipLaunchKernelGGL((k_gn_apply_stats_fp8<BF16>), grid, dim3(256), lds, gc::S(stream), (const unsigned short *)x, (unsigned char *)y8,
                                   (unsigned)HW, (unsigned)C, (unsigned)C_padded, G, group_stats, gamma, beta, eps, act, qscale),
                hipLaunchKernelGGL((k_gn_apply_stats_fp8<F16>), grid, dim3(256), lds, gc::S(stream), (const unsigned short *)x, (unsigned char *)y8,
                                   (unsigned)HW, (unsigned)C, (unsigned)C_padded, G, group_stats, gamma, beta, eps, act, qscale));
    return gc::check_launch("gc_dn_groupnorm_apply_fp8");
}

int gc_dn_layernorm(int dtype, const void *x, void *y, int64_t M, int C, const float *gamma, const float *beta,
                    float eps, void *stream)
{
    GC_REQUIRE(C % 8 == 0 && C <= 2048, "layernorm: C must be a multiple of 8 and <= 2048");
    dim3 grid((unsigned)((M + 3) / 4));
    DN_DISPATCH(dtype,
                hipLaunchKernelGGL((k_layernorm<BF16>), grid, dim3(256), 0, gc::S(stream), (const unsigned short *)x, (unsigned short *)y, M, C, gamma, beta, eps),
                hipLaunchKernelGGL((k_layernorm<F16>), grid, dim3(256), 0, gc::S(stream), (const unsigned short *)x, (unsigned short *)y, M, C, gamma, beta, eps));
    return gc::check_launch("gc_dn_layernorm");
}

int gc_dn_layernorm_fp8(int dtype, const void *x, void *y8, int64_t M, int C, const float *gamma, const float *beta,
                        float eps, int a_scale, void *stream)
{
    GC_REQUIRE(C % 16 == 0 && C <= 2048, "layernorm_fp8: C must be a multiple of 16 and <= 2048");
    GC_REQUIRE(a_scale > 0 && a_scale < 255 && M > 0 && x && y8 && gamma && beta, "layernorm_fp8: bad scale / null argument");
    const float qscale = exp2f((float)(127 - a_scale));
    dim3 grid((unsigned)((M + 3) / 4));
    DN_DISPATCH(dtype,
                hipLaunchKernelGGL((k_layernorm_fp8<BF16>), grid, dim3(256), 0, gc::S(stream), (const unsigned short *)x, (unsigned char *)y8, M, C, gamma, beta, eps, qscale),
                hipLaunchKernelGGL((k_layernorm_fp8<F16>), grid, dim3(256), 0, gc::S(stream), (const unsigned short *)x, (unsigned char *)y8, M, C, gamma, beta, eps, qscale));
    return gc::check_launch("gc_dn_layernorm_fp8");
}

int gc_dn_concat_add(int dtype, const void *a, int C1, const void *b, const void *c, int C2, void *out, int64_t M, int64_t rows_per_batch,
                     float *group_stats, int gn_groups, void *stream)
{
    GC_REQUIRE(C1 % 8 == 0 && C2 % 8 == 0, "concat: channel counts must be multiples of 8");
    float *chan_stats = group_stats;
    if (chan_stats) {      // copy + per-(batch, group) sums of the output, added into the caller-zeroed group_stats[B][gn_groups][2]
        GC_REQUIRE(rows_per_batch > 0 && M % rows_per_batch == 0 && M / rows_per_batch <= 65535, "concat: M must be B * rows_per_batch");
        GC_REQUIRE(gn_groups >= 1 && gn_groups <= 64 && (C1 + C2) % gn_groups == 0, "concat: gn_groups must divide C1 + C2");
        int nslab, ppb, ny, nchb;
        gn_plan(M / rows_per_batch, rows_per_batch, C1 + C2, &nslab, &ppb, &ny, &nchb);
        const int lanes = 256 / nchb;
        dim3 grid((unsigned)nslab, ny, (unsigned)(M / rows_per_batch));
        const size_t lds = sizeof(float) * 16 * (size_t)lanes * nchb;
        DN_DISPATCH(dtype,
                    hipLaunchKernelGGL((k_concat_add_stats<BF16>), grid, dim3(256), lds, gc::S(stream), (const unsigned short *)a, C1,
                                       (const unsigned short *)b, (const unsigned short *)c, C2, (unsigned short *)out, (int)rows_per_batch, nchb, ppb, gn_groups, chan_stats),
                    hipLaunchKernelGGL((k_concat_add_stats<F16>), grid, dim3(256), lds, gc::S(stream), (const unsigned short *)a, C1,
                                       (const unsigned short *)b, (const unsigned short *)c, C2, (unsigned short *)out, (int)rows_per_batch, nchb, ppb, gn_groups, chan_stats));
        return gc::check_launch("gc_dn_concat_add");
    }
    const int64_t chunks = M * ((C1 + C2) / 8);
    DN_DISPATCH(dtype,
                hipLaunchKernelGGL((k_concat_add<BF16>), dim3(ew_grid(chunks)), dim3(256), 0, gc::S(stream), (const unsigned short *)a, C1,
                                   (const unsigned short *)b, (const unsigned short *)c, C2, (unsigned short *)out, (unsigned)chunks),
                hipLaunchKernelGGL((k_concat_add<F16>), dim3(ew_grid(chunks)), dim3(256), 0, gc::S(stream), (const unsigned short *)a, C1,
                                   (const unsigned short *)b, (const unsigned short *)c, C2, (unsigned short *)out, (unsigned)chunks));
    return gc::check_launch("gc_dn_concat_add");
}

int gc_dn_groupnorm_coef_parts(int64_t B, int64_t HW, int C, int G, const float *gamma, const float *beta, float eps, const float *parts,
                               int64_t rows_per_slab, int nslab, int slab_mode, int col_tile, float *coef, void *stream)
{
    GC_REQUIRE(C % G == 0 && G <= 128 && parts && gamma && beta && coef && rows_per_slab > 0 && nslab > 0 && col_tile > 0 && B * HW < (int64_t)1 << 31,
               "groupnorm_coef_parts: bad arguments");
    hipLaunchKernelGGL(k_gn_coef_parts, dim3((unsigned)B), dim3(256), 0, gc::S(stream), (unsigned)HW, (unsigned)C, (unsigned)G, parts, nslab,
                       (unsigned)rows_per_slab, slab_mode, (unsigned)col_tile, gamma, beta, eps, coef);
    return gc::check_launch("gc_dn_groupnorm_coef_parts");
}

static void concat_parts_plan(int64_t HW, int C, int *nslab, int *ppb, int *ny, int *nchb);

int gc_dn_groupnorm_apply_parts(int dtype, const void *x, void *y, int64_t B, int64_t HW, int C, int G, const float *gamma, const float *beta,
                                float eps, int act, const float *parts, int64_t rows_per_slab, int nslab, int slab_mode, int col_tile, void *stream)
{
    GC_REQUIRE(C % 8 == 0 && C % G == 0 && G <= 128 && parts && gamma && beta, "groupnorm_apply_parts: C must be a multiple of 8 and of G; partials required");
    GC_REQUIRE(HW * (int64_t)C < (int64_t)1 << 31 && B * HW < (int64_t)1 << 31 && B <= 65535 && rows_per_slab > 0 && nslab > 0 && col_tile > 0,
               "groupnorm_apply_parts: bad sizes");
    int ns, ppb, ny, nchb;                      // pixel slabs of 16 / 32 pixels x channel slices of <= 256 chunks (the plan of gc_dn_concat_add_parts)
    concat_parts_plan(HW, C, &ns, &ppb, &ny, &nchb);
    dim3 grid((unsigned)ns, ny, (unsigned)B);
    DN_DISPATCH(dtype,
                hipLaunchKernelGGL((k_gn_apply_parts<BF16>), grid, dim3(256), 0, gc::S(stream), (const unsigned short *)x, (unsigned short *)y,
                                   (unsigned)HW, (unsigned)C, (unsigned)G, parts, nslab, (unsigned)rows_per_slab, slab_mode, (unsigned)col_tile, gamma, beta, eps, act, nchb, ppb),
                hipLaunchKernelGGL((k_gn_apply_parts<F16>), grid, dim3(256), 0, gc::S(stream), (const unsigned short *)x, (unsigned short *)y,
                                   (unsigned)HW, (unsigned)C, (unsigned)G, parts, nslab, (unsigned)rows_per_slab, slab_mode, (unsigned)col_tile, gamma, beta, eps, act, nchb, ppb));
    return gc::check_launch("gc_dn_groupnorm_apply_parts");
}

int gc_dn_groupnorm_apply_parts_fp8(int dtype, const void *x, void *y8, int64_t B, int64_t HW, int C, int C_padded, int G, const float *gamma,
                                    const float *beta, float eps, int act, const float *parts, int64_t rows_per_slab, int nslab, int slab_mode,
                                    int col_tile, int a_scale, void *stream)
{
    GC_REQUIRE(C % 8 == 0 && C % G == 0 && G <= 128 && parts && gamma && beta, "groupnorm_apply_parts_fp8: C must be a multiple of 8 and of G; partials required");
    GC_REQUIRE(C_padded >= C && C_padded % 128 == 0 && a_scale > 0 && a_scale < 255, "groupnorm_apply_parts_fp8: padded channel count % 128 == 0; E8M0 scale byte");
    GC_REQUIRE(HW * (int64_t)C_padded < (int64_t)1 << 31 && B * HW < (int64_t)1 << 31 && B <= 65535 && rows_per_slab > 0 && nslab > 0 && col_tile > 0,
               "groupnorm_apply_parts_fp8: bad sizes");
    int ns, ppb, ny, nchb;
    concat_parts_plan(HW, C, &ns, &ppb, &ny, &nchb);
    dim3 grid((unsigned)ns, ny, (unsigned)B);
    const float qscale = exp2f((float)(127 - a_scale));
    DN_DISPATCH(dtype,
                hipLaunchKernelGGL((k_gn_apply_parts<BF16, true>), grid, dim3(256), 0, gc::S(stream), (const unsigned short *)x, (unsigned short *)y8,
                                   (unsigned)HW, (unsigned)C, (unsigned)G, parts, nslab, (unsigned)rows_per_slab, slab_mode, (unsigned)col_tile, gamma, beta, eps, act, nchb, ppb,
                                   (unsigned)C_padded, qscale),
                hipLaunchKernelGGL((k_gn_apply_parts<F16, true>), grid, dim3(256), 0, gc::S(stream), (const unsigned short *)x, (unsigned short *)y8,
                                   (unsigned)HW, (unsigned)C, (unsigned)G, parts, nslab, (unsigned)rows_per_slab, slab_mode, (unsigned)col_tile, gamma, beta, eps, act, nchb, ppb,
                                   (unsigned)C_padded, qscale));
    return gc::check_launch("gc_dn_groupnorm_apply_parts_fp8");
}

static void concat_parts_plan(int64_t HW, int C, int *nslab, int *ppb, int *ny, int *nchb)
{
    const int nch = C / 8;
    int y = 1;
    while (nch % y != 0 || nch / y > 256) ++y;
    *ny = y; *nchb = nch / y;
    const int lanes = 256 / *nchb;
    int p = HW >= 4096 ? 32 : 16;
    if (p < lanes) p = lanes;
    *ppb = p; *nslab = (int)((HW + p - 1) / p);
}

int gc_dn_concat_parts_layout(int64_t rows_per_batch, int C, int gn_groups, int64_t *rows_per_slab, int *nslab, int *col_tile)
{
    GC_REQUIRE(rows_per_slab && nslab && col_tile && rows_per_batch > 0 && C % 8 == 0 && gn_groups >= 1, "concat_parts_layout: bad arguments");
    *rows_per_slab = 0; *nslab = 0; *col_tile = 0;
    int ns, ppb, ny, nchb;
    concat_parts_plan(rows_per_batch, C, &ns, &ppb, &ny, &nchb);
    if (C % gn_groups != 0 || (nchb * 8) % (C / gn_groups) != 0 || nchb * 8 / (C / gn_groups) > 256) return GC_OK;     // a channel slice must hold whole groups
    *rows_per_slab = ppb; *nslab = ns; *col_tile = nchb * 8;
    return GC_OK;
}

int gc_dn_concat_add_parts(int dtype, const void *a, int C1, const void *b, const void *c, int C2, void *out, int64_t M, int64_t rows_per_batch,
                           int gn_groups, float *parts, void *stream)
{
    GC_REQUIRE(gn_groups >= 1 && (C1 + C2) % gn_groups == 0, "concat_parts: gn_groups must divide C1 + C2");
    const int cpg = (C1 + C2) / gn_groups;
    GC_REQUIRE(C1 % 8 == 0 && C2 % 8 == 0 && parts && out, "concat_parts: channel counts must be multiples of 8; output and partials buffer required");
    GC_REQUIRE(rows_per_batch > 0 && M % rows_per_batch == 0 && M / rows_per_batch <= 65535, "concat_parts: M must be B * rows_per_batch");
    int nslab, ppb, ny, nchb;
    concat_parts_plan(rows_per_batch, C1 + C2, &nslab, &ppb, &ny, &nchb);
    GC_REQUIRE((nchb * 8) % cpg == 0 && nchb * 8 / cpg <= 256, "concat_parts: a channel slice must hold whole groups (see gc_dn_concat_parts_layout)");
    const int lanes = 256 / nchb;
    dim3 grid((unsigned)nslab, ny, (unsigned)(M / rows_per_batch));
    const size_t lds = sizeof(float) * 16 * (size_t)lanes * nchb;
    DN_DISPATCH(dtype,
                hipLaunchKernelGGL((k_concat_add_parts<BF16>), grid, dim3(256), lds, gc::S(stream), (const unsigned short *)a, C1,
                                   (const unsigned short *)b, (const unsigned short *)c, C2, (unsigned short *)out, (int)rows_per_batch, nchb, ppb, cpg, parts),
                hipLaunchKernelGGL((k_concat_add_parts<F16>), grid, dim3(256), lds, gc::S(stream), (const unsigned short *)a, C1,
                                   (const unsigned short *)b, (const unsigned short *)c, C2, (unsigned short *)out, (int)rows_per_batch, nchb, ppb, cpg, parts));
    return gc::check_launch("gc_dn_concat_add_parts");
}

int gc_dn_group_stats(int dtype, const void *x, int64_t B, int64_t HW, int C, int G, float *group_stats, void *stream)
{
    GC_REQUIRE(C % 8 == 0 && G >= 1 && G <= 64 && C % G == 0 && group_stats && B <= 65535, "group_stats: C % 8 == 0, G | C, statistics buffer required");
    int nslab, ppb, ny, nchb;
    gn_plan(B, HW, C, &nslab, &ppb, &ny, &nchb);
    const int lanes = 256 / nchb;
    dim3 grid((unsigned)nslab, ny, (unsigned)B);
    const size_t lds = sizeof(float) * 16 * (size_t)lanes * nchb;
    DN_DISPATCH(dtype,
                hipLaunchKernelGGL((k_concat_add_stats<BF16>), grid, dim3(256), lds, gc::S(stream), (const unsigned short *)x, C, (const unsigned short *)nullptr,
                                   (const unsigned short *)nullptr, 0, (unsigned short *)nullptr, (int)HW, nchb, ppb, G, group_stats),
                hipLaunchKernelGGL((k_concat_add_stats<F16>), grid, dim3(256), lds, gc::S(stream), (const unsigned short *)x, C, (const unsigned short *)nullptr,
                                   (const unsigned short *)nullptr, 0, (unsigned short *)nullptr, (int)HW, nchb, ppb, G, group_stats));
    return gc::check_launch("gc_dn_group_stats");
}

int gc_dn_axpby(int dtype, const void *a, float sa, const void *b, float sb, int act, void *out, int64_t n, void *stream)
{
    GC_REQUIRE(n % 8 == 0, "axpby: element count must be a multiple of 8");
    const int64_t chunks = n / 8;
    DN_DISPATCH(dtype,
                hipLaunchKernelGGL((k_axpby<BF16>), dim3(ew_grid(chunks)), dim3(256), 0, gc::S(stream), (const unsigned short *)a, sa,
                                   (const unsigned short *)b, sb, act, (unsigned short *)out, chunks),
                hipLaunchKernelGGL((k_axpby<F16>), dim3(ew_grid(chunks)), dim3(256), 0, gc::S(stream), (const unsigned short *)a, sa,
                                   (const unsigned short *)b, sb, act, (unsigned short *)out, chunks));
    return gc::check_launch("gc_dn_axpby");
}

int gc_dn_cast_f32(int dtype, const float *a, int act, void *out, int64_t n, void *stream)
{
    DN_DISPATCH(dtype,
                hipLaunchKernelGGL((k_cast_f32<BF16>), dim3(ew_grid(n)), dim3(256), 0, gc::S(stream), a, act, (unsigned short *)out, n),
                hipLaunchKernelGGL((k_cast_f32<F16>), dim3(ew_grid(n)), dim3(256), 0, gc::S(stream), a, act, (unsigned short *)out, n));
    return gc::check_launch("gc_dn_cast_f32");
}

int gc_dn_softmax_rows(int dtype, void *s, int64_t M, int64_t N, int64_t ld, float scale, void *stream)
{
    DN_DISPATCH(dtype,
                hipLaunchKernelGGL((k_softmax_rows<BF16>), dim3((unsigned)M), dim3(256), 0, gc::S(stream), (unsigned short *)s, N, ld, scale),
                hipLaunchKernelGGL((k_softmax_rows<F16>), dim3((unsigned)M), dim3(256), 0, gc::S(stream), (unsigned short *)s, N, ld, scale));
    return gc::check_launch("gc_dn_softmax_rows");
}

int gc_dn_depth_to_disparity(int dtype, const float *depth, int64_t HW, void *out, unsigned *max_ws, void *stream)
{
    hipStream_t s = gc::S(stream);
    if (hipMemsetAsync(max_ws, 0, 4, s) != hipSuccess) return GC_ELAUNCH;
    // 64 workgroups: the maximum ends in ONE word, and 4096 waves doing an atomicMax on it cost 48 us (~12 ns each), 256 cost ~3 us
    hipLaunchKernelGGL(k_disp_max, dim3(std::min<unsigned>(ew_grid(HW), 64u)), dim3(256), 0, s, depth, HW, max_ws);
    DN_DISPATCH(dtype,
                hipLaunchKernelGGL((k_disp_write<BF16>), dim3(ew_grid(HW)), dim3(256), 0, s, depth, HW, max_ws, (unsigned short *)out),
                hipLaunchKernelGGL((k_disp_write<F16>), dim3(ew_grid(HW)), dim3(256), 0, s, depth, HW, max_ws, (unsigned short *)out));
    return gc::check_launch("gc_dn_depth_to_disparity");
}

int gc_dn_mask_composite(const float *edited, int ld_e, const float *unedited, const float *mask, float *out, int64_t HW,
                         void *stream)
{
    GC_REQUIRE(edited && out && (mask == nullptr || unedited != nullptr), "mask_composite: bad arguments");
    hipLaunchKernelGGL(k_mask_composite, dim3(ew_grid(HW)), dim3(256), 0, gc::S(stream), edited, ld_e, unedited, mask, out, HW);
    return gc::check_launch("gc_dn_mask_composite");
}

int gc_dn_cfg_ddim_step(int dtype, const float *eps, int ld_eps, int64_t frames, int64_t HW, float guidance, int cfg,
                        float alpha_t, float alpha_prev, float *latents, void *xin, int nrep, void *stream)
{
    GC_REQUIRE(ld_eps >= 4 && ld_eps % 4 == 0 && alpha_t > 0.f, "cfg_ddim: bad arguments");
    const float c_x = sqrtf(alpha_prev / alpha_t);
    const float c_e = sqrtf(1.f - alpha_prev) - sqrtf(alpha_prev * (1.f - alpha_t) / alpha_t);
    const int64_t n = frames * HW;
    DN_DISPATCH(dtype,
                hipLaunchKernelGGL((k_cfg_ddim<BF16>), dim3(ew_grid(n)), dim3(256), 0, gc::S(stream), eps, ld_eps, frames, HW, guidance, cfg, c_x, c_e, latents, (unsigned short *)xin, nrep),
                hipLaunchKernelGGL((k_cfg_ddim<F16>), dim3(ew_grid(n)), dim3(256), 0, gc::S(stream), eps, ld_eps, frames, HW, guidance, cfg, c_x, c_e, latents, (unsigned short *)xin, nrep));
    return gc::check_launch("gc_dn_cfg_ddim_step");
}

}  // extern "C"
