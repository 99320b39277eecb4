// dn_gemm.hip -- bf16/f16 MFMA GEMM and implicit-GEMM 3x3 convolution for the SD1.5 UNet / ControlNet /
// VAE blocks (gfx950).  Replaces the cuBLAS / cuDNN calls diffusers issues for every Linear and Conv2d
// of UNet2DConditionModel / ControlNetModel / AutoencoderKL reached from
// /root/reference/gaussctrl/gc_pipeline.py:142-145,209-219.
//
// out[m][n] = epilogue( sum_k Act[m][k] * W[n][k] ),  Act = row-major matrix (Linear / 1x1 conv on NHWC)
// or the on-the-fly im2col of an NHWC tensor (3x3, pad 0|1, stride 1|2, optional fused nearest x2 upsample).
//
// CDNA4 mapping: 128(m) x {128|160}(n) x 64(k) workgroup tile, 256 lanes = 4 wave64 in 2x2, each wave a
// 64 x {64|80} sub-tile = 4 x {4|5} v_mfma_f32_16x16x32 accumulators (fp32).  160-wide n tiles exist because every
// SD1.5 channel count (320/640/960/1280/1920/2560) is a multiple of 160 but not of 128.
// Operands are staged global -> VGPR -> LDS in 16-byte chunks, XOR-swizzled so every ds_read_b128 fragment read is
// bank-conflict free.  The k loop is software pipelined two tiles deep: while tile t is multiplied out of LDS
// buffer t&1, tile t+1 sits in one register set (already landed or landing) and tile t+2's global loads are being
// issued into the other -- the loop is otherwise bound by the ~1 us global->LDS->MFMA dependency chain per k-tile
// (measured: 22 us for 20 k-tiles on an idle chip).  One barrier per k-tile.
// The MFMA is issued "swapped" (A operand = weights, B operand = activations) so each lane ends up with 4
// CONSECUTIVE output channels of one output row: bias / row-vector / residual / SiLU / GEGLU are lane-local and the
// store is one 8-byte write per accumulator.
// Small-M / long-K problems (3x3 convs on 16x16 / 8x8 feature maps: 12 / 3 m-tiles, 90-360 k-tiles) are split along K
// across workgroups (fp32 partial slabs + a small reduce-epilogue kernel) so that all 256 CUs stream the (large) weight
// matrix together.
#pragma once
#include "dn_common.h"
#include <type_traits>

namespace dng {
struct GemmArgs {
    int64_t M, N, K;
    const void *A; int64_t lda;
    int B, Hi, Wi, Cin, Ho, Wo, stride, ups, pad;
    const void *W;
    const float *bias;
    const float *rowvec; int64_t ld_rowvec; int64_t rows_per_batch;
    const void *residual; int64_t ldr;
    float out_scale;
    int act, geglu;
    void *out; int64_t ldc; int out_f32;
    void *out_t; int64_t ldt; int64_t t_batch_stride; int64_t t_col0;
    int splits; int tiles_per_split;    // split-K: k-tiles [z*tps, min(nk, (z+1)*tps))
    int persist;                        // > 0: k_gemm8p with this many workgroups (multi-round short-K linears)
    float *ws;                          // fp32 [splits][M][N] partial slabs when splits > 1
    const void *zeros;                  // >= 16 bytes of zeros (k_gemm8: source of out-of-range / padding lanes)
    // LayerNorm folded into this GEMM (consumer side): Act rows are the UN-normalised x, W carries gamma, and
    // out = rstd[m] * (acc - mean[m] * colsum[n]) + bias'[n]   with (mean, rstd) from row_stats[m] = (sum x, sum x^2) over the K columns
    const float *row_stats; int row_stat_slots; const float *colsum; float ln_eps, ln_inv_k;
    // statistics of THIS GEMM's stored output (producer side):
    float *out_row_stats;               // [slots][M][2] (sum, sum^2) of each row over one column slab per slot, PLAIN stores (no atomics, no
                                        // zero-init): slot = wave column (16 NTW wide) of the fused epilogue / 256-column block of the
                                        // split-K reduce; the consumer adds the slots up  -> LayerNorm folded into the next GEMM
    float *out_group_stats;             // [M / rows_per_batch][groups][2] per (batch, GroupNorm group), reduced in LDS per workgroup, then
    int gn_groups, gn_cpg;              // a few float atomics into the zero-initialised buffer     -> gc_dn_groupnorm_apply
    const unsigned char *w_scale;       // fp8 path: E8M0 scale byte per weight row [N] (weights stored as e4m3 * 2^(127 - byte))
    int a_scale;                        // fp8 path: E8M0 scale byte of the whole activation tensor
    int out_fp8; float out_qscale;      // fp8 path, plain epilogue: store e4m3 BYTES of v * out_qscale (= 2^(127 - out_fp8)) at out[m * ldc + n]: the
                                        // e4m3 activation operand of the next fp8 GEMM (GEGLU hidden -> FF down projection)
    int dbg;                            // experiment switches (kernel_variant bits 8..): 1 no global group atomics, 2 no LDS atomics, 4 no DPP
    // partial GroupNorm-group sums of the stored output, the statistics pass of the GroupNorm that follows (k_gemm8 CS = true,
    // k_splitk_epilogue_cs): chan_parts[b][slab][group][half] = (sum, sum^2) over the rows of batch b inside the slab-th row tile (cp_rows
    // rows each, tiles counted over all M rows) that overlaps batch b and over the group's channels inside this workgroup's column tile:
    // half 0 when the tile holds the group's first channel, half 1 for the rest of a group that straddles two column tiles (gn_cpg <= tile
    // width: at most two).  PLAIN stores: no atomics (round 2's per-(batch, group) float atomics cost 26 ms per chunk in same-line
    // contention), no zero-init; gc_dn_groupnorm_apply_parts adds them up in its prologue.
    float *chan_parts; int cp_nslab; int cp_rows;
    // lean LayerNorm-fold kernels only (LNV != 0): weight SETS selected by the row tile -- rows [s w_set_rows, (s + 1) w_set_rows) multiply weight
    // matrix s (W + s w_set_stride elements; bias / colsum + s N): the text cross-attention folded into two GEMMs has one matrix pair per CFG half
    // (the text differs).  0: one set.  sm_keys (LNV 3): valid score columns per 80-column head block of the softmax-heads epilogue.
    int64_t w_set_rows, w_set_stride;
    int pw;                             // column-panel width of the tile order (tile_coords below); 0 = m-major
    int sm_keys;
    int conv_korder;                    // k_gemm8 fast convs: 1 = tap-inner k-tile sequence (default), 0 = tap-outer (kernel_variant 0x1000)
};

// (blockIdx.x, blockIdx.y) of a (tiles, k-slices) launch grid -> (m block, n block, k-slice) of this workgroup.
// Workgroups go to the 8 XCDs round-robin in linear dispatch order (XCD = (x + gridDim.x * y) % 8) and every XCD has its own L2, so each XCD is
// given a CONTIGUOUS run of the logical order [k-slice][column panel][m block][n block inside the panel]:
//  * with k-slices an XCD owns whole slices (or a run inside one): every weight / activation k-range is fetched over the fabric by one XCD
//    instead of by all eight (the 8 x 8-map convolutions: 30 tiles x 15 slices);
//  * inside a slice the run is a compact patch of the tile grid, `pw` n blocks wide (0 = whole rows, the m-major order): the workgroups that
//    run together on an XCD share rows AND columns.  The host picks pw from the panel bytes (gc_dn_gemm: small-M / wide-N problems take narrow
//    panels -- their weight panels are what the fabric moves; the 64 x 64-map convolutions keep whole rows -- their activations are).
// The assignment does not change what any (tile, slice) computes: results are bit-identical for every pw.
__device__ __forceinline__ void tile_coords(int64_t bid, int nbm, int nbn, int pw, int64_t &mblk, int64_t &nblk)
{
    if (pw <= 0 || pw >= nbn) { mblk = bid / nbn; nblk = bid % nbn; return; }
    const int64_t per = (int64_t)nbm * pw, p = bid / per, within = bid - p * per;
    const int64_t w = (nbn - p * pw) < pw ? (nbn - p * pw) : pw;              // the last panel may be narrower
    mblk = within / w; nblk = p * pw + within % w;
}
__device__ __forceinline__ void wg_tile(const GemmArgs &g, int nbm, int nbn, int64_t &mblk, int64_t &nblk, int &slice)
{
    const int64_t tiles = gridDim.x, total = tiles * gridDim.y, lin = blockIdx.x + tiles * blockIdx.y;
    const int64_t xcd = lin & 7, qq = total >> 3, rr = total & 7;
    const int64_t L = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (lin >> 3);
    slice = (int)(L / tiles);
    tile_coords(L - (int64_t)slice * tiles, nbm, nbn, g.pw, mblk, nblk);
}
}  // namespace dng

namespace {
using namespace dn;
using dng::GemmArgs;

template <int I, int N, class F> __device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

constexpr int BM = 128, BK = 64;
constexpr int NT = 256;


constexpr int GS_SLOTS = 20, GS_MAXG = 32;      // LDS scratch of the group statistics: batches a workgroup tile can touch x groups

// sum over the 16 lanes of a DPP row (lanes 16 j .. 16 j + 15), result in every lane: row_ror 8, 4, 2, 1
__device__ __forceinline__ float row16_sum(float v)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    return v;
}

// byte offset of 16-byte chunk `c` (0..7) of row `r` inside a [rows][64] 2-byte tile
__device__ __forceinline__ int lds_off(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

// epilogue of one lane's 4 consecutive output channels (n .. n+3) of row m; v = raw accumulators (+ gate for GEGLU)
// (split-K reduce kernel) epilogue of 4 consecutive output channels of one row; on return v holds the values as stored
template <class T>
__device__ __forceinline__ void epilogue_store(const GemmArgs &g, int64_t m, int64_t n, int64_t on, float *v, const float *gate, const uint2 *res_pre = nullptr)
{       // res_pre: the residual words of (m, on .. on + 3), loaded by the caller ahead of its own loads (split-K reduce kernels)
    const int64_t bidx = g.rowvec ? m / g.rows_per_batch : 0;
    if (g.row_stats) {
        float2 rs = make_float2(0.f, 0.f);
        for (int sl = 0; sl < g.row_stat_slots; ++sl) {
            const float2 t = *reinterpret_cast<const float2 *>(g.row_stats + ((int64_t)sl * g.M + m) * 2);
            rs.x += t.x; rs.y += t.y;
        }
        const float mean = rs.x * g.ln_inv_k, rstd = rsqrtf(fmaxf(rs.y * g.ln_inv_k - mean * mean, 0.f) + g.ln_eps);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = rstd * (v[r] - mean * g.colsum[n + r]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (g.bias) v[r] += g.bias[n + r];
        if (g.rowvec) v[r] += g.rowvec[bidx * g.ld_rowvec + n + r];
    }
    if (g.geglu) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float gt = gate[r];
            if (g.bias) gt += g.bias[n + 16 + r];
            v[r] = v[r] * gelu_erf(gt);
        }
    }
    if (g.act == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = silu(v[r]);
    } else if (g.act == 2) {   // image post-process of pipe(output_type='pt'): (x/2 + 0.5).clamp(0,1)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fminf(fmaxf(v[r] * 0.5f + 0.5f, 0.f), 1.f);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] *= g.out_scale;
    if (g.residual) {
        const uint2 rr = res_pre ? *res_pre : *reinterpret_cast<const uint2 *>((const unsigned char *)g.residual + (m * g.ldr + on) * 2);
        v[0] += T::to_f((unsigned short)(rr.x & 0xffff)); v[1] += T::to_f((unsigned short)(rr.x >> 16));
        v[2] += T::to_f((unsigned short)(rr.y & 0xffff)); v[3] += T::to_f((unsigned short)(rr.y >> 16));
    }
    const bool to_t = g.out_t && on >= g.t_col0;
    const uint2 pk = make_uint2(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]));
    if (g.out && !(to_t && g.t_col0 > 0)) {
        if (g.out_f32)
            *reinterpret_cast<float4 *>((float *)g.out + m * g.ldc + on) = make_float4(v[0], v[1], v[2], v[3]);
        else
            *reinterpret_cast<uint2 *>((unsigned char *)g.out + (m * g.ldc + on) * 2) = pk;
    }
    if (to_t) {   // transposed copy out_t[b][n - t_col0][tok] (V operand of the attention kernel)
        const int64_t b = m / g.rows_per_batch, tok = m - b * g.rows_per_batch;
        unsigned short *o = (unsigned short *)g.out_t + b * g.t_batch_stride + tok;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[(on - g.t_col0 + r) * g.ldt] = T::from_f(v[r]);
    }
    if (!g.out_f32) {
        v[0] = T::to_f((unsigned short)(pk.x & 0xffff)); v[1] = T::to_f((unsigned short)(pk.x >> 16));
        v[2] = T::to_f((unsigned short)(pk.y & 0xffff)); v[3] = T::to_f((unsigned short)(pk.y >> 16));
    }
}

// LayerNorm-folded consumer, 8-wave kernels: the (sum, sum^2) partials the producer left per column slab are added up for the
// workgroup's BM rows at kernel START into an LDS array behind the operand ring -- the L2 round trips sit before / under the first
// k-tiles instead of in the epilogue's critical path (measured: +7 .. +35 us when the slab loop ran there).  512 threads: row = t % 256,
// slab parity = t / 256, 4 loads in flight per thread; the two halves meet through LDS float adds.
template <int BM>
__device__ __forceinline__ void row_stats_prologue(const GemmArgs &g, int64_t m_base, float *srow)
{
    const int t = threadIdx.x, r = t & 255, half = t >> 8;
    if (t < 2 * BM) srow[t] = 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (r < BM) {
        const int64_t m = m_base + r < g.M ? m_base + r : g.M - 1;
        const float *rp = g.row_stats + m * 2;
        const int64_t st = g.M * 2;
        float sx = 0.f, sy = 0.f;
        int sl = half;
        for (; sl + 6 < g.row_stat_slots; sl += 8) {
            const float2 a = *reinterpret_cast<const float2 *>(rp + sl * st), b = *reinterpret_cast<const float2 *>(rp + (sl + 2) * st);
            const float2 c = *reinterpret_cast<const float2 *>(rp + (sl + 4) * st), d = *reinterpret_cast<const float2 *>(rp + (sl + 6) * st);
            sx += (a.x + b.x) + (c.x + d.x); sy += (a.y + b.y) + (c.y + d.y);
        }
        for (; sl < g.row_stat_slots; sl += 2) {
            const float2 a = *reinterpret_cast<const float2 *>(rp + sl * st);
            sx += a.x; sy += a.y;
        }
        __hip_atomic_fetch_add(srow + 2 * r, sx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(srow + 2 * r + 1, sy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // nothing of this prologue is in flight when the counted LDS-DMA waits start
}

// The epilogue of one wave's (16 MT) x (16 NTW) accumulator tile, shared by k_gemm (MT = 4) and k_gemm8.
// Lane (fr = lane & 15, fc = lane >> 4) holds out[m = m_wave + 16 mt + fr][n = n_wave + 16 nt + 4 fc + r], r = 0..3.
// All operand loads of the tile (bias, LN column sums once; row statistics, row-vector and residual of every m-tile) are issued
// before the first one is consumed: the naive per-accumulator load -> use chain costs ~1 us of latency per m-tile on every launch.
template <class T, int NTW, int MT, int NTHREADS>
__device__ __forceinline__ void wave_epilogue(const GemmArgs &g, f32x4 (&acc)[NTW][MT], int64_t m_base, int64_t m_wave, int64_t n_wave, int lane,
                                              unsigned char *smem, const float *srow = nullptr, int slice = 0)
{
    const int fr = lane & 15, fc = lane >> 4;
    const int64_t n_lane = n_wave + fc * 4;
    if (g.splits > 1) {   // partial sums of this k-slice: plain 16-byte stores into slab `slice`
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int64_t m = m_wave + mt * 16 + fr;
            if (m >= g.M) continue;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int64_t n = n_lane + nt * 16;
                if (n < g.N)
                    *reinterpret_cast<float4 *>(g.ws + ((int64_t)slice * g.M + m) * g.N + n) =
                        make_float4(acc[nt][mt][0], acc[nt][mt][1], acc[nt][mt][2], acc[nt][mt][3]);
            }
        }
        return;
    }
    if (g.out_group_stats) {      // LDS scratch of the group statistics: raw barriers (LDS only -- no vmcnt drain of loads / stores in flight)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // every wave has finished reading the last k-tile
        for (int idx = threadIdx.x; idx < GS_SLOTS * GS_MAXG * 2; idx += NTHREADS) reinterpret_cast<float *>(smem)[idx] = 0.f;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    float4 bia[NTW], csm[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int64_t n = n_lane + nt * 16;
        bia[nt] = (g.bias && n < g.N) ? *reinterpret_cast<const float4 *>(g.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        csm[nt] = (g.row_stats && n < g.N) ? *reinterpret_cast<const float4 *>(g.colsum + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    uint2 rs_all[MT][NTW];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int64_t m = m_wave + mt * 16 + fr;
        const bool okm = m < g.M;
        const int64_t mc = okm ? m : 0;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int64_t n = n_lane + nt * 16;
            const bool okn = okm && n < g.N && !(g.geglu && (nt & 1));
            const int64_t on = g.geglu ? (n_wave + nt * 16) / 2 + fc * 4 : n;
            rs_all[mt][nt] = (g.residual && okn) ? *reinterpret_cast<const uint2 *>((const unsigned char *)g.residual + (mc * g.ldr + on) * 2) : make_uint2(0u, 0u);
        }
    }
    // GroupNorm statistics of the output: per-lane channel sums over the wave's rows -> DPP sum over the 16 rows of a tile -> LDS
    // atomics into [batch slot][group] of the workgroup -> ONE global float atomic per (batch, group) the tile touches.
    const bool want_cs = g.out_group_stats != nullptr;
    float *red = reinterpret_cast<float *>(smem);           // [GS_SLOTS][GS_MAXG][2]; the operand tiles in LDS are dead by now
    const int64_t b0 = want_cs ? m_base / g.rows_per_batch : 0;
    float cs[NTW][4], css[NTW][4];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { cs[nt][r] = 0.f; css[nt][r] = 0.f; }
    int64_t cs_b = -1;       // batch of the rows accumulated in cs / css (an m-tile of 16 rows never straddles batches: rows_per_batch % 16 == 0)
    auto flush_cs = [&]() __attribute__((always_inline)) {
        float *rb = red + (cs_b - b0) * (GS_MAXG * 2);
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int64_t n = n_lane + nt * 16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a = (g.dbg & 4) ? cs[nt][r] : row16_sum(cs[nt][r]), b = (g.dbg & 4) ? css[nt][r] : row16_sum(css[nt][r]);
                if (fr == 0 && n + r < g.N && !(g.dbg & 2)) {
                    const int gi = (int)(n + r) / g.gn_cpg;
                    __hip_atomic_fetch_add(rb + 2 * gi, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(rb + 2 * gi + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                cs[nt][r] = 0.f; css[nt][r] = 0.f;
            }
        }
    };
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int64_t m = m_wave + mt * 16 + fr;
        const int64_t m_tile = m_wave + mt * 16;                 // wave-uniform
        if (m_tile >= g.M) break;
        const bool okm = m < g.M;
        if (want_cs) {
            const int64_t b = m_tile / g.rows_per_batch;
            if (cs_b >= 0 && b != cs_b) flush_cs();
            cs_b = b;
        }
        const uint2 *rs = rs_all[mt];
        float mean = 0.f, rstd = 1.f;
        if (g.row_stats) {       // the producer left one partial (sum, sum^2) per column slab: add them up (scalars: no register array)
            float sx = 0.f, sy = 0.f;
            if (srow) {          // summed over the slabs by row_stats_prologue while the first k-tiles were in flight
                const float2 t = *reinterpret_cast<const float2 *>(srow + 2 * (okm ? (int)(m - m_base) : 0));
                sx = t.x; sy = t.y;
            } else {
                const float *rp = g.row_stats + (okm ? m : 0) * 2;
                for (int sl = 0; sl < g.row_stat_slots; ++sl) {
                    const float2 t = *reinterpret_cast<const float2 *>(rp + (int64_t)sl * g.M * 2);
                    sx += t.x; sy += t.y;
                }
            }
            mean = sx * g.ln_inv_k;
            rstd = rsqrtf(fmaxf(sy * g.ln_inv_k - mean * mean, 0.f) + g.ln_eps);
        }
        float4 rv[NTW];
        const int64_t bidx = (g.rowvec && okm) ? m / g.rows_per_batch : 0;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int64_t n = n_lane + nt * 16;
            rv[nt] = (g.rowvec && n < g.N) ? *reinterpret_cast<const float4 *>(g.rowvec + bidx * g.ld_rowvec + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float rsum = 0.f, rsq = 0.f;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int64_t n = n_lane + nt * 16;
            if (n >= g.N) continue;
            if (g.geglu && (nt & 1)) continue;
            float a0 = acc[nt][mt][0], a1 = acc[nt][mt][1], a2 = acc[nt][mt][2], a3 = acc[nt][mt][3];
            if (g.row_stats) {
                a0 = rstd * (a0 - mean * csm[nt].x); a1 = rstd * (a1 - mean * csm[nt].y);
                a2 = rstd * (a2 - mean * csm[nt].z); a3 = rstd * (a3 - mean * csm[nt].w);
            }
            float v[4] = {a0 + bia[nt].x + rv[nt].x, a1 + bia[nt].y + rv[nt].y, a2 + bia[nt].z + rv[nt].z, a3 + bia[nt].w + rv[nt].w};
            int64_t on = n;
            if (g.geglu) {   // weights are row-permuted in 16-blocks [x | gate]; partner tile = nt + 1 (NTW is even here)
                constexpr int NP = NTW - 1;
                const int np = nt + 1 < NTW ? nt + 1 : NP;
                float g0 = acc[np][mt][0], g1 = acc[np][mt][1], g2 = acc[np][mt][2], g3 = acc[np][mt][3];
                if (g.row_stats) {
                    g0 = rstd * (g0 - mean * csm[np].x); g1 = rstd * (g1 - mean * csm[np].y);
                    g2 = rstd * (g2 - mean * csm[np].z); g3 = rstd * (g3 - mean * csm[np].w);
                }
                v[0] *= gelu_erf(g0 + bia[np].x); v[1] *= gelu_erf(g1 + bia[np].y);
                v[2] *= gelu_erf(g2 + bia[np].z); v[3] *= gelu_erf(g3 + bia[np].w);
                on = (n_wave + nt * 16) / 2 + fc * 4;
            }
            if (g.act == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = silu(v[r]);
            } else if (g.act == 2) {   // image post-process of pipe(output_type='pt'): (x/2 + 0.5).clamp(0,1)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fminf(fmaxf(v[r] * 0.5f + 0.5f, 0.f), 1.f);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= g.out_scale;
            if (g.residual) {
                v[0] += T::to_f((unsigned short)(rs[nt].x & 0xffff)); v[1] += T::to_f((unsigned short)(rs[nt].x >> 16));
                v[2] += T::to_f((unsigned short)(rs[nt].y & 0xffff)); v[3] += T::to_f((unsigned short)(rs[nt].y >> 16));
            }
            if (!okm) continue;
            const bool to_t = g.out_t && on >= g.t_col0;     // fused QKV: columns >= t_col0 (V) go ONLY to the transposed buffer
            const uint2 pk = make_uint2(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]));
            if (g.out && !(to_t && g.t_col0 > 0)) {
                if (g.out_f32)
                    *reinterpret_cast<float4 *>((float *)g.out + m * g.ldc + on) = make_float4(v[0], v[1], v[2], v[3]);
                else
                    *reinterpret_cast<uint2 *>((unsigned char *)g.out + (m * g.ldc + on) * 2) = pk;
            }
            if (to_t) {   // transposed copy out_t[b][n - t_col0][tok] (V operand of the attention kernel)
                const int64_t b = m / g.rows_per_batch, tok = m - b * g.rows_per_batch;
                unsigned short *o = (unsigned short *)g.out_t + b * g.t_batch_stride + tok;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[(on - g.t_col0 + r) * g.ldt] = T::from_f(v[r]);
            }
            if (g.out_row_stats || want_cs) {     // statistics of the values as STORED (rounded to the activation type)
                float t[4];
                if (g.out_f32) { t[0] = v[0]; t[1] = v[1]; t[2] = v[2]; t[3] = v[3]; }
                else {
                    t[0] = T::to_f((unsigned short)(pk.x & 0xffff)); t[1] = T::to_f((unsigned short)(pk.x >> 16));
                    t[2] = T::to_f((unsigned short)(pk.y & 0xffff)); t[3] = T::to_f((unsigned short)(pk.y >> 16));
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    rsum += t[r]; rsq += t[r] * t[r];
                    cs[nt][r] += t[r]; css[nt][r] += t[r] * t[r];
                }
            }
        }
        if (g.out_row_stats && n_wave < g.N) {   // the 4 lanes fc = 0..3 of a row hold disjoint column quads: combine, one plain store per (row, wave column)
            rsum += __shfl_xor(rsum, 16, 64); rsq += __shfl_xor(rsq, 16, 64);
            rsum += __shfl_xor(rsum, 32, 64); rsq += __shfl_xor(rsq, 32, 64);
            const int64_t slot = n_wave / (16 * NTW);
            if (fc == 0 && okm) *reinterpret_cast<float2 *>(g.out_row_stats + (slot * g.M + m) * 2) = make_float2(rsum, rsq);
        }
    }
    if (want_cs) {
        if (cs_b >= 0) flush_cs();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the LDS atomics; NOT the output stores still in flight
        __builtin_amdgcn_s_barrier();
        for (int idx = threadIdx.x; idx < GS_SLOTS * GS_MAXG * 2; idx += NTHREADS) {
            const float v = red[idx];
            if (v != 0.f && !(g.dbg & 1)) {
                const int sl = idx / (GS_MAXG * 2), rem = idx - sl * (GS_MAXG * 2);
                unsafeAtomicAdd(g.out_group_stats + ((b0 + sl) * g.gn_groups) * 2 + rem, v);
            }
        }
    }
}

// MODE: 0 linear, 1 conv generic (any Cin % 8 == 0), 2 conv fast (Cin % 64 == 0: one tap per k-tile)
// NTW : n-tiles (of 16) per wave: 4 -> BN = 128, 5 -> BN = 160
// FUSE = false: the plain epilogue (bias / row-vector / residual / activation / GEGLU / transposed V), kept lean -- the statistics and
// LayerNorm-fold code roughly doubles the straight-line epilogue, and every launch pays for the instruction fetch of what it skips;
// FUSE = true: wave_epilogue (producer-side row / group statistics, LayerNorm-folded consumer).
template <class T, int MODE, int NTW, bool FUSE>
__global__ __launch_bounds__(NT, 2) void k_gemm(const GemmArgs g)
{
    constexpr int BN = 32 * NTW;
    constexpr int WCH = BN * 8 / NT;            // W chunks per lane per k-tile (4 | 5)
    constexpr int STAGE = BM * 128 + BN * 128;  // bytes per LDS stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int nbn = (int)((g.N + BN - 1) / BN);
    // workgroup b runs on XCD b % 8: give each XCD a contiguous run of logical tiles (bijective remap) so the
    // workgroups that share an activation panel share an L2
    int64_t mblk, nblk;
    int slice;
    wg_tile(g, (int)((g.M + BM - 1) / BM), nbn, mblk, nblk, slice);
    const int64_t m_base = mblk * BM, n_base = nblk * BN;
    const int nk = (int)((g.K + BK - 1) / BK);
    const int kt0 = slice * g.tiles_per_split, kt1 = min(nk, kt0 + g.tiles_per_split);
    if (kt0 >= kt1) return;

    // ---- per-lane staging coordinates: 4 Act chunks + WCH W chunks per k-tile.  All per-lane address arithmetic is
    // 32-bit element offsets from the (uniform) tensor bases; for the 3x3 fast path the offset of a chunk is
    // pixel_offset(lane) + tap_offset(k-tile, uniform): one add and two compares per chunk.
    int a_row[4], a_chunk[4];
    int a_off[4];                 // MODE 0: m*lda + chunk*8 ; MODE 1/2: b*Hi*Wi*Cin (+ chunk*8 [+ (y0*Wi+x0)*Cin when !ups])
    int a_y[4], a_x[4];           // MODE 1/2: oy*stride - pad, ox*stride - pad
    bool a_ok[4];
    int w_off[WCH];
    bool w_ok[WCH];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = tid + NT * i;
        a_row[i] = q >> 3; a_chunk[i] = q & 7;
        const int64_t m = m_base + a_row[i];
        a_ok[i] = m < g.M;
        const int mm = a_ok[i] ? (int)m : 0;
        if (MODE != 0) {
            const int hw = g.Ho * g.Wo;
            const int b = mm / hw;
            const int rem = mm - b * hw;
            const int oy = rem / g.Wo;
            a_y[i] = oy * g.stride - g.pad; a_x[i] = (rem - oy * g.Wo) * g.stride - g.pad;
            a_off[i] = b * g.Hi * g.Wi * g.Cin + (MODE == 2 ? a_chunk[i] * 8 : 0);
            if (MODE == 2 && !g.ups) a_off[i] += (a_y[i] * g.Wi + a_x[i]) * g.Cin;
        } else {
            a_off[i] = mm * (int)g.lda + a_chunk[i] * 8;
            a_y[i] = a_x[i] = 0;
        }
    }
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
        const int q = tid + NT * i;
        const int64_t n = n_base + (q >> 3);
        w_ok[i] = n < g.N;
        w_off[i] = (w_ok[i] ? (int)n : 0) * (int)g.K + (q & 7) * 8;
    }
    const int Hin = g.ups ? g.Hi * 2 : g.Hi, Win = g.ups ? g.Wi * 2 : g.Wi;
    const unsigned char *Ab = (const unsigned char *)g.A, *Wb = (const unsigned char *)g.W;
    const int ktl = kt1 - 1;
    // loader state (wave-uniform): tap / channel offset of the NEXT k-tile to load (tiles are loaded in order)
    int ld_tap = 0, ld_ci = 0;
    if (MODE == 2) { ld_tap = (kt0 * BK) / g.Cin; ld_ci = kt0 * BK - ld_tap * g.Cin; }

    // Loads are UNCONDITIONAL (invalid lanes read offset 0 and are zeroed when the tile is written to LDS): no
    // exec-mask branches around the global loads, so they stay in flight with counted s_waitcnt vmcnt(N).
    auto load_tile = [&](int kt, uint4 *ra, uint4 *rw, unsigned &mask) __attribute__((always_inline)) {
        const int kb = kt * BK;
        int dy_u = 0, dx_u = 0, tap_off = 0;
        if (MODE == 2) {
            dy_u = ld_tap / 3; dx_u = ld_tap - dy_u * 3;
            tap_off = g.ups ? ld_ci : (dy_u * g.Wi + dx_u) * g.Cin + ld_ci;
            if (kt < ktl) { ld_ci += BK; if (ld_ci >= g.Cin) { ld_ci -= g.Cin; ++ld_tap; } }
        }
        unsigned mk = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bool ok;
            int off;
            if (MODE == 2) {
                int yi = a_y[i] + dy_u, xi = a_x[i] + dx_u;
                ok = a_ok[i] && (unsigned)yi < (unsigned)Hin && (unsigned)xi < (unsigned)Win;
                off = a_off[i] + tap_off;
                if (g.ups) off += ((yi >> 1) * g.Wi + (xi >> 1)) * g.Cin;
            } else if (MODE == 1) {
                const int k0 = kb + a_chunk[i] * 8;
                const int kc = k0 < (int)g.K ? k0 : 0;
                const int tap = kc / g.Cin;
                const int ci = kc - tap * g.Cin;
                const int dy = tap / 3, dx = tap - dy * 3;
                int yi = a_y[i] + dy, xi = a_x[i] + dx;
                ok = a_ok[i] && k0 < (int)g.K && (unsigned)yi < (unsigned)Hin && (unsigned)xi < (unsigned)Win;
                if (g.ups) { yi >>= 1; xi >>= 1; }
                off = a_off[i] + (yi * g.Wi + xi) * g.Cin + ci;
            } else {
                ok = a_ok[i] && (kb + a_chunk[i] * 8) < (int)g.K;
                off = a_off[i] + kb;
            }
            ra[i] = *reinterpret_cast<const uint4 *>(Ab + (size_t)(unsigned)((ok ? off : 0) * 2));
            mk |= (ok ? 1u : 0u) << i;
        }
#pragma unroll
        for (int i = 0; i < WCH; ++i) {
            const int q = tid + NT * i;
            const bool ok = w_ok[i] && (kb + (q & 7) * 8) < (int)g.K;
            rw[i] = *reinterpret_cast<const uint4 *>(Wb + (size_t)(unsigned)((ok ? w_off[i] + kb : 0) * 2));
            mk |= (ok ? 1u : 0u) << (8 + i);
        }
        mask = mk;
    };
    constexpr unsigned FULL = 0xFu | (((1u << WCH) - 1u) << 8);
    int sa_off[4], sw_off[WCH];
#pragma unroll
    for (int i = 0; i < 4; ++i) sa_off[i] = lds_off(a_row[i], a_chunk[i]);
#pragma unroll
    for (int i = 0; i < WCH; ++i) { const int q = tid + NT * i; sw_off[i] = BM * 128 + lds_off(q >> 3, q & 7); }
    auto store_tile = [&](int buf, uint4 *ra, uint4 *rw, unsigned mask) __attribute__((always_inline)) {
        unsigned char *sa = smem + buf * STAGE;
        if (!__all(mask == FULL)) {   // border / tail tiles only: zero the lanes that read a clamped address
#pragma unroll
            for (int i = 0; i < 4; ++i) {   // value-level masking (a ?: on the arrays would become a pointer select -> scratch)
                const unsigned km = 0u - ((mask >> i) & 1u);
                ra[i] = make_uint4(ra[i].x & km, ra[i].y & km, ra[i].z & km, ra[i].w & km);
            }
#pragma unroll
            for (int i = 0; i < WCH; ++i) {
                const unsigned km = 0u - ((mask >> (8 + i)) & 1u);
                rw[i] = make_uint4(rw[i].x & km, rw[i].y & km, rw[i].z & km, rw[i].w & km);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4 *>(sa + sa_off[i]) = ra[i];
#pragma unroll
        for (int i = 0; i < WCH; ++i) *reinterpret_cast<uint4 *>(sa + sw_off[i]) = rw[i];
    };

    f32x4 acc[NTW][4];   // [nt][mt]
#pragma unroll
    for (int a = 0; a < NTW; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fc = lane >> 4;
    // fragment addresses: row = rowbase + 16 t + fr (rowbase % 16 == 0) => the swizzle term ((row >> 1) & 7) does not depend
    // on t: address(t, ks) = base_ks + t * 2048 -- two VGPRs per operand, everything else is an immediate offset
    const int swz = (fr >> 1) & 7;
    const int fx0 = ((fc ^ swz) << 4), fx1 = (((fc + 4) ^ swz) << 4);
    const int aw0 = BM * 128 + (wn * (16 * NTW) + fr) * 128, aa0 = (wm * 64 + fr) * 128;
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const unsigned char *sb = smem + buf * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int fx = ks ? fx1 : fx0;
            uint4 fw[NTW], fa[4];
#pragma unroll
            for (int t = 0; t < NTW; ++t) fw[t] = *reinterpret_cast<const uint4 *>(sb + aw0 + fx + t * 2048);
#pragma unroll
            for (int t = 0; t < 4; ++t) fa[t] = *reinterpret_cast<const uint4 *>(sb + aa0 + fx + t * 2048);
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = T::mfma(fw[nt], fa[mt], acc[nt][mt]);
        }
    };

    // Loads / LDS writes past the last tile are issued anyway (clamped to the last tile, written to the buffer nobody
    // reads any more): straight-line loop body, no conditionally written register arrays.
    uint4 raA[4], rwA[WCH], raB[4], rwB[WCH];
    unsigned mkA = 0, mkB = 0;
    load_tile(kt0, raA, rwA, mkA);
    load_tile(min(kt0 + 1, ktl), raB, rwB, mkB);
    store_tile(0, raA, rwA, mkA);
    __syncthreads();
    for (int kt = kt0; kt < kt1; kt += 2) {
        load_tile(min(kt + 2, ktl), raA, rwA, mkA);
        compute(0);
        store_tile(1, raB, rwB, mkB);
        __syncthreads();
        if (kt + 1 >= kt1) break;
        load_tile(min(kt + 3, ktl), raB, rwB, mkB);
        compute(1);
        store_tile(0, raA, rwA, mkA);
        __syncthreads();
    }

    if constexpr (FUSE) {
        wave_epilogue<T, NTW, 4, NT>(g, acc, m_base, m_base + wm * 64, n_base + wn * (16 * NTW), lane, smem, nullptr, slice);
        return;
    }
    // ---- epilogue: lane holds out[m = m0 + (lane&15)][n = n0 + 4*(lane>>4) + r], r = 0..3.
    // All operand loads of a row (bias once per lane; row-vector + residual per m-tile) are issued together BEFORE they are
    // used: the naive per-accumulator load->use chain cost ~1 us of latency x 20 accumulators on every launch.
    const int64_t n_lane = n_base + wn * (16 * NTW) + fc * 4;
    if (g.splits > 1) {   // partial sums of this k-slice: plain 16-byte stores into slab `slice`
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int64_t m = m_base + wm * 64 + mt * 16 + fr;
            if (m >= g.M) continue;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int64_t n = n_lane + nt * 16;
                if (n < g.N)
                    *reinterpret_cast<float4 *>(g.ws + ((int64_t)slice * g.M + m) * g.N + n) =
                        make_float4(acc[nt][mt][0], acc[nt][mt][1], acc[nt][mt][2], acc[nt][mt][3]);
            }
        }
        return;
    }
    float4 bia[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int64_t n = n_lane + nt * 16;
        bia[nt] = (g.bias && n < g.N) ? *reinterpret_cast<const float4 *>(g.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // residual / row-vector operands of ALL four m-tiles are requested before the first one is consumed (one latency, not four)
    uint2 rs_all[4][NTW];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int64_t m = m_base + wm * 64 + mt * 16 + fr;
        const bool okm = m < g.M;
        const int64_t mc = okm ? m : 0;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int64_t n = n_lane + nt * 16;
            const bool okn = okm && n < g.N && !(g.geglu && (nt & 1));
            const int64_t on = g.geglu ? (n_base + wn * (16 * NTW) + nt * 16) / 2 + fc * 4 : n;
            rs_all[mt][nt] = (g.residual && okn) ? *reinterpret_cast<const uint2 *>((const unsigned char *)g.residual + (mc * g.ldr + on) * 2) : make_uint2(0u, 0u);
        }
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int64_t m = m_base + wm * 64 + mt * 16 + fr;
        if (m >= g.M) continue;
        const uint2 *rs = rs_all[mt];
        float4 rv[NTW];
        const int64_t bidx = g.rowvec ? m / g.rows_per_batch : 0;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int64_t n = n_lane + nt * 16;
            rv[nt] = (g.rowvec && n < g.N) ? *reinterpret_cast<const float4 *>(g.rowvec + bidx * g.ld_rowvec + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int64_t n = n_lane + nt * 16;
            if (n >= g.N) continue;
            if (g.geglu && (nt & 1)) continue;
            float v[4] = {acc[nt][mt][0] + bia[nt].x + rv[nt].x, acc[nt][mt][1] + bia[nt].y + rv[nt].y,
                          acc[nt][mt][2] + bia[nt].z + rv[nt].z, acc[nt][mt][3] + bia[nt].w + rv[nt].w};
            int64_t on = n;
            if (g.geglu) {   // weights are row-permuted in 16-blocks [x | gate]; partner tile = nt + 1 (NTW is even here)
                constexpr int NP = NTW - 1;
                const int np = nt + 1 < NTW ? nt + 1 : NP;
                v[0] *= gelu_erf(acc[np][mt][0] + bia[np].x); v[1] *= gelu_erf(acc[np][mt][1] + bia[np].y);
                v[2] *= gelu_erf(acc[np][mt][2] + bia[np].z); v[3] *= gelu_erf(acc[np][mt][3] + bia[np].w);
                on = (n_base + wn * (16 * NTW) + nt * 16) / 2 + fc * 4;
            }
            if (g.act == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = silu(v[r]);
            } else if (g.act == 2) {   // image post-process of pipe(output_type='pt'): (x/2 + 0.5).clamp(0,1)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fminf(fmaxf(v[r] * 0.5f + 0.5f, 0.f), 1.f);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= g.out_scale;
            if (g.residual) {
                v[0] += T::to_f((unsigned short)(rs[nt].x & 0xffff)); v[1] += T::to_f((unsigned short)(rs[nt].x >> 16));
                v[2] += T::to_f((unsigned short)(rs[nt].y & 0xffff)); v[3] += T::to_f((unsigned short)(rs[nt].y >> 16));
            }
            const bool to_t = g.out_t && on >= g.t_col0;     // fused QKV: columns >= t_col0 (V) go ONLY to the transposed buffer
            if (g.out && !(to_t && g.t_col0 > 0)) {
                if (g.out_f32)
                    *reinterpret_cast<float4 *>((float *)g.out + m * g.ldc + on) = make_float4(v[0], v[1], v[2], v[3]);
                else
                    *reinterpret_cast<uint2 *>((unsigned char *)g.out + (m * g.ldc + on) * 2) = make_uint2(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]));
            }
            if (to_t) {   // transposed copy out_t[b][n - t_col0][tok] (V operand of the attention kernel)
                const int64_t b = m / g.rows_per_batch, tok = m - b * g.rows_per_batch;
                unsigned short *o = (unsigned short *)g.out_t + b * g.t_batch_stride + tok;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[(on - g.t_col0 + r) * g.ldt] = T::from_f(v[r]);
            }
        }
    }
}

// =====================================================================================================================
// k_gemm8 -- same 128 x {128,160} x 64 tile, but 8 wave64 (4 x 2, each 32 x {64,80}), operands DMA'd straight into LDS with
// global_load_lds_dwordx4 (no VGPR staging, no ds_write pass) through a 3-stage ring: tile t is multiplied while t+1 and
// t+2 are in flight, ONE barrier per k-tile, counted s_waitcnt vmcnt(N) (never 0 in the main loop).  The LDS image is the
// same XOR-swizzled layout as k_gemm: an LDS-DMA instruction writes wave-uniform-base + lane*16, so the swizzle is applied
// to the per-lane SOURCE chunk (lane l of row-group g writes slot l&7 of row 8g + (l>>3) and therefore fetches chunk
// (l&7) ^ ((row>>1)&7)).  Lanes that fall outside the tensor / in the conv padding fetch from a 16-byte zero page.
// The DMA is issued from inline asm (M0 = LDS base) so hipcc's waitcnt pass does not drain it with vmcnt(0) before every
// ds_read; ordering is by the explicit vmcnt + s_barrier below.  One workgroup per CU (96-108 KiB LDS), 2 waves per SIMD.
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// uniform base + 32-bit lane offset form: no per-lane address arithmetic (M0 is not used by anything else in these kernels)
__device__ __forceinline__ void glds16_s(const void *sbase, unsigned voff, unsigned lds_dst)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// MODE 3 (k_gemm8 only): linear with K % 64 == 0 -- rows past M / N re-read the last valid row (their outputs are never stored),
// so a k-tile's DMA is a uniform base + constant per-lane offset: no VALU at all in the issue path.
// MT: m-tiles (of 16) per wave: workgroup tile (64 MT) x (32 NTW), waves 4 (M) x 2 (N), wave tile (16 MT) x (16 NTW).
// LEAN: the epilogue of the common case only -- bias / row-vector / scale / residual / 2-byte store (no GEGLU, no activation, no fp32 or
// transposed output; checked by the launcher).  The full epilogue is ~470 instructions per accumulator block x MT NTW blocks (7 000 of the
// kernel's 9 400 instructions, 56 KB: more than the instruction cache) although a launch executes a small part of it; every wave fetches
// its way through the rest.  CS implies LEAN.
// LNV (round 5): the LayerNorm fold WITHOUT the everything-epilogue of FUSE -- 1 = producer: the lean epilogue also leaves, per row and wave
// column, the (sum, sum^2) of the values as stored ([slots][M][2], plain stores: GemmArgs::out_row_stats); 2 = consumer: Act rows are the
// un-normalised x, W carries gamma: v = rstd[m] (acc - mean[m] colsum[n]) + bias'[n], the row's (mean, rstd) summed over the producer's slots at
// kernel start (row_stats_prologue) -- on the lean epilogue (attn2.to_q) and on the plain one (Q|K|V^T, GEGLU).
template <class T, int MODE, int NTW, int MT, bool FUSE, bool CS = false, bool LEAN_ = false, int LNV = 0>
__global__ __launch_bounds__(512, MT <= 2 ? 2 : 1) void k_gemm8(const GemmArgs g)
{
    static_assert(!(FUSE && CS), "one statistics epilogue at a time");
    static_assert(LNV == 0 || (!FUSE && !CS && MODE == 3), "lean LayerNorm fold: K % 64 == 0 linears on the non-FUSE epilogues");
    static_assert(LNV != 1 || LEAN_, "row partials come from the lean epilogue");
    static_assert(LNV != 3 || (NTW == 5 && !LEAN_), "softmax-heads epilogue: one 80-column head block per wave column");
    // LNV 3 (text cross-attention folded into GEMMs): LayerNorm-folded consumer whose N columns are attention SCORES against the
    // <= 80 text keys of each head (W = K_text Wq: one 80-row block per head) -- the epilogue takes the softmax over each head block (= this
    // wave's 80 columns of the row: 20 values per lane, 4 lanes per row) and stores the probabilities.
    constexpr bool LNIN = LNV >= 2;
    constexpr bool LEAN = LEAN_ || CS;
    constexpr int BM = 64 * MT;
    constexpr bool CONVF = MODE == 2 || MODE == 4, UPS = MODE == 4;   // MODE 4 = MODE 2 + fused nearest-x2 upsample
    constexpr int BN = 32 * NTW;
    constexpr int STAGE = BM * 128 + BN * 128;
    constexpr int NS = 3;
    constexpr int AG = BM / 8, WG = BN / 8;           // 8-row groups (one LDS-DMA instruction each)
    constexpr int AI = AG / 8, WI = (WG + 7) / 8;     // instructions per wave per tile: A MT, W 2|3
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int nbn = (int)((g.N + BN - 1) / BN);
    int64_t mblk, nblk;
    int slice;
    wg_tile(g, (int)((g.M + BM - 1) / BM), nbn, mblk, nblk, slice);
    const int64_t m_base = mblk * BM, n_base = nblk * BN;
    const int nk_all = (int)((g.K + BK - 1) / BK);
    const int kt0 = slice * g.tiles_per_split;                           // split-K: this workgroup's k-tiles [kt0, kt0 + nk)
    const int nk = min(nk_all, kt0 + g.tiles_per_split) - kt0;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    float *srow = reinterpret_cast<float *>(smem + NS * STAGE);          // FUSE: [BM][2] row sums of the LayerNorm-folded consumer
    if constexpr (FUSE) {
        if (g.row_stats && g.splits == 1) row_stats_prologue<BM>(g, m_base, srow);
    }
    float *ctab = reinterpret_cast<float *>(smem + NS * STAGE);          // CS: [2 batch slots][BN][2] channel sums of this tile, behind the ring
    if constexpr (CS) {      // zeroed here: the k loop's barriers order it before the epilogue's LDS adds
        for (int i = tid; i < 4 * BN; i += 512) ctab[i] = 0.f;
    }

    // ---- per-lane DMA coordinates
    const int lr = lane >> 3, ls = lane & 7;
    int a_off[AI], a_y[AI], a_x[AI], a_ck[AI];
    bool a_ok[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int row = (wid + 8 * i) * 8 + lr;
        a_ck[i] = ls ^ ((row >> 1) & 7);
        const int64_t m = m_base + row;
        a_ok[i] = m < g.M;
        const int mm = a_ok[i] ? (int)m : 0;
        if (MODE == 1 || CONVF) {
            const int hw = g.Ho * g.Wo;
            const int b = mm / hw;
            const int rem = mm - b * hw;
            const int oy = rem / g.Wo;
            a_y[i] = oy * g.stride - g.pad; a_x[i] = (rem - oy * g.Wo) * g.stride - g.pad;
            a_off[i] = b * g.Hi * g.Wi * g.Cin + (CONVF ? a_ck[i] * 8 : 0);
            if (MODE == 2) a_off[i] += (a_y[i] * g.Wi + a_x[i]) * g.Cin;
        } else {
            a_off[i] = mm * (int)g.lda + a_ck[i] * 8;
            if (MODE == 3) a_off[i] = (int)(m < g.M ? m : g.M - 1) * (int)g.lda + a_ck[i] * 8;
            a_y[i] = a_x[i] = 0;
        }
    }
    int w_off[WI], w_ck[WI];
    bool w_ok[WI];
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const int grp = wid + 8 * i;
        const int row = grp * 8 + lr;
        w_ck[i] = ls ^ ((row >> 1) & 7);
        const int64_t n = n_base + row;
        w_ok[i] = grp < WG && n < g.N;
        w_off[i] = (w_ok[i] ? (int)n : 0) * (int)g.K + w_ck[i] * 8;
        if (MODE == 3) w_off[i] = (int)(n < g.N ? n : g.N - 1) * (int)g.K + w_ck[i] * 8;
    }
    const bool ups = CONVF ? UPS : (g.ups != 0);
    const int Hin = ups ? g.Hi * 2 : g.Hi, Win = ups ? g.Wi * 2 : g.Wi;
    // MODE 2: per-lane source pointer of the centre-less tap origin and a 9-bit mask of the taps that fall inside the image
    const unsigned char *a_ptr[AI];
    unsigned a_vm[AI];
    if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            a_ptr[i] = (const unsigned char *)g.A + (int64_t)a_off[i] * 2;
            unsigned vm = 0;
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                const int yi = a_y[i] + tp / 3, xi = a_x[i] + tp % 3;
                if (a_ok[i] && (unsigned)yi < (unsigned)Hin && (unsigned)xi < (unsigned)Win) vm |= 1u << tp;
            }
            a_vm[i] = vm;
        }
    }
    const int64_t wset = (LNV != 0 && g.w_set_rows > 0) ? m_base / g.w_set_rows : 0;          // weight set of this row tile (never straddles: host check)
    const unsigned char *Ab = (const unsigned char *)g.A, *Wb = (const unsigned char *)g.W + (LNV != 0 ? wset * g.w_set_stride * 2 : 0);
    const unsigned char *Zp = (const unsigned char *)g.zeros;
    // k order of the fast convs (round 6).  The K axis of W is (tap, ci); the k-tile SEQUENCE j = kt0 + kt may walk it in either order:
    //   tap-inner (default, g.conv_korder = 1): j -> (ci slice j / 9, tap j % 9).  Nine consecutive k-tiles fetch the SAME 64-channel slice of
    //     pixels shifted by one tap: the A lines of a k-tile are L1 / L2 hits of the previous ones (reuse distance one k-tile).
    //   tap-outer (rounds 1-5, conv_korder = 0): j -> (tap j / (Cin / 64), ci slice j % (Cin / 64)): a line is touched again only after a sweep over
    //     the workgroup's whole activation footprint (Cin / 64 k-tiles), per XCD a cyclic sweep over ~4 MB: the LRU worst case
    //     (scripts/ubench/gemm_loop.hip, profiles/r06_gemm_loop_conv_order.txt: 644 -> 864 TF/s at Cin = 320 on the same loop).
    // The products summed per output are the same set; only the summation order differs.
    const bool tap_inner = CONVF && g.conv_korder != 0;
    int ld_tap = 0, ld_ci = 0;
    if (CONVF && kt0 > 0) {
        if (tap_inner) { ld_tap = kt0 % 9; ld_ci = (kt0 / 9) * BK; }
        else { ld_tap = (kt0 * BK) / g.Cin; ld_ci = kt0 * BK - ld_tap * g.Cin; }
    }

    // DMA of one k-tile, split into per-instruction pieces so that the main loop can place them between MFMAs.
    struct TileSrc { int kb, dy_u, dx_u, tap_off, tap; unsigned sbase; };
    auto issue_begin = [&](int kt, int stage) __attribute__((always_inline)) -> TileSrc {
        TileSrc t;
        t.kb = (kt0 + kt) * BK; t.dy_u = 0; t.dx_u = 0; t.tap_off = 0; t.tap = 0;
        if (CONVF) {
            t.tap = ld_tap;
            t.kb = __builtin_amdgcn_readfirstlane(ld_tap * g.Cin + ld_ci);          // W column of this k-tile (= (kt0 + kt) * BK in the tap-outer order)
            t.dy_u = ld_tap / 3; t.dx_u = ld_tap - t.dy_u * 3;
            t.tap_off = UPS ? ld_ci : (t.dy_u * g.Wi + t.dx_u) * g.Cin + ld_ci;
            if (tap_inner) { if (++ld_tap == 9) { ld_tap = 0; ld_ci += BK; } }
            else { ld_ci += BK; if (ld_ci >= g.Cin) { ld_ci -= g.Cin; ++ld_tap; } }
        }
        t.sbase = lds0 + stage * STAGE;
        return t;
    };
    auto issue_a = [&](const TileSrc &t, int i) __attribute__((always_inline)) {
        const unsigned dst = t.sbase + (unsigned)((wid + 8 * i) * 1024);
        if (MODE == 3) { glds16_s(Ab + (size_t)t.kb * 2, (unsigned)(a_off[i] * 2), dst); return; }
        if (MODE == 2) {     // tap validity from the precomputed mask, address = lane pointer + uniform tap offset
            const bool okm = (a_vm[i] >> t.tap) & 1u;
            glds16(okm ? a_ptr[i] + (int64_t)t.tap_off * 2 : Zp, dst);
            return;
        }
        bool ok;
        int off;
        if (MODE == 4) {
            int yi = a_y[i] + t.dy_u, xi = a_x[i] + t.dx_u;
            ok = a_ok[i] && (unsigned)yi < (unsigned)Hin && (unsigned)xi < (unsigned)Win;
            off = a_off[i] + t.tap_off;
            off += ((yi >> 1) * g.Wi + (xi >> 1)) * g.Cin;
        } else if (MODE == 1) {
            const int k0 = t.kb + a_ck[i] * 8;
            const int kc = k0 < (int)g.K ? k0 : 0;
            const int tap = kc / g.Cin;
            const int ci = kc - tap * g.Cin;
            const int dy = tap / 3, dx = tap - dy * 3;
            int yi = a_y[i] + dy, xi = a_x[i] + dx;
            ok = a_ok[i] && k0 < (int)g.K && (unsigned)yi < (unsigned)Hin && (unsigned)xi < (unsigned)Win;
            if (g.ups) { yi >>= 1; xi >>= 1; }
            off = a_off[i] + (yi * g.Wi + xi) * g.Cin + ci;
        } else {
            ok = a_ok[i] && (t.kb + a_ck[i] * 8) < (int)g.K;
            off = a_off[i] + t.kb;
        }
        const unsigned char *src = ok ? Ab + (size_t)(unsigned)(off * 2) : Zp;
        glds16(src, dst);
    };
    // has_w(i): W group i of this wave exists (the last group only for the waves with wid + 8 (WI-1) < WG: "w3" waves)
    auto issue_w = [&](const TileSrc &t, int i, bool has) __attribute__((always_inline)) {
        if (has) {
            const unsigned dst = t.sbase + (unsigned)(BM * 128 + (wid + 8 * i) * 1024);
            if (MODE == 3 || CONVF) { glds16_s(Wb + (size_t)t.kb * 2, (unsigned)(w_off[i] * 2), dst); return; }   // K % 64 == 0: rows past N re-read row 0
            const bool ok = w_ok[i] && (t.kb + w_ck[i] * 8) < (int)g.K;
            const unsigned char *src = ok ? Wb + (size_t)(unsigned)((w_off[i] + t.kb) * 2) : Zp;
            glds16(src, dst);
        }
    };
    auto issue = [&](int kt, int stage) __attribute__((always_inline)) {
        const TileSrc t = issue_begin(kt, stage);
#pragma unroll
        for (int i = 0; i < AI; ++i) issue_a(t, i);
#pragma unroll
        for (int i = 0; i < WI; ++i) issue_w(t, i, wid + 8 * i < WG);
    };

    f32x4 acc[NTW][MT];
#pragma unroll
    for (int a = 0; a < NTW; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fc = lane >> 4;
    const int swz = (fr >> 1) & 7;
    const int fx0 = ((fc ^ swz) << 4), fx1 = (((fc + 4) ^ swz) << 4);
    const int aw0 = BM * 128 + (wn * (16 * NTW) + fr) * 128, aa0 = (wm * (16 * MT) + fr) * 128;
    struct Frag { uint4 w[NTW], a[MT]; };
    auto load_frag = [&](Frag &f, int stage, int ks) __attribute__((always_inline)) {
        const unsigned char *sb = smem + stage * STAGE;
        const int fx = ks ? fx1 : fx0;
#pragma unroll
        for (int t = 0; t < NTW; ++t) f.w[t] = *reinterpret_cast<const uint4 *>(sb + aw0 + fx + t * 2048);
#pragma unroll
        for (int t = 0; t < MT; ++t) f.a[t] = *reinterpret_cast<const uint4 *>(sb + aa0 + fx + t * 2048);
    };
    // one MFMA block (NTW x MT MFMAs on the fragments of one k-half).  After the FIRST MFMA the fragment reads of the next
    // block are issued (the s_waitcnt for this block's operands then never waits behind fresh reads); with DMA = true the LDS-DMA
    // instructions of a later k-tile (and their address arithmetic) are spread between the remaining MFMAs, where a few VALU /
    // SALU instructions per MFMA issue for free.
    constexpr int NMM = NTW * MT, NPC = AI + WI;
    auto block = [&](const Frag &f, Frag &fn, int st_next, int ks_next, bool load_next, const TileSrc &t, auto dma_tag, auto w3_tag) __attribute__((always_inline)) {
        constexpr bool DMA = decltype(dma_tag)::value, W3 = decltype(w3_tag)::value;
        static_for<0, NMM>([&](auto m_) __attribute__((always_inline)) {
            constexpr int m = decltype(m_)::value, nt = m / MT, mt = m % MT;
            acc[nt][mt] = T::mfma(f.w[nt], f.a[mt], acc[nt][mt]);
            if constexpr (m == 0) {
                __builtin_amdgcn_sched_barrier(0);
                if (load_next) load_frag(fn, st_next, ks_next);
                __builtin_amdgcn_sched_barrier(0);
            } else if constexpr (DMA && m >= 2) {
                // pieces p = 0 .. NPC-1 after MFMA index 2 + p * (NMM - 3) / NPC
                static_for<0, NPC>([&](auto p_) __attribute__((always_inline)) {
                    constexpr int pp = decltype(p_)::value;
                    if constexpr (m == 2 + (pp * (NMM - 3)) / NPC) {
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (pp < AI) issue_a(t, pp); else issue_w(t, pp - AI, (pp - AI) < WI - 1 || W3 || WG % 8 == 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
            }
        });
    };

    // Software pipeline over k-tiles (3 LDS stages, LDS-DMA two tiles ahead; fragment registers double-buffered per k half):
    //   [ MFMAs on F0 = (kt, half 0) | read F1 <- (kt, half 1) ]  wait(tile kt+1 landed) + barrier
    //   [ MFMAs on F1 | read F0 <- (kt+1, half 0) | DMA tile kt+3 -> stage of kt ]
    // ONE barrier per k-tile, sitting between two MFMA blocks whose operands are already in registers; every ds_read has a full
    // MFMA block (16-20 x 16 clk) to land.  Counted vmcnt: tile kt+2 stays in flight across the barrier.
    const bool w3 = (wid + 8 * (WI - 1)) < WG;     // this wave owns the last W group (instructions per tile are wave-uniform)
    Frag f0, f1;
    const TileSrc tnone = {0, 0, 0, 0, 0, 0u};
    // The whole k loop is instantiated twice (W3 = this wave issues WI / WI-1 W loads per tile) so that the counted waits and the
    // DMA pieces carry no run-time branches; the steady state (tiles kt+1 .. kt+3 exist) is a branch-free loop, the last three
    // k-tiles run through the generic tail.
    auto run = [&](auto w3_tag) __attribute__((always_inline)) {
        constexpr bool W3 = decltype(w3_tag)::value;
        constexpr int GRPW = AI + ((W3 || WG % 8 == 0) ? WI : WI - 1);        // LDS-DMA instructions of this wave per k-tile
        static_assert(2 * GRPW < 64, "vmcnt range");
        auto wait_tiles = [&](auto n_) __attribute__((always_inline)) {        // n tiles may stay in flight
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(decltype(n_)::value * GRPW) : "memory");
        };
        issue(0, 0);
        if (nk > 1) issue(1, 1);
        if (nk > 2) issue(2, 2);
        // lean LayerNorm-folded consumer: the rows' statistics are summed over the producer's slabs HERE, under the first k-tiles' flight (the
        // prologue ends in vmcnt(0): it waits for those tiles too, which the loop would do next anyway)
        if constexpr (LNIN) row_stats_prologue<BM>(g, m_base, srow);
        if (nk > 2) wait_tiles(std::integral_constant<int, 2>{}); else if (nk > 1) wait_tiles(std::integral_constant<int, 1>{}); else wait_tiles(std::integral_constant<int, 0>{});
        __builtin_amdgcn_s_barrier();
        load_frag(f0, 0, 0);
        int st = 0, kt = 0;                   // st = stage of tile kt
        for (; kt + 3 < nk; ++kt) {
            const int st1 = st + 1 == NS ? 0 : st + 1;
            block(f0, f1, st, 1, true, tnone, std::false_type{}, w3_tag);
            wait_tiles(std::integral_constant<int, 1>{});
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of stage st are complete
            __builtin_amdgcn_s_barrier();    // tile kt+1 is in LDS for every wave; every wave finished reading stage st
            const TileSrc t = issue_begin(kt + 3, st);
            block(f1, f0, st1, 0, true, t, std::true_type{}, w3_tag);
            st = st1;
        }
        for (; kt < nk; ++kt) {
            const int st1 = st + 1 == NS ? 0 : st + 1;
            block(f0, f1, st, 1, true, tnone, std::false_type{}, w3_tag);
            if (kt + 1 < nk) {
                if (kt + 2 < nk) wait_tiles(std::integral_constant<int, 1>{}); else wait_tiles(std::integral_constant<int, 0>{});
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                block(f1, f0, st1, 0, true, tnone, std::false_type{}, w3_tag);
            } else {
                block(f1, f0, st1, 0, false, tnone, std::false_type{}, w3_tag);
            }
            st = st1;
        }
    };
    if (w3) run(std::true_type{}); else run(std::false_type{});

    if constexpr (FUSE) {
        wave_epilogue<T, NTW, MT, 512>(g, acc, m_base, m_base + wm * (16 * MT), n_base + wn * (16 * NTW), lane, smem,
                                       (g.row_stats && g.splits == 1) ? srow : nullptr, slice);
        return;
    }
    // ---- epilogue (same math as k_gemm; MT m-tiles per wave)
    const int64_t n_lane = n_base + wn * (16 * NTW) + fc * 4;
    if (g.splits > 1) {   // partial sums of this k-slice: plain 16-byte stores into slab `slice`
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int64_t m = m_base + wm * (16 * MT) + mt * 16 + fr;
            if (m >= g.M) continue;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int64_t n = n_lane + nt * 16;
                if (n < g.N)
                    *reinterpret_cast<float4 *>(g.ws + ((int64_t)slice * g.M + m) * g.N + n) =
                        make_float4(acc[nt][mt][0], acc[nt][mt][1], acc[nt][mt][2], acc[nt][mt][3]);
            }
        }
        return;
    }
    if constexpr (LNV == 3) {
        // ---- softmax-heads epilogue (LayerNorm-folded scores -> probabilities)
        const float *biasp = g.bias + wset * g.N, *csp = g.colsum + wset * g.N;
        float4 bia[NTW], csm[NTW];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int64_t n = n_lane + nt * 16, nc = n < g.N ? n : g.N - 4;
            bia[nt] = *reinterpret_cast<const float4 *>(biasp + nc);
            csm[nt] = *reinterpret_cast<const float4 *>(csp + nc);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int64_t m = m_base + wm * (16 * MT) + mt * 16 + fr;
            const float2 st = *reinterpret_cast<const float2 *>(srow + 2 * (wm * (16 * MT) + mt * 16 + fr));
            const float mean = st.x * g.ln_inv_k, rstd = rsqrtf(fmaxf(st.y * g.ln_inv_k - mean * mean, 0.f) + g.ln_eps);
            float v[NTW][4];
            float mx = -3.0e38f;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                v[nt][0] = rstd * (acc[nt][mt][0] - mean * csm[nt].x) + bia[nt].x; v[nt][1] = rstd * (acc[nt][mt][1] - mean * csm[nt].y) + bia[nt].y;
                v[nt][2] = rstd * (acc[nt][mt][2] - mean * csm[nt].z) + bia[nt].z; v[nt][3] = rstd * (acc[nt][mt][3] - mean * csm[nt].w) + bia[nt].w;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool key = nt * 16 + fc * 4 + r < g.sm_keys;          // column inside the head block = text key index
                    v[nt][r] = key ? v[nt][r] : -3.0e38f;
                    mx = fmaxf(mx, v[nt][r]);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64)); mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sum = 0.f;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[nt][r] = exp2f(v[nt][r] - mx); sum += v[nt][r]; }     // (the scores carry log2(e) / sqrt(D): folded into Wq)
            sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);
            const float inv = 1.f / sum;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int64_t n = n_lane + nt * 16;
                if (m < g.M && n < g.N)
                    *reinterpret_cast<uint2 *>((unsigned char *)g.out + (m * g.ldc + n) * 2) =
                        make_uint2(pack2<T>(v[nt][0] * inv, v[nt][1] * inv), pack2<T>(v[nt][2] * inv, v[nt][3] * inv));
            }
        }
        return;
    }
    if constexpr (LEAN) {
        // ---- lean epilogue.  EVERY operand load is issued first, branch-free (rows / columns past the edge re-read a valid element,
        // absent operands read the zero page), then the tile is computed and stored.  The generic epilogue loads bias / row-vector /
        // residual under per-lane branches inside the m loop: at every control-flow join hipcc falls back to s_waitcnt vmcnt(0), and on
        // gfx9 that counter also holds the STORES in flight -- MT x NTW serialised store round trips per wave (seen in the ISA).
        const unsigned char *zp = (const unsigned char *)g.zeros;
        const bool has_b = g.bias != nullptr, has_rv = g.rowvec != nullptr, has_res = g.residual != nullptr;
        const float *biasp = has_b ? g.bias + (LNV != 0 ? wset * g.N : 0) : reinterpret_cast<const float *>(zp);
        const float *rvp = has_rv ? g.rowvec : reinterpret_cast<const float *>(zp);
        const unsigned char *resp = has_res ? (const unsigned char *)g.residual : zp;
        // the row vector is per batch: a tile holds rows of at most two batches (rows_per_batch >= BM, checked by the launcher)
        const int64_t bA = m_base / g.rows_per_batch, b_last = (g.M - 1) / g.rows_per_batch;
        const int64_t bB = bA + 1 < b_last ? bA + 1 : b_last;
        const int64_t m_rv = (bA + 1) * g.rows_per_batch;                       // first row of batch bB
        float4 bia[NTW], rvA[NTW], rvB[NTW];
        uint2 rs[MT][NTW];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int64_t n = n_lane + nt * 16, nc = n < g.N ? n : g.N - 4;
            bia[nt] = *reinterpret_cast<const float4 *>(biasp + (has_b ? nc : 0));
            rvA[nt] = *reinterpret_cast<const float4 *>(rvp + (has_rv ? bA * g.ld_rowvec + nc : 0));
            rvB[nt] = *reinterpret_cast<const float4 *>(rvp + (has_rv ? bB * g.ld_rowvec + nc : 0));
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int64_t m = m_base + wm * (16 * MT) + mt * 16 + fr, mc = m < g.M ? m : g.M - 1;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int64_t n = n_lane + nt * 16, nc = n < g.N ? n : g.N - 4;
                rs[mt][nt] = *reinterpret_cast<const uint2 *>(resp + (has_res ? (mc * g.ldr + nc) * 2 : 0));
            }
        }
        float4 csm[LNV == 2 ? NTW : 1];
        if constexpr (LNV == 2) {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int64_t n = n_lane + nt * 16, nc = n < g.N ? n : g.N - 4;
                csm[nt] = *reinterpret_cast<const float4 *>(g.colsum + wset * g.N + nc);
            }
        }
        uint2 pks[CS ? MT : 1][CS ? NTW : 1];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int64_t m = m_base + wm * (16 * MT) + mt * 16 + fr;
            const bool second = m >= m_rv;
            float mean = 0.f, rstd = 1.f;
            if constexpr (LNV == 2) {       // the row's statistics, summed over the producer's column slabs by row_stats_prologue
                const float2 st = *reinterpret_cast<const float2 *>(srow + 2 * (wm * (16 * MT) + mt * 16 + fr));
                mean = st.x * g.ln_inv_k;
                rstd = rsqrtf(fmaxf(st.y * g.ln_inv_k - mean * mean, 0.f) + g.ln_eps);
            }
            float rsum = 0.f, rsq = 0.f;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int64_t n = n_lane + nt * 16;
                const float4 rv = second ? rvB[nt] : rvA[nt];
                float a0 = acc[nt][mt][0], a1 = acc[nt][mt][1], a2 = acc[nt][mt][2], a3 = acc[nt][mt][3];
                if constexpr (LNV == 2) {
                    a0 = rstd * (a0 - mean * csm[nt].x); a1 = rstd * (a1 - mean * csm[nt].y);
                    a2 = rstd * (a2 - mean * csm[nt].z); a3 = rstd * (a3 - mean * csm[nt].w);
                }
                float v[4] = {a0 + bia[nt].x + rv.x, a1 + bia[nt].y + rv.y, a2 + bia[nt].z + rv.z, a3 + bia[nt].w + rv.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= g.out_scale;
                v[0] += T::to_f((unsigned short)(rs[mt][nt].x & 0xffff)); v[1] += T::to_f((unsigned short)(rs[mt][nt].x >> 16));
                v[2] += T::to_f((unsigned short)(rs[mt][nt].y & 0xffff)); v[3] += T::to_f((unsigned short)(rs[mt][nt].y >> 16));
                const uint2 pk = make_uint2(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]));
                const bool ok = m < g.M && n < g.N;
                if (ok) *reinterpret_cast<uint2 *>((unsigned char *)g.out + (m * g.ldc + n) * 2) = pk;
                if constexpr (CS) pks[mt][nt] = ok ? pk : make_uint2(0u, 0u);      // statistics of the values as STORED
                if constexpr (LNV == 1) {   // row partials of the values as STORED (columns past N contribute nothing)
                    const float w_ = n < g.N ? 1.f : 0.f;
                    const float t0 = w_ * T::to_f((unsigned short)(pk.x & 0xffff)), t1 = w_ * T::to_f((unsigned short)(pk.x >> 16));
                    const float t2 = w_ * T::to_f((unsigned short)(pk.y & 0xffff)), t3 = w_ * T::to_f((unsigned short)(pk.y >> 16));
                    rsum += (t0 + t1) + (t2 + t3); rsq += (t0 * t0 + t1 * t1) + (t2 * t2 + t3 * t3);
                }
            }
            if constexpr (LNV == 1) {       // the 4 lanes fc = 0..3 of a row hold disjoint column quads: combine, ONE plain store per (row, wave column)
                rsum += __shfl_xor(rsum, 16, 64); rsq += __shfl_xor(rsq, 16, 64);
                rsum += __shfl_xor(rsum, 32, 64); rsq += __shfl_xor(rsq, 32, 64);
                const int64_t slot = (n_base + wn * (16 * NTW)) / (16 * NTW);
                if (fc == 0 && m < g.M && n_base + wn * (16 * NTW) < g.N)
                    *reinterpret_cast<float2 *>(g.out_row_stats + (slot * g.M + m) * 2) = make_float2(rsum, rsq);
            }
        }
        if constexpr (CS) {
            // Statistics pass, ONE copy of the code (the first version flushed at every batch boundary inside the unrolled m loop: MT + 1
            // copies of 40 DPP reductions, +3 000 instructions and +5.5 us per launch of pure instruction fetch): pass p sums the rows of batch
            // slot p -- slot 0 = the batch of the tile's first row, slot 1 = the next one, present only in a tile that straddles a batch
            // boundary (a 16-row m-tile never does: rows_per_batch % 16 == 0; BM <= rows_per_batch: at most one boundary).
            const int64_t m_split = (m_base / g.rows_per_batch + 1) * g.rows_per_batch;          // first row of the next batch
            const int64_t m_end = m_base + BM < g.M ? m_base + BM : g.M;
            const int npass = m_split < m_end ? 2 : 1;
    #pragma unroll 1
            for (int pass = 0; pass < npass; ++pass) {
                float cs[NTW][4], cq[NTW][4];
    #pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
    #pragma unroll
                    for (int r = 0; r < 4; ++r) { cs[nt][r] = 0.f; cq[nt][r] = 0.f; }
    #pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int64_t m_tile = m_base + wm * (16 * MT) + mt * 16;             // wave-uniform
                    const float wgt = ((m_tile >= m_split ? 1 : 0) == pass) ? 1.f : 0.f;
    #pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) {
                        const uint2 pk = pks[mt][nt];
                        // (select, not multiply: an inf / NaN in the neighbouring batch's rows must not reach this batch's sums through 0 * inf)
                        const float t0 = wgt != 0.f ? T::to_f((unsigned short)(pk.x & 0xffff)) : 0.f, t1 = wgt != 0.f ? T::to_f((unsigned short)(pk.x >> 16)) : 0.f;
                        const float t2 = wgt != 0.f ? T::to_f((unsigned short)(pk.y & 0xffff)) : 0.f, t3 = wgt != 0.f ? T::to_f((unsigned short)(pk.y >> 16)) : 0.f;
                        cs[nt][0] += t0; cq[nt][0] += t0 * t0; cs[nt][1] += t1; cq[nt][1] += t1 * t1;
                        cs[nt][2] += t2; cq[nt][2] += t2 * t2; cs[nt][3] += t3; cq[nt][3] += t3 * t3;
                    }
                }
                float *tb = ctab + (pass * BN + wn * (16 * NTW) + fc * 4) * 2;
    #pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
    #pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float sa = row16_sum(cs[nt][r]), sb = row16_sum(cq[nt][r]);
                        if (fr == 0) {
                            __hip_atomic_fetch_add(tb + (nt * 16 + r) * 2, sa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_fetch_add(tb + (nt * 16 + r) * 2 + 1, sb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the LDS adds; NOT the output stores still in flight
            __builtin_amdgcn_s_barrier();
            const int64_t b0 = m_base / g.rows_per_batch;
            const int cpg = g.gn_cpg, G = (int)(g.N / cpg);
            const int n_hi = (int)(n_base + BN < g.N ? n_base + BN : g.N);
            const int g_first = (int)n_base / cpg, ng = (n_hi - 1) / cpg - g_first + 1;        // groups that overlap this column tile (<= BN / cpg + 2)
            for (int i = tid; i < 2 * ng; i += 512) {
                const int slot = i >= ng ? 1 : 0, gg = g_first + i - slot * ng;
                if (slot == 1 && !(m_split < m_end)) continue;
                const int c_lo = gg * cpg > (int)n_base ? gg * cpg : (int)n_base, c_hi = (gg + 1) * cpg < n_hi ? (gg + 1) * cpg : n_hi;
                float s1 = 0.f, s2 = 0.f;
                for (int c = c_lo; c < c_hi; ++c) { const float2 t = *reinterpret_cast<const float2 *>(ctab + (slot * BN + c - (int)n_base) * 2); s1 += t.x; s2 += t.y; }
                const int64_t b = b0 + slot;
                const int64_t slab = mblk - (b * g.rows_per_batch) / BM;       // tiles are counted over all M rows
                const int half = gg * cpg < (int)n_base ? 1 : 0;
                *reinterpret_cast<float2 *>(g.chan_parts + ((((b * g.cp_nslab + slab) * G + gg) * 2 + half) * 2)) = make_float2(s1, s2);
            }
        }
    } else {
    float4 bia[NTW], csm[LNV == 2 ? NTW : 1];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int64_t n = n_lane + nt * 16;
        bia[nt] = (g.bias && n < g.N) ? *reinterpret_cast<const float4 *>(g.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (LNV == 2) csm[nt] = n < g.N ? *reinterpret_cast<const float4 *>(g.colsum + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int64_t m = m_base + wm * (16 * MT) + mt * 16 + fr;
        if (m >= g.M) continue;
        if constexpr (LNV == 2) {       // LayerNorm fold: normalise this m-tile's accumulators in place (gate columns of GEGLU included)
            const float2 st = *reinterpret_cast<const float2 *>(srow + 2 * (wm * (16 * MT) + mt * 16 + fr));
            const float mean = st.x * g.ln_inv_k, rstd = rsqrtf(fmaxf(st.y * g.ln_inv_k - mean * mean, 0.f) + g.ln_eps);
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                acc[nt][mt][0] = rstd * (acc[nt][mt][0] - mean * csm[nt].x); acc[nt][mt][1] = rstd * (acc[nt][mt][1] - mean * csm[nt].y);
                acc[nt][mt][2] = rstd * (acc[nt][mt][2] - mean * csm[nt].z); acc[nt][mt][3] = rstd * (acc[nt][mt][3] - mean * csm[nt].w);
            }
        }
        float4 rv[NTW];
        uint2 rs[NTW];
        const int64_t bidx = g.rowvec ? m / g.rows_per_batch : 0;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int64_t n = n_lane + nt * 16;
            const bool okn = n < g.N && !(!LEAN && g.geglu && (nt & 1));
            const int64_t on = (!LEAN && g.geglu) ? (n_base + wn * (16 * NTW) + nt * 16) / 2 + fc * 4 : n;
            rv[nt] = (g.rowvec && n < g.N) ? *reinterpret_cast<const float4 *>(g.rowvec + bidx * g.ld_rowvec + n) : make_float4(0.f, 0.f, 0.f, 0.f);
            rs[nt] = (g.residual && okn) ? *reinterpret_cast<const uint2 *>((const unsigned char *)g.residual + (m * g.ldr + on) * 2) : make_uint2(0u, 0u);
        }
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int64_t n = n_lane + nt * 16;
            if (n >= g.N) continue;
            if (!LEAN && g.geglu && (nt & 1)) continue;
            float v[4] = {acc[nt][mt][0] + bia[nt].x + rv[nt].x, acc[nt][mt][1] + bia[nt].y + rv[nt].y,
                          acc[nt][mt][2] + bia[nt].z + rv[nt].z, acc[nt][mt][3] + bia[nt].w + rv[nt].w};
            int64_t on = n;
            if (!LEAN && g.geglu) {
                constexpr int NP = NTW - 1;
                const int np = nt + 1 < NTW ? nt + 1 : NP;
                v[0] *= gelu_erf(acc[np][mt][0] + bia[np].x); v[1] *= gelu_erf(acc[np][mt][1] + bia[np].y);
                v[2] *= gelu_erf(acc[np][mt][2] + bia[np].z); v[3] *= gelu_erf(acc[np][mt][3] + bia[np].w);
                on = (n_base + wn * (16 * NTW) + nt * 16) / 2 + fc * 4;
            }
            if (!LEAN && g.act == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = silu(v[r]);
            } else if (!LEAN && g.act == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fminf(fmaxf(v[r] * 0.5f + 0.5f, 0.f), 1.f);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= g.out_scale;
            if (g.residual) {
                v[0] += T::to_f((unsigned short)(rs[nt].x & 0xffff)); v[1] += T::to_f((unsigned short)(rs[nt].x >> 16));
                v[2] += T::to_f((unsigned short)(rs[nt].y & 0xffff)); v[3] += T::to_f((unsigned short)(rs[nt].y >> 16));
            }
            const bool to_t = !LEAN && g.out_t && on >= g.t_col0;
            const uint2 pk = make_uint2(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]));
            if (g.out && !(to_t && g.t_col0 > 0)) {
                if (!LEAN && g.out_f32)
                    *reinterpret_cast<float4 *>((float *)g.out + m * g.ldc + on) = make_float4(v[0], v[1], v[2], v[3]);
                else
                    *reinterpret_cast<uint2 *>((unsigned char *)g.out + (m * g.ldc + on) * 2) = pk;
            }
            if (to_t) {
                const int64_t b = m / g.rows_per_batch, tok = m - b * g.rows_per_batch;
                unsigned short *o = (unsigned short *)g.out_t + b * g.t_batch_stride + tok;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[(on - g.t_col0 + r) * g.ldt] = T::from_f(v[r]);
            }
        }
    }
    }
}


// =====================================================================================================================
// k_gemm8p -- PERSISTENT form of k_gemm8 for multi-round short-K linears (MODE 3: K % 64 == 0; the GEGLU FF-up projections:
// M = 24576, K = 320, N = 2560 is 1920 tiles of 5 k-tiles each).  In k_gemm8 such a tile spends ~2 us in MFMAs and ~10 us in its
// launch slot, pipeline fill and epilogue.  Here one workgroup per CU walks tiles bid, bid + G, ...; when the k loop of a tile ends
// it issues the first three k-tiles of the NEXT tile (the LDS ring is free, the accumulators are not touched by the DMA) and only
// then runs the epilogue, so fill latency and epilogue overlap and there is no per-tile dispatch.
template <class T, int NTW, int MT, int LNV = 0>
__global__ __launch_bounds__(512, 1) void k_gemm8p(const GemmArgs g)
{
    static_assert(LNV == 0 || LNV == 2, "persistent kernel: plain or LayerNorm-folded consumer");
    constexpr int BM = 64 * MT, BN = 32 * NTW, STAGE = BM * 128 + BN * 128, NS = 3;
    constexpr int AG = BM / 8, WG = BN / 8, AI = AG / 8, WI = (WG + 7) / 8;
    static_assert(WG % 8 == 0, "every wave issues the same number of W loads");
    constexpr int GRPW = AI + WI;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int nbn = (int)((g.N + BN - 1) / BN);
    const int64_t ntiles = ((g.M + BM - 1) / BM) * nbn;
    const int nk = (int)(g.K / BK);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    float *srow = reinterpret_cast<float *>(smem + NS * STAGE);          // LNV 2: [BM][2] row sums of the current tile, behind the ring
    const int lr = lane >> 3, ls = lane & 7;
    const unsigned char *Ab = (const unsigned char *)g.A, *Wb = (const unsigned char *)g.W;
    // dispatch-order tile d -> XCD d % 8 (gridDim.x is a multiple of 8); remapped so that consecutive tiles share an XCD's L2
    auto tile_of = [&](int64_t d, int64_t &m_base, int64_t &n_base) __attribute__((always_inline)) {
        const int64_t xcd = d & 7, qq = ntiles >> 3, rr = ntiles & 7;
        const int64_t bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (d >> 3);
        int64_t mblk, nblk;
        dng::tile_coords(bid, (int)((g.M + BM - 1) / BM), nbn, g.pw, mblk, nblk);
        m_base = mblk * BM; n_base = nblk * BN;
    };
    int a_off[AI], w_off[WI];
    auto set_offsets = [&](int64_t m_base, int64_t n_base) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int row = (wid + 8 * i) * 8 + lr;
            const int64_t m = m_base + row;
            a_off[i] = (int)(m < g.M ? m : g.M - 1) * (int)g.lda + (ls ^ ((row >> 1) & 7)) * 8;
        }
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int row = (wid + 8 * i) * 8 + lr;
            const int64_t n = n_base + row;
            w_off[i] = (int)(n < g.N ? n : g.N - 1) * (int)g.K + (ls ^ ((row >> 1) & 7)) * 8;
        }
    };
    auto issue_a = [&](int kt, int stage, int i) __attribute__((always_inline)) {
        glds16_s(Ab + (size_t)kt * (BK * 2), (unsigned)(a_off[i] * 2), lds0 + stage * STAGE + (unsigned)((wid + 8 * i) * 1024));
    };
    auto issue_w = [&](int kt, int stage, int i) __attribute__((always_inline)) {
        glds16_s(Wb + (size_t)kt * (BK * 2), (unsigned)(w_off[i] * 2), lds0 + stage * STAGE + (unsigned)(BM * 128 + (wid + 8 * i) * 1024));
    };
    auto issue = [&](int kt, int stage) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < AI; ++i) issue_a(kt, stage, i);
#pragma unroll
        for (int i = 0; i < WI; ++i) issue_w(kt, stage, i);
    };
    const int fr = lane & 15, fc = lane >> 4;
    const int swz = (fr >> 1) & 7;
    const int fx0 = ((fc ^ swz) << 4), fx1 = (((fc + 4) ^ swz) << 4);
    const int aw0 = BM * 128 + (wn * (16 * NTW) + fr) * 128, aa0 = (wm * (16 * MT) + fr) * 128;
    struct Frag { uint4 w[NTW], a[MT]; };
    auto load_frag = [&](Frag &f, int stage, int ks) __attribute__((always_inline)) {
        const unsigned char *sb = smem + stage * STAGE;
        const int fx = ks ? fx1 : fx0;
#pragma unroll
        for (int t = 0; t < NTW; ++t) f.w[t] = *reinterpret_cast<const uint4 *>(sb + aw0 + fx + t * 2048);
#pragma unroll
        for (int t = 0; t < MT; ++t) f.a[t] = *reinterpret_cast<const uint4 *>(sb + aa0 + fx + t * 2048);
    };
    f32x4 acc[NTW][MT];
    constexpr int NMM = NTW * MT, NPC = AI + WI;
    auto block = [&](const Frag &f, Frag &fn, int st_next, int ks_next, bool load_next, int dma_kt, int dma_stage, auto dma_tag) __attribute__((always_inline)) {
        constexpr bool DMA = decltype(dma_tag)::value;
        static_for<0, NMM>([&](auto m_) __attribute__((always_inline)) {
            constexpr int m = decltype(m_)::value, nt = m / MT, mt = m % MT;
            acc[nt][mt] = T::mfma(f.w[nt], f.a[mt], acc[nt][mt]);
            if constexpr (m == 0) {
                __builtin_amdgcn_sched_barrier(0);
                if (load_next) load_frag(fn, st_next, ks_next);
                __builtin_amdgcn_sched_barrier(0);
            } else if constexpr (DMA && m >= 2) {
                static_for<0, NPC>([&](auto p_) __attribute__((always_inline)) {
                    constexpr int pp = decltype(p_)::value;
                    if constexpr (m == 2 + (pp * (NMM - 3)) / NPC) {
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (pp < AI) issue_a(dma_kt, dma_stage, pp); else issue_w(dma_kt, dma_stage, pp - AI);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
            }
        });
    };
    auto wait_tiles = [&](auto n_) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(decltype(n_)::value * GRPW) : "memory");
    };

    int64_t d = blockIdx.x, m_base, n_base;
    if (d >= ntiles) return;
    tile_of(d, m_base, n_base);
    set_offsets(m_base, n_base);
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    if (nk > 2) issue(2, 2);
    bool first = true;
    for (;;) {
#pragma unroll
        for (int a = 0; a < NTW; ++a)
#pragma unroll
            for (int b = 0; b < MT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        // ---- k loop (the first min(nk, 3) k-tiles of this tile are already in flight)
        if (first) { if (nk > 2) wait_tiles(std::integral_constant<int, 2>{}); else if (nk > 1) wait_tiles(std::integral_constant<int, 1>{}); else wait_tiles(std::integral_constant<int, 0>{}); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // prefetched under the previous epilogue (its stores are counted too)
        first = false;
        __builtin_amdgcn_s_barrier();
        Frag f0, f1;
        load_frag(f0, 0, 0);
        int st = 0, kt = 0;
        for (; kt + 3 < nk; ++kt) {
            const int st1 = st + 1 == NS ? 0 : st + 1;
            block(f0, f1, st, 1, true, 0, 0, std::false_type{});
            wait_tiles(std::integral_constant<int, 1>{});
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            block(f1, f0, st1, 0, true, kt + 3, st, std::true_type{});
            st = st1;
        }
        for (; kt < nk; ++kt) {
            const int st1 = st + 1 == NS ? 0 : st + 1;
            block(f0, f1, st, 1, true, 0, 0, std::false_type{});
            if (kt + 1 < nk) {
                if (kt + 2 < nk) wait_tiles(std::integral_constant<int, 1>{}); else wait_tiles(std::integral_constant<int, 0>{});
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                block(f1, f0, st1, 0, true, 0, 0, std::false_type{});
            } else {
                block(f1, f0, st1, 0, false, 0, 0, std::false_type{});
            }
            st = st1;
        }
        // ---- next tile: its first k-tiles go into the (now idle) ring before this tile's epilogue
        const int64_t cm = m_base, cn = n_base;
        d += gridDim.x;
        const bool more = d < ntiles;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                 // every wave has read its last fragments
        if constexpr (LNV == 2) {
            // LayerNorm fold: (sum, sum^2) of this tile's BM rows over the producer's column slabs -> LDS, 512 threads, 4 loads in flight each
            // (a per-lane slot loop in the epilogue is a chain of 10-20 dependent L2 round trips per m-tile: measured slower than the LayerNorm
            // launch it replaces).  Before the next tile's DMA: the prologue ends in vmcnt(0), which must not wait for that DMA.
            row_stats_prologue<BM>(g, cm, srow);
            __builtin_amdgcn_s_barrier();                             // every thread's LDS adds have landed before the epilogue reads
        }
        if (more) {
            tile_of(d, m_base, n_base);
            set_offsets(m_base, n_base);
            issue(0, 0);
            if (nk > 1) issue(1, 1);
            if (nk > 2) issue(2, 2);
        }
        // ---- epilogue of tile (cm, cn): bias, GEGLU / activation, scale, residual; 8-byte stores
        {
            const int64_t n_lane = cn + wn * (16 * NTW) + fc * 4;
            float4 bia[NTW], csm[LNV == 2 ? NTW : 1];
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int64_t n = n_lane + nt * 16;
                bia[nt] = (g.bias && n < g.N) ? *reinterpret_cast<const float4 *>(g.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (LNV == 2) csm[nt] = n < g.N ? *reinterpret_cast<const float4 *>(g.colsum + n) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            float2 st[LNV == 2 ? MT : 1];
            if constexpr (LNV == 2) {      // (summed over the producer's slabs by row_stats_prologue before the next tile's DMA was issued)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) st[mt] = *reinterpret_cast<const float2 *>(srow + 2 * (wm * (16 * MT) + mt * 16 + fr));
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int64_t m = cm + wm * (16 * MT) + mt * 16 + fr;
                if (m >= g.M) continue;
                if constexpr (LNV == 2) {
                    const float mean = st[mt].x * g.ln_inv_k, rstd = rsqrtf(fmaxf(st[mt].y * g.ln_inv_k - mean * mean, 0.f) + g.ln_eps);
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) {
                        acc[nt][mt][0] = rstd * (acc[nt][mt][0] - mean * csm[nt].x); acc[nt][mt][1] = rstd * (acc[nt][mt][1] - mean * csm[nt].y);
                        acc[nt][mt][2] = rstd * (acc[nt][mt][2] - mean * csm[nt].z); acc[nt][mt][3] = rstd * (acc[nt][mt][3] - mean * csm[nt].w);
                    }
                }
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    const int64_t n = n_lane + nt * 16;
                    if (n >= g.N) continue;
                    if (g.geglu && (nt & 1)) continue;
                    float v[4] = {acc[nt][mt][0] + bia[nt].x, acc[nt][mt][1] + bia[nt].y, acc[nt][mt][2] + bia[nt].z, acc[nt][mt][3] + bia[nt].w};
                    int64_t on = n;
                    if (g.geglu) {
                        constexpr int NP = NTW - 1;
                        const int np = nt + 1 < NTW ? nt + 1 : NP;
                        v[0] *= gelu_erf(acc[np][mt][0] + bia[np].x); v[1] *= gelu_erf(acc[np][mt][1] + bia[np].y);
                        v[2] *= gelu_erf(acc[np][mt][2] + bia[np].z); v[3] *= gelu_erf(acc[np][mt][3] + bia[np].w);
                        on = (cn + wn * (16 * NTW) + nt * 16) / 2 + fc * 4;
                    }
                    if (g.act == 1) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = silu(v[r]);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= g.out_scale;
                    if (g.residual) {
                        const uint2 rs = *reinterpret_cast<const uint2 *>((const unsigned char *)g.residual + (m * g.ldr + on) * 2);
                        v[0] += T::to_f((unsigned short)(rs.x & 0xffff)); v[1] += T::to_f((unsigned short)(rs.x >> 16));
                        v[2] += T::to_f((unsigned short)(rs.y & 0xffff)); v[3] += T::to_f((unsigned short)(rs.y >> 16));
                    }
                    *reinterpret_cast<uint2 *>((unsigned char *)g.out + (m * g.ldc + on) * 2) = make_uint2(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]));
                }
            }
        }
        if (!more) break;
    }
}

template <class T, int NTW, int MT, int LNV = 0>
void launch8p(const GemmArgs &g, int nwg, hipStream_t s)
{
    constexpr size_t lds = 3 * (64 * MT * 128 + 32 * NTW * 128) + (LNV == 2 ? 64 * MT * 8 : 0);
    static_assert(lds <= 160 * 1024, "LDS ring");
    static gc::AttrOnce once;
    gc::ensure_dynamic_lds(once, (const void *)k_gemm8p<T, NTW, MT, LNV>, (int)lds);
    hipLaunchKernelGGL((k_gemm8p<T, NTW, MT, LNV>), dim3(nwg), dim3(512), lds, s, g);
}

// =====================================================================================================================
// k_gemm8q -- the 8-wave LDS-DMA GEMM on OCP fp8 (e4m3) operands with the block-scaled MFMA v_mfma_scale_f32_16x16x128_f8f6f4
// (K = 128 per instruction, 2x the bf16 MFMA rate; MI355X_MICROARCH.md "Matrix cores").  BASELINE configs[3] "fp8 MFMA UNet path".
// Operands: Act e4m3 bytes [..][Cin_p] (NHWC, Cin_p % 128 == 0: one 3x3 tap per 128-byte k-tile) or [M][K] (linear, K % 128 == 0);
// W e4m3 bytes [N][K]; scales are E8M0 exponents: one per weight ROW (w_scale[n], constant over the row's 32-element blocks) and one for
// the whole activation tensor (a 3x3 window mixes pixels, so a per-pixel activation scale cannot be applied per k-block).
// A k-tile is 128 BYTES per row exactly as in the bf16 kernel (64 elements x 2 B), so the LDS image, the XOR swizzle and the LDS-DMA
// plan are the same in bytes; the tile is ONE MFMA k-step: lane (fr, fc) feeds the 32 bytes of k-block fc of row fr (two ds_read_b128).
// Pipeline: 3-stage ring two tiles ahead, one barrier per k-tile; tile kt+1's fragments are read and tile kt+3's DMA is issued between
// the MFMAs of tile kt.  Output / epilogue (bf16 or f16, all fusions) = wave_epilogue.
typedef __attribute__((ext_vector_type(8))) int i32x8;

template <class T, int MODE, int NTW, int MT, bool FUSE, bool CS = false>
__global__ __launch_bounds__(512, 1) void k_gemm8q(const GemmArgs g)
{
    static_assert(MODE == 2 || MODE == 3, "fp8 path: fast conv (2) and aligned linear (3)");
    constexpr int BM = 64 * MT, BN = 32 * NTW, STAGE = BM * 128 + BN * 128, NS = 3;
    constexpr int AG = BM / 8, WG = BN / 8, AI = AG / 8, WI = (WG + 7) / 8;
    constexpr int BKB = 128;                       // k-tile: 128 bytes = 128 fp8 elements per row
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int nbn = (int)((g.N + BN - 1) / BN);
    int64_t mblk, nblk;
    int slice;
    wg_tile(g, (int)((g.M + BM - 1) / BM), nbn, mblk, nblk, slice);
    const int64_t m_base = mblk * BM, n_base = nblk * BN;
    // k-slices (long-K problems on part-filled grids: the 16 x 16-map convolutions): slice `slice` covers k-tiles [k0, k0 + nk) and leaves an
    // fp32 partial slab for the split-K reduce kernels, exactly as k_gemm8 does
    const int nk_all = (int)(g.K / BKB);
    const int k0 = g.splits > 1 ? slice * g.tiles_per_split : 0;
    const int nk = g.splits > 1 ? (nk_all - k0 < g.tiles_per_split ? nk_all - k0 : g.tiles_per_split) : nk_all;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    float *srow = reinterpret_cast<float *>(smem + NS * STAGE);
    if constexpr (FUSE) {
        if (g.row_stats) row_stats_prologue<BM>(g, m_base, srow);
    }
    static_assert(!(FUSE && CS), "one statistics epilogue at a time");
    float *ctab = reinterpret_cast<float *>(smem + NS * STAGE);          // CS: [2 batch slots][BN][2] channel sums of this tile, behind the ring
    if constexpr (CS) {      // zeroed here: the k loop's barriers order it before the epilogue's LDS adds
        for (int i = tid; i < 4 * BN; i += 512) ctab[i] = 0.f;
    }
    const int lr = lane >> 3, ls = lane & 7;
    // ---- per-lane DMA plan (bytes)
    int a_off[AI];
    unsigned a_vm[AI];
    const unsigned char *a_ptr[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int row = (wid + 8 * i) * 8 + lr;
        const int ck = ls ^ ((row >> 1) & 7);
        const int64_t m = m_base + row;
        const bool ok = m < g.M;
        a_vm[i] = 0; a_ptr[i] = (const unsigned char *)g.A; a_off[i] = 0;
        if (MODE == 2) {
            const int mm = ok ? (int)m : 0, hw = g.Ho * g.Wo, b = mm / hw, rem = mm - b * hw, oy = rem / g.Wo;
            const int y0 = oy * g.stride - g.pad, x0 = (rem - oy * g.Wo) * g.stride - g.pad;
            a_ptr[i] = (const unsigned char *)g.A + ((int64_t)(b * g.Hi + y0) * g.Wi + x0) * g.Cin + ck * 16;
            unsigned vm = 0;
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                const int yi = y0 + tp / 3, xi = x0 + tp % 3;
                if (ok && (unsigned)yi < (unsigned)g.Hi && (unsigned)xi < (unsigned)g.Wi) vm |= 1u << tp;
            }
            a_vm[i] = vm;
        } else {
            a_off[i] = (int)(m < g.M ? m : g.M - 1) * (int)g.lda + ck * 16;     // rows past M re-read the last row (never stored)
        }
    }
    int w_off[WI];
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const int row = (wid + 8 * i) * 8 + lr;
        const int64_t n = n_base + row;
        w_off[i] = (int)(n < g.N ? n : g.N - 1) * (int)g.K + (ls ^ ((row >> 1) & 7)) * 16;
    }
    const unsigned char *Ab = (const unsigned char *)g.A, *Wb = (const unsigned char *)g.W, *Zp = (const unsigned char *)g.zeros;
    // (k order of the fast convs: tap-inner by default, as in k_gemm8 -- see the comment there; g.conv_korder = 0 restores tap-outer)
    const bool tap_inner = MODE == 2 && g.conv_korder != 0;
    int ld_tap = 0, ld_ci = 0;
    if (MODE == 2) {
        if (tap_inner) { ld_tap = k0 % 9; ld_ci = (k0 / 9) * BKB; }
        else { ld_tap = (k0 * BKB) / g.Cin; ld_ci = (k0 * BKB) % g.Cin; }
    }
    struct TileSrc { int kb, tap, tap_off; unsigned sbase; };
    auto issue_begin = [&](int kt, int stage) __attribute__((always_inline)) -> TileSrc {
        TileSrc t;
        t.kb = (k0 + kt) * BKB; t.tap = 0; t.tap_off = 0;
        if (MODE == 2) {
            t.tap = ld_tap;
            t.kb = __builtin_amdgcn_readfirstlane(ld_tap * g.Cin + ld_ci);
            const int dy = ld_tap / 3, dx = ld_tap - dy * 3;
            t.tap_off = (dy * g.Wi + dx) * g.Cin + ld_ci;
            if (tap_inner) { if (++ld_tap == 9) { ld_tap = 0; ld_ci += BKB; } }
            else { ld_ci += BKB; if (ld_ci >= g.Cin) { ld_ci -= g.Cin; ++ld_tap; } }
        }
        t.sbase = lds0 + stage * STAGE;
        return t;
    };
    auto issue_a = [&](const TileSrc &t, int i) __attribute__((always_inline)) {
        const unsigned dst = t.sbase + (unsigned)((wid + 8 * i) * 1024);
        if (MODE == 3) { glds16_s(Ab + (size_t)t.kb, (unsigned)a_off[i], dst); return; }
        const bool okm = (a_vm[i] >> t.tap) & 1u;
        glds16(okm ? a_ptr[i] + (int64_t)t.tap_off : Zp, dst);
    };
    auto issue_w = [&](const TileSrc &t, int i, bool has) __attribute__((always_inline)) {
        if (has) glds16_s(Wb + (size_t)t.kb, (unsigned)w_off[i], t.sbase + (unsigned)(BM * 128 + (wid + 8 * i) * 1024));
    };
    auto issue = [&](int kt, int stage) __attribute__((always_inline)) {
        const TileSrc t = issue_begin(kt, stage);
#pragma unroll
        for (int i = 0; i < AI; ++i) issue_a(t, i);
#pragma unroll
        for (int i = 0; i < WI; ++i) issue_w(t, i, wid + 8 * i < WG);
    };

    f32x4 acc[NTW][MT];
#pragma unroll
    for (int a = 0; a < NTW; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fc = lane >> 4;
    const int swz = (fr >> 1) & 7;
    // Which two 16-byte chunks of the row's 128-byte k-tile lane group fc feeds is FREE (the k-tile is ONE MFMA k-step: the dot product runs
    // over every (lane group, byte) position, and A and W use the same association).  Chunks (fc, fc + 4) -- the bf16 kernel's pair -- make
    // the two ds_read_b128 conflict-free under this swizzle; the round-4 choice (2 fc, 2 fc + 1) put lane groups fc = 0 and 1 of a
    // ds_read_b128 lane group on the same 8 bank quartets: SQ_LDS_BANK_CONFLICT = 50 % of SQ_LDS_IDX_ACTIVE (profiles/r05_pmc_gemm8q.txt).
    const int fx0 = ((fc ^ swz) << 4), fx1 = (((fc + 4) ^ swz) << 4);
    const int aw0 = BM * 128 + (wn * (16 * NTW) + fr) * 128, aa0 = (wm * (16 * MT) + fr) * 128;
    // Fragment registers: 8-register MFMA operands assembled ONCE at load time from two ds_read_b128, double-buffered per k-tile (the two
    // buffers alternate by NAME in the 2x unrolled loop: an in-place reload would cost a register copy per fragment at the back edge).
    typedef __attribute__((ext_vector_type(4))) int i32x4;
    struct Frag { i32x8 w[NTW], a[MT]; };
    auto ld32 = [&](const unsigned char *p0, const unsigned char *p1) __attribute__((always_inline)) -> i32x8 {
        const i32x4 lo = *reinterpret_cast<const i32x4 *>(p0), hi = *reinterpret_cast<const i32x4 *>(p1);
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto load_frag = [&](Frag &f, int stage) __attribute__((always_inline)) {
        const unsigned char *sb = smem + stage * STAGE;
#pragma unroll
        for (int t = 0; t < NTW; ++t) f.w[t] = ld32(sb + aw0 + fx0 + t * 2048, sb + aw0 + fx1 + t * 2048);
#pragma unroll
        for (int t = 0; t < MT; ++t) f.a[t] = ld32(sb + aa0 + fx0 + t * 2048, sb + aa0 + fx1 + t * 2048);
    };
    // block scales: the weight row's exponent (constant over k) and the tensor-wide activation exponent, in byte 0 of the scale VGPRs.
    // These are the only compiler-visible global loads of the kernel: they are waited for HERE, before the first LDS-DMA is issued --
    // a compiler-placed vmcnt for them inside the k loop would count (and drain) the inline-asm DMA loads every iteration.
    int sw[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int64_t n = n_base + wn * (16 * NTW) + t * 16 + fr;
        sw[t] = g.w_scale[n < g.N ? n : g.N - 1];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int t = 0; t < NTW; ++t) asm volatile("" : "+v"(sw[t]));
    const int sa = g.a_scale;
    constexpr int NMM = NTW * MT, NPC = AI + WI;
    auto block = [&](const Frag &f, Frag &fn, int st_next, bool load_next, const TileSrc &t, auto dma_tag, auto w3_tag) __attribute__((always_inline)) {
        constexpr bool DMA = decltype(dma_tag)::value, W3 = decltype(w3_tag)::value;
        static_for<0, NMM>([&](auto m_) __attribute__((always_inline)) {
            constexpr int m = decltype(m_)::value, nt = m / MT, mt = m % MT;
            acc[nt][mt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(f.w[nt], f.a[mt], acc[nt][mt], 0, 0, 0, sw[nt], 0, sa);
            if constexpr (m == 0) {
                __builtin_amdgcn_sched_barrier(0);
                if (load_next) load_frag(fn, st_next);
                __builtin_amdgcn_sched_barrier(0);
            } else if constexpr (DMA && m >= 2) {
                static_for<0, NPC>([&](auto p_) __attribute__((always_inline)) {
                    constexpr int pp = decltype(p_)::value;
                    if constexpr (m == 2 + (pp * (NMM - 3)) / NPC) {
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (pp < AI) issue_a(t, pp); else issue_w(t, pp - AI, (pp - AI) < WI - 1 || W3 || WG % 8 == 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
            }
        });
    };
    const bool w3 = (wid + 8 * (WI - 1)) < WG;
    Frag f0, f1;
    const TileSrc tnone = {0, 0, 0, 0u};
    auto run = [&](auto w3_tag) __attribute__((always_inline)) {
        constexpr bool W3 = decltype(w3_tag)::value;
        constexpr int GRPW = AI + ((W3 || WG % 8 == 0) ? WI : WI - 1);
        auto wait_tiles = [&](auto n_) __attribute__((always_inline)) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(decltype(n_)::value * GRPW) : "memory");
        };
        issue(0, 0);
        if (nk > 1) issue(1, 1);
        if (nk > 2) issue(2, 2);
        if (nk > 2) wait_tiles(std::integral_constant<int, 2>{}); else if (nk > 1) wait_tiles(std::integral_constant<int, 1>{}); else wait_tiles(std::integral_constant<int, 0>{});
        __builtin_amdgcn_s_barrier();
        load_frag(f0, 0);
        int st = 0, kt = 0;                   // st = stage of tile kt (whose fragments are in f0 at the top of an even step)
        auto step = [&](Frag &fc_, Frag &fn_) __attribute__((always_inline)) {
            const int st1 = st + 1 == NS ? 0 : st + 1;
            if (kt + 3 < nk) {                // steady state: tiles kt+1, kt+2 in flight, tile kt+3 is issued into stage st
                wait_tiles(std::integral_constant<int, 1>{});
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                    // tile kt+1 landed for everyone; everyone holds tile kt in registers
                const TileSrc t = issue_begin(kt + 3, st);
                block(fc_, fn_, st1, true, t, std::true_type{}, w3_tag);
            } else if (kt + 1 < nk) {
                if (kt + 2 < nk) wait_tiles(std::integral_constant<int, 1>{}); else wait_tiles(std::integral_constant<int, 0>{});
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                block(fc_, fn_, st1, true, tnone, std::false_type{}, w3_tag);
            } else {
                block(fc_, fn_, st1, false, tnone, std::false_type{}, w3_tag);
            }
            st = st1; ++kt;
        };
        while (kt < nk) {
            step(f0, f1);
            if (kt < nk) step(f1, f0);
        }
    };
    if (w3) run(std::true_type{}); else run(std::false_type{});
    if constexpr (FUSE) {
        wave_epilogue<T, NTW, MT, 512>(g, acc, m_base, m_base + wm * (16 * MT), n_base + wn * (16 * NTW), lane, smem, g.row_stats ? srow : nullptr, slice);
        return;
    }
    // ---- epilogue (same math as k_gemm; MT m-tiles per wave)
    const int64_t n_lane = n_base + wn * (16 * NTW) + fc * 4;
    if (g.splits > 1) {   // partial sums of this k-slice: plain 16-byte stores into slab `slice`
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int64_t m = m_base + wm * (16 * MT) + mt * 16 + fr;
            if (m >= g.M) continue;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int64_t n = n_lane + nt * 16;
                if (n < g.N)
                    *reinterpret_cast<float4 *>(g.ws + ((int64_t)slice * g.M + m) * g.N + n) =
                        make_float4(acc[nt][mt][0], acc[nt][mt][1], acc[nt][mt][2], acc[nt][mt][3]);
            }
        }
        return;
    }
    if constexpr (CS) {
        // ---- lean epilogue + channel partials (the CS epilogue of k_gemm8, same arithmetic and layout).  EVERY operand load is issued first, branch-free (rows / columns past the edge re-read a valid element,
        // absent operands read the zero page), then the tile is computed and stored.  The generic epilogue loads bias / row-vector /
        // residual under per-lane branches inside the m loop: at every control-flow join hipcc falls back to s_waitcnt vmcnt(0), and on
        // gfx9 that counter also holds the STORES in flight -- MT x NTW serialised store round trips per wave (seen in the ISA).
        const unsigned char *zp = (const unsigned char *)g.zeros;
        const bool has_b = g.bias != nullptr, has_rv = g.rowvec != nullptr, has_res = g.residual != nullptr;
        const float *biasp = has_b ? g.bias : reinterpret_cast<const float *>(zp);
        const float *rvp = has_rv ? g.rowvec : reinterpret_cast<const float *>(zp);
        const unsigned char *resp = has_res ? (const unsigned char *)g.residual : zp;
        // the row vector is per batch: a tile holds rows of at most two batches (rows_per_batch >= BM, checked by the launcher)
        const int64_t bA = m_base / g.rows_per_batch, b_last = (g.M - 1) / g.rows_per_batch;
        const int64_t bB = bA + 1 < b_last ? bA + 1 : b_last;
        const int64_t m_rv = (bA + 1) * g.rows_per_batch;                       // first row of batch bB
        float4 bia[NTW], rvA[NTW], rvB[NTW];
        uint2 rs[MT][NTW];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int64_t n = n_lane + nt * 16, nc = n < g.N ? n : g.N - 4;
            bia[nt] = *reinterpret_cast<const float4 *>(biasp + (has_b ? nc : 0));
            rvA[nt] = *reinterpret_cast<const float4 *>(rvp + (has_rv ? bA * g.ld_rowvec + nc : 0));
            rvB[nt] = *reinterpret_cast<const float4 *>(rvp + (has_rv ? bB * g.ld_rowvec + nc : 0));
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int64_t m = m_base + wm * (16 * MT) + mt * 16 + fr, mc = m < g.M ? m : g.M - 1;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int64_t n = n_lane + nt * 16, nc = n < g.N ? n : g.N - 4;
                rs[mt][nt] = *reinterpret_cast<const uint2 *>(resp + (has_res ? (mc * g.ldr + nc) * 2 : 0));
            }
        }
        uint2 pks[MT][NTW];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int64_t m = m_base + wm * (16 * MT) + mt * 16 + fr;
            const bool second = m >= m_rv;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int64_t n = n_lane + nt * 16;
                const float4 rv = second ? rvB[nt] : rvA[nt];
                float v[4] = {acc[nt][mt][0] + bia[nt].x + rv.x, acc[nt][mt][1] + bia[nt].y + rv.y,
                              acc[nt][mt][2] + bia[nt].z + rv.z, acc[nt][mt][3] + bia[nt].w + rv.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= g.out_scale;
                v[0] += T::to_f((unsigned short)(rs[mt][nt].x & 0xffff)); v[1] += T::to_f((unsigned short)(rs[mt][nt].x >> 16));
                v[2] += T::to_f((unsigned short)(rs[mt][nt].y & 0xffff)); v[3] += T::to_f((unsigned short)(rs[mt][nt].y >> 16));
                const uint2 pk = make_uint2(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]));
                const bool ok = m < g.M && n < g.N;
                if (ok) *reinterpret_cast<uint2 *>((unsigned char *)g.out + (m * g.ldc + n) * 2) = pk;
                pks[mt][nt] = ok ? pk : make_uint2(0u, 0u);      // statistics of the values as STORED
            }
        }
        {
            // Statistics pass, ONE copy of the code (the first version flushed at every batch boundary inside the unrolled m loop: MT + 1
            // copies of 40 DPP reductions, +3 000 instructions and +5.5 us per launch of pure instruction fetch): pass p sums the rows of batch
            // slot p -- slot 0 = the batch of the tile's first row, slot 1 = the next one, present only in a tile that straddles a batch
            // boundary (a 16-row m-tile never does: rows_per_batch % 16 == 0; BM <= rows_per_batch: at most one boundary).
            const int64_t m_split = (m_base / g.rows_per_batch + 1) * g.rows_per_batch;          // first row of the next batch
            const int64_t m_end = m_base + BM < g.M ? m_base + BM : g.M;
            const int npass = m_split < m_end ? 2 : 1;
    #pragma unroll 1
            for (int pass = 0; pass < npass; ++pass) {
                float cs[NTW][4], cq[NTW][4];
    #pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
    #pragma unroll
                    for (int r = 0; r < 4; ++r) { cs[nt][r] = 0.f; cq[nt][r] = 0.f; }
    #pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int64_t m_tile = m_base + wm * (16 * MT) + mt * 16;             // wave-uniform
                    const float wgt = ((m_tile >= m_split ? 1 : 0) == pass) ? 1.f : 0.f;
    #pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) {
                        const uint2 pk = pks[mt][nt];
                        // (select, not multiply: an inf / NaN in the neighbouring batch's rows must not reach this batch's sums through 0 * inf)
                        const float t0 = wgt != 0.f ? T::to_f((unsigned short)(pk.x & 0xffff)) : 0.f, t1 = wgt != 0.f ? T::to_f((unsigned short)(pk.x >> 16)) : 0.f;
                        const float t2 = wgt != 0.f ? T::to_f((unsigned short)(pk.y & 0xffff)) : 0.f, t3 = wgt != 0.f ? T::to_f((unsigned short)(pk.y >> 16)) : 0.f;
                        cs[nt][0] += t0; cq[nt][0] += t0 * t0; cs[nt][1] += t1; cq[nt][1] += t1 * t1;
                        cs[nt][2] += t2; cq[nt][2] += t2 * t2; cs[nt][3] += t3; cq[nt][3] += t3 * t3;
                    }
                }
                float *tb = ctab + (pass * BN + wn * (16 * NTW) + fc * 4) * 2;
    #pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
    #pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float sa = row16_sum(cs[nt][r]), sb = row16_sum(cq[nt][r]);
                        if (fr == 0) {
                            __hip_atomic_fetch_add(tb + (nt * 16 + r) * 2, sa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_fetch_add(tb + (nt * 16 + r) * 2 + 1, sb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the LDS adds; NOT the output stores still in flight
            __builtin_amdgcn_s_barrier();
            const int64_t b0 = m_base / g.rows_per_batch;
            const int cpg = g.gn_cpg, G = (int)(g.N / cpg);
            const int n_hi = (int)(n_base + BN < g.N ? n_base + BN : g.N);
            const int g_first = (int)n_base / cpg, ng = (n_hi - 1) / cpg - g_first + 1;        // groups that overlap this column tile (<= BN / cpg + 2)
            for (int i = tid; i < 2 * ng; i += 512) {
                const int slot = i >= ng ? 1 : 0, gg = g_first + i - slot * ng;
                if (slot == 1 && !(m_split < m_end)) continue;
                const int c_lo = gg * cpg > (int)n_base ? gg * cpg : (int)n_base, c_hi = (gg + 1) * cpg < n_hi ? (gg + 1) * cpg : n_hi;
                float s1 = 0.f, s2 = 0.f;
                for (int c = c_lo; c < c_hi; ++c) { const float2 t = *reinterpret_cast<const float2 *>(ctab + (slot * BN + c - (int)n_base) * 2); s1 += t.x; s2 += t.y; }
                const int64_t b = b0 + slot;
                const int64_t slab = mblk - (b * g.rows_per_batch) / BM;       // tiles are counted over all M rows
                const int half = gg * cpg < (int)n_base ? 1 : 0;
                *reinterpret_cast<float2 *>(g.chan_parts + ((((b * g.cp_nslab + slab) * G + gg) * 2 + half) * 2)) = make_float2(s1, s2);
            }
        }
        return;
    }
    float4 bia[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int64_t n = n_lane + nt * 16;
        bia[nt] = (g.bias && n < g.N) ? *reinterpret_cast<const float4 *>(g.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int64_t m = m_base + wm * (16 * MT) + mt * 16 + fr;
        if (m >= g.M) continue;
        float4 rv[NTW];
        uint2 rs[NTW];
        const int64_t bidx = g.rowvec ? m / g.rows_per_batch : 0;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int64_t n = n_lane + nt * 16;
            const bool okn = n < g.N && !(g.geglu && (nt & 1));
            const int64_t on = g.geglu ? (n_base + wn * (16 * NTW) + nt * 16) / 2 + fc * 4 : n;
            rv[nt] = (g.rowvec && n < g.N) ? *reinterpret_cast<const float4 *>(g.rowvec + bidx * g.ld_rowvec + n) : make_float4(0.f, 0.f, 0.f, 0.f);
            rs[nt] = (g.residual && okn) ? *reinterpret_cast<const uint2 *>((const unsigned char *)g.residual + (m * g.ldr + on) * 2) : make_uint2(0u, 0u);
        }
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int64_t n = n_lane + nt * 16;
            if (n >= g.N) continue;
            if (g.geglu && (nt & 1)) continue;
            float v[4] = {acc[nt][mt][0] + bia[nt].x + rv[nt].x, acc[nt][mt][1] + bia[nt].y + rv[nt].y,
                          acc[nt][mt][2] + bia[nt].z + rv[nt].z, acc[nt][mt][3] + bia[nt].w + rv[nt].w};
            int64_t on = n;
            if (g.geglu) {
                constexpr int NP = NTW - 1;
                const int np = nt + 1 < NTW ? nt + 1 : NP;
                v[0] *= gelu_erf(acc[np][mt][0] + bia[np].x); v[1] *= gelu_erf(acc[np][mt][1] + bia[np].y);
                v[2] *= gelu_erf(acc[np][mt][2] + bia[np].z); v[3] *= gelu_erf(acc[np][mt][3] + bia[np].w);
                on = (n_base + wn * (16 * NTW) + nt * 16) / 2 + fc * 4;
            }
            if (g.act == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = silu(v[r]);
            } else if (g.act == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fminf(fmaxf(v[r] * 0.5f + 0.5f, 0.f), 1.f);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= g.out_scale;
            if (g.residual) {
                v[0] += T::to_f((unsigned short)(rs[nt].x & 0xffff)); v[1] += T::to_f((unsigned short)(rs[nt].x >> 16));
                v[2] += T::to_f((unsigned short)(rs[nt].y & 0xffff)); v[3] += T::to_f((unsigned short)(rs[nt].y >> 16));
            }
            const bool to_t = g.out_t && on >= g.t_col0;
            if (g.out_fp8) {   // e4m3 bytes, saturated like gc_dn_groupnorm_apply_fp8 (4 columns = one 32-bit store)
                const float q0 = fminf(fmaxf(v[0] * g.out_qscale, -448.f), 448.f), q1 = fminf(fmaxf(v[1] * g.out_qscale, -448.f), 448.f);
                const float q2 = fminf(fmaxf(v[2] * g.out_qscale, -448.f), 448.f), q3 = fminf(fmaxf(v[3] * g.out_qscale, -448.f), 448.f);
                int w8 = 0;
                w8 = __builtin_amdgcn_cvt_pk_fp8_f32(q0, q1, w8, false); w8 = __builtin_amdgcn_cvt_pk_fp8_f32(q2, q3, w8, true);
                *reinterpret_cast<int *>((unsigned char *)g.out + m * g.ldc + on) = w8;
            } else if (g.out && !(to_t && g.t_col0 > 0)) {
                if (g.out_f32)
                    *reinterpret_cast<float4 *>((float *)g.out + m * g.ldc + on) = make_float4(v[0], v[1], v[2], v[3]);
                else
                    *reinterpret_cast<uint2 *>((unsigned char *)g.out + (m * g.ldc + on) * 2) = make_uint2(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]));
            }
            if (to_t) {
                const int64_t b = m / g.rows_per_batch, tok = m - b * g.rows_per_batch;
                unsigned short *o = (unsigned short *)g.out_t + b * g.t_batch_stride + tok;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[(on - g.t_col0 + r) * g.ldt] = T::from_f(v[r]);
            }
        }
    }
}

// The split-K reduce kernels add the slabs in slab order (the result does not depend on the slice -> workgroup assignment), ZC slabs per
// round trip: the loads of a chunk are all issued before the first add.  (Round 1..4: one slab per iteration = `splits` dependent round trips
// to the fabric, 11 us for the 15 slabs of an 8 x 8-map convolution.)
constexpr int ZC = 8;
// epilogue of a split-K problem: ws fp32 [M][N] -> out (same epilogue as the fused path; no GEGLU)
template <class T>
__global__ __launch_bounds__(256) void k_splitk_epilogue_plain(const GemmArgs g)
{
    const int64_t nq = g.N / 4;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < g.M * nq; q += (int64_t)gridDim.x * 256) {
        const int64_t m = q / nq, n = (q - m * nq) * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        // (the residual is fetched with the slabs, not after them: one fabric round trip less per launch)
        const uint2 rpre = g.residual ? *reinterpret_cast<const uint2 *>((const unsigned char *)g.residual + (m * g.ldr + n) * 2) : make_uint2(0u, 0u);
        for (int z0 = 0; z0 < g.splits; z0 += ZC) {          // ZC slab loads in flight, added in slab order (see ZC)
            float4 a[ZC];
#pragma unroll
            for (int u = 0; u < ZC; ++u)
                if (z0 + u < g.splits) a[u] = *reinterpret_cast<const float4 *>(g.ws + ((int64_t)(z0 + u) * g.M + m) * g.N + n);
#pragma unroll
            for (int u = 0; u < ZC; ++u)
                if (z0 + u < g.splits) { v[0] += a[u].x; v[1] += a[u].y; v[2] += a[u].z; v[3] += a[u].w; }
        }
        float gate[4] = {0.f, 0.f, 0.f, 0.f};
        epilogue_store<T>(g, m, n, n, v, gate, &rpre);
    }
}

// the same with output statistics / LayerNorm fold (FUSE problems): workgroup = 64 column quads (256 columns) x 4 row lanes over 16
// rows of ONE batch, 4 rows per thread with independent loads: coalesced 16-byte slab reads; row statistics by a wave sum
// (slot = blockIdx.x), group statistics via LDS.
template <class T>
__global__ __launch_bounds__(256) void k_splitk_epilogue(const GemmArgs g)
{
    __shared__ float red[GS_MAXG * 2];
    constexpr int RPB = 16;
    const int64_t nq = g.N / 4;
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int64_t cq = (int64_t)blockIdx.x * 64 + cl;
    const int64_t m0 = (int64_t)blockIdx.y * RPB;
    const bool okc = cq < nq;
    const int64_t n = (okc ? cq : 0) * 4;
    if (g.out_group_stats) {
        if (threadIdx.x < GS_MAXG * 2) red[threadIdx.x] = 0.f;
        __syncthreads();
    }
    float v[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i][0] = 0.f; v[i][1] = 0.f; v[i][2] = 0.f; v[i][3] = 0.f; }
    if (okc) {
        for (int z0 = 0; z0 < g.splits; z0 += ZC / 2) {           // 4 rows x ZC / 2 slabs: 16-byte loads in flight together, added in slab order
            float4 a[ZC / 2][4];
#pragma unroll
            for (int u = 0; u < ZC / 2; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int64_t m = m0 + rl + 4 * i;
                    if (z0 + u < g.splits && m < g.M) a[u][i] = *reinterpret_cast<const float4 *>(g.ws + ((int64_t)(z0 + u) * g.M + m) * g.N + n);
                }
#pragma unroll
            for (int u = 0; u < ZC / 2; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int64_t m = m0 + rl + 4 * i;
                    if (z0 + u < g.splits && m < g.M) { v[i][0] += a[u][i].x; v[i][1] += a[u][i].y; v[i][2] += a[u][i].z; v[i][3] += a[u][i].w; }
                }
        }
    }
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t m = m0 + rl + 4 * i;          // rl is wave-uniform: every lane of a wave works on the same rows
        if (m >= g.M) break;
        if (okc) {
            float gate[4] = {0.f, 0.f, 0.f, 0.f};
            epilogue_store<T>(g, m, n, n, v[i], gate);
        }
        if (g.out_row_stats) {
            const float a = wave_sum_f(okc ? (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]) : 0.f);
            const float b = wave_sum_f(okc ? (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]) : 0.f);
            if (cl == 0) *reinterpret_cast<float2 *>(g.out_row_stats + ((int64_t)blockIdx.x * g.M + m) * 2) = make_float2(a, b);
        }
        if (okc) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { cs[2 * r] += v[i][r]; cs[2 * r + 1] += v[i][r] * v[i][r]; }
        }
    }
    if (g.out_group_stats) {
        if (okc) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gi = (int)(n + r) / g.gn_cpg;
                __hip_atomic_fetch_add(red + 2 * gi, cs[2 * r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(red + 2 * gi + 1, cs[2 * r + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        __syncthreads();
        if (threadIdx.x < GS_MAXG * 2) {
            const float t = red[threadIdx.x];
            if (t != 0.f) unsafeAtomicAdd(g.out_group_stats + (m0 / g.rows_per_batch) * g.gn_groups * 2 + threadIdx.x, t);
        }
    }
}


// the plain reduce-epilogue + per-channel partial sums of the stored output (GemmArgs::chan_parts, slabs of CS_RB rows; rows_per_batch %
// CS_RB == 0 so a slab never straddles batches): workgroup = 16 column quads (64 columns) x 16 row lanes over CS_RB rows -- 2 rows per
// thread, all slab loads of a thread in flight together (the first version, 64 quads x 4 row lanes x 8 rows, ran 19.5 us against 7.9 us
// for the plain grid-stride kernel: a chain of dependent round trips on 240 workgroups)
constexpr int CS_RB = 32;
template <class T>
__global__ __launch_bounds__(256) void k_splitk_epilogue_cs(const GemmArgs g)
{
    __shared__ float red[16][16][8];
    constexpr int RPT = CS_RB / 16;
    const int64_t nq = g.N / 4;
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int64_t cq4 = (int64_t)blockIdx.x * 16 + cl;
    const int64_t m0 = (int64_t)blockIdx.y * CS_RB;
    const bool okc = cq4 < nq;
    const int64_t n = (okc ? cq4 : 0) * 4;
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float v[RPT][4];
#pragma unroll
    for (int i = 0; i < RPT; ++i) { v[i][0] = 0.f; v[i][1] = 0.f; v[i][2] = 0.f; v[i][3] = 0.f; }
    if (okc) {
        uint2 rpre[RPT];                                       // the residual rows travel with the slabs
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int64_t m = m0 + rl + 16 * i;
            rpre[i] = (g.residual && m < g.M) ? *reinterpret_cast<const uint2 *>((const unsigned char *)g.residual + (m * g.ldr + n) * 2) : make_uint2(0u, 0u);
        }
        for (int z0 = 0; z0 < g.splits; z0 += ZC) {            // RPT rows x ZC slabs in flight together, added in slab order
            float4 a[ZC][RPT];
#pragma unroll
            for (int u = 0; u < ZC; ++u)
#pragma unroll
                for (int i = 0; i < RPT; ++i) {
                    const int64_t m = m0 + rl + 16 * i;
                    if (z0 + u < g.splits && m < g.M) a[u][i] = *reinterpret_cast<const float4 *>(g.ws + ((int64_t)(z0 + u) * g.M + m) * g.N + n);
                }
#pragma unroll
            for (int u = 0; u < ZC; ++u)
#pragma unroll
                for (int i = 0; i < RPT; ++i) {
                    const int64_t m = m0 + rl + 16 * i;
                    if (z0 + u < g.splits && m < g.M) { v[i][0] += a[u][i].x; v[i][1] += a[u][i].y; v[i][2] += a[u][i].z; v[i][3] += a[u][i].w; }
                }
        }
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int64_t m = m0 + rl + 16 * i;
            if (m >= g.M) break;
            float gate[4] = {0.f, 0.f, 0.f, 0.f};
            epilogue_store<T>(g, m, n, n, v[i], gate, &rpre[i]);      // on return v holds the values as stored
#pragma unroll
            for (int r = 0; r < 4; ++r) { cs[2 * r] += v[i][r]; cs[2 * r + 1] += v[i][r] * v[i][r]; }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[rl][cl][j] = cs[j];
    __syncthreads();
    __shared__ float chs[64][2];
    if (threadIdx.x < 128) {           // thread = (column, value): sum over the 16 row lanes
        const int c2 = threadIdx.x >> 3, j = threadIdx.x & 7;
        float o = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) o += red[r][c2][j];
        chs[c2 * 4 + (j >> 1)][j & 1] = o;
    }
    __syncthreads();
    if (m0 < g.M) {
        const int cpg = g.gn_cpg, G = (int)(g.N / cpg);
        const int n_base = (int)blockIdx.x * 64, n_hi = n_base + 64 < (int)g.N ? n_base + 64 : (int)g.N;
        const int g_first = n_base / cpg, ng = (n_hi - 1) / cpg - g_first + 1;
        if ((int)threadIdx.x < ng) {
            const int gg = g_first + threadIdx.x;
            const int c_lo = gg * cpg > n_base ? gg * cpg : n_base, c_hi = (gg + 1) * cpg < n_hi ? (gg + 1) * cpg : n_hi;
            float s1 = 0.f, s2 = 0.f;
            for (int c = c_lo; c < c_hi; ++c) { s1 += chs[c - n_base][0]; s2 += chs[c - n_base][1]; }
            const int64_t b = m0 / g.rows_per_batch, slab = (m0 - b * g.rows_per_batch) / CS_RB;
            const int half = gg * cpg < n_base ? 1 : 0;
            *reinterpret_cast<float2 *>(g.chan_parts + ((((b * g.cp_nslab + slab) * G + gg) * 2 + half) * 2)) = make_float2(s1, s2);
        }
    }
}


inline bool fuse_of(const GemmArgs &g) { return g.row_stats || g.out_row_stats || g.out_group_stats; }

// ---- launch templates: instantiated with FUSE = false in dn_gemm_plain.hip and FUSE = true in dn_gemm_fuse.hip (one translation
// unit each: the kernel templates are large and the two halves compile in parallel)
template <class T, int MODE, int NTW, bool FUSE>
void launch4(const GemmArgs &g, dim3 grid, hipStream_t s)
{
    constexpr size_t lds = 2 * (BM * 128 + 32 * NTW * 128);
    static gc::AttrOnce once;      // per device, thread-safe (no function-local bool latch)
    gc::ensure_dynamic_lds(once, (const void *)k_gemm<T, MODE, NTW, FUSE>, (int)lds);
    hipLaunchKernelGGL((k_gemm<T, MODE, NTW, FUSE>), grid, dim3(NT), lds, s, g);
}

template <class T, int MODE, int NTW, int MT, bool FUSE, bool CS = false, bool LEAN = false, int LNV = 0>
void launch8(const GemmArgs &g, dim3 grid, hipStream_t s)
{
    // FUSE / LNV 2: + the row-sum array of row_stats_prologue; CS: + the [2][BN][2] channel-sum table
    constexpr size_t lds = 3 * (64 * MT * 128 + 32 * NTW * 128) + ((FUSE || LNV >= 2) ? 64 * MT * 8 : 0) + (CS ? 32 * NTW * 16 : 0);
    static_assert(lds <= 160 * 1024, "LDS ring");
    static gc::AttrOnce once;
    gc::ensure_dynamic_lds(once, (const void *)k_gemm8<T, MODE, NTW, MT, FUSE, CS, LEAN, LNV>, (int)lds);
    hipLaunchKernelGGL((k_gemm8<T, MODE, NTW, MT, FUSE, CS, LEAN, LNV>), grid, dim3(512), lds, s, g);
}

// ---- lean LayerNorm fold (round 5; dn_gemm_ln.hip).  Producer (lnv 1): lean epilogue + row partials.  Consumer (lnv 2): lean or plain epilogue
// with the rstd / mean correction.  K % 64 == 0 linears (MODE 3) only; NTW 4 | 5; MT 1 (two workgroups per CU) .. 4.
inline bool ln_lean_producer_of(const GemmArgs &g, int mode, int splits)
{
    return g.out_row_stats && !g.row_stats && !g.out_group_stats && !g.chan_parts && splits == 1 && mode == 0 && g.K % 64 == 0 && !g.geglu &&
           g.act == 0 && !g.out_f32 && !g.out_t && g.out && g.N >= 4 && (!g.rowvec || g.rows_per_batch >= 256);
}
inline bool ln_lean_consumer_of(const GemmArgs &g, int mode, int splits)
{
    return g.row_stats && !g.out_row_stats && !g.out_group_stats && !g.chan_parts && splits == 1 && mode == 0 && g.K % 64 == 0 && g.N >= 4;
}
template <class T, int NTW, int MT>
void dispatch8ln_m(const GemmArgs &g, int lnv, bool lean, dim3 grid, hipStream_t s)
{
    if (lnv == 3) { if constexpr (NTW == 5) launch8<T, 3, 5, MT, false, false, false, 3>(g, grid, s); return; }
    if (lnv == 1) launch8<T, 3, NTW, MT, false, false, true, 1>(g, grid, s);
    else if (lean) launch8<T, 3, NTW, MT, false, false, true, 2>(g, grid, s);
    else launch8<T, 3, NTW, MT, false, false, false, 2>(g, grid, s);
}
template <class T>
void dispatch8ln(const GemmArgs &g, int lnv, bool lean, int ntw, int mt, dim3 grid, hipStream_t s)
{
    if (lnv == 2 && g.persist > 0 && ntw == 4 && mt == 4) { launch8p<T, 4, 4, 2>(g, g.persist, s); return; }
    if (mt == 1 && ntw == 4) { dispatch8ln_m<T, 4, 1>(g, lnv, lean, grid, s); return; }
    if (mt < 2) mt = 2;
    if (ntw == 5) {
        if (mt == 4) dispatch8ln_m<T, 5, 4>(g, lnv, lean, grid, s); else if (mt == 3) dispatch8ln_m<T, 5, 3>(g, lnv, lean, grid, s); else dispatch8ln_m<T, 5, 2>(g, lnv, lean, grid, s);
    } else {
        if (mt == 4) dispatch8ln_m<T, 4, 4>(g, lnv, lean, grid, s); else if (mt == 3) dispatch8ln_m<T, 4, 3>(g, lnv, lean, grid, s); else dispatch8ln_m<T, 4, 2>(g, lnv, lean, grid, s);
    }
}

// lean epilogue (no GEGLU / activation / fp32 / transposed output): fast conv, upsample-fused conv, K % 64 == 0 linear
inline bool lean_of(const GemmArgs &g, int mode)
{
    return !g.geglu && g.act == 0 && !g.out_f32 && !g.out_t && g.out && mode != 1 && (mode != 0 || g.K % 64 == 0) && g.N >= 4 &&
           (!g.rowvec || g.rows_per_batch >= 256);
}
template <class T, int NTW, int MT>
void dispatch8lean_m(const GemmArgs &g, int mode, dim3 grid, hipStream_t s)
{
    if (mode == 0) launch8<T, 3, NTW, MT, false, false, true>(g, grid, s);
    else if (g.ups) launch8<T, 4, NTW, MT, false, false, true>(g, grid, s);
    else launch8<T, 2, NTW, MT, false, false, true>(g, grid, s);
}
template <class T>
void dispatch8lean(const GemmArgs &g, int mode, int ntw, int mt, dim3 grid, hipStream_t s)
{
    if (mt == 1 && ntw == 4 && mode == 0) { launch8<T, 3, 4, 1, false, false, true>(g, grid, s); return; }
    if (mt < 2) mt = 2;
    if (ntw == 5) {
        if (mt == 4) dispatch8lean_m<T, 5, 4>(g, mode, grid, s); else if (mt == 3) dispatch8lean_m<T, 5, 3>(g, mode, grid, s); else dispatch8lean_m<T, 5, 2>(g, mode, grid, s);
    } else {
        if (mt == 4) dispatch8lean_m<T, 4, 4>(g, mode, grid, s); else if (mt == 3) dispatch8lean_m<T, 4, 3>(g, mode, grid, s); else dispatch8lean_m<T, 4, 2>(g, mode, grid, s);
    }
}

// channel-partial epilogue (CS): the modes whose output feeds a GroupNorm -- generic conv (conv_in), fast conv, K % 64 == 0 linear (proj_out)
template <class T, int NTW, int MT>
void dispatch8cs_m(const GemmArgs &g, int mode, dim3 grid, hipStream_t s)
{
    if (mode == 0) launch8<T, 3, NTW, MT, false, true>(g, grid, s);
    else if (mode == 1) launch8<T, 1, NTW, MT, false, true>(g, grid, s);
    else launch8<T, 2, NTW, MT, false, true>(g, grid, s);
}
template <class T>
void dispatch8cs(const GemmArgs &g, int mode, int ntw, int mt, dim3 grid, hipStream_t s)
{
    if (mt < 2) mt = 2;
    if (ntw == 5) {
        if (mt == 4) dispatch8cs_m<T, 5, 4>(g, mode, grid, s); else if (mt == 3) dispatch8cs_m<T, 5, 3>(g, mode, grid, s); else dispatch8cs_m<T, 5, 2>(g, mode, grid, s);
    } else {
        if (mt == 4) dispatch8cs_m<T, 4, 4>(g, mode, grid, s); else if (mt == 3) dispatch8cs_m<T, 4, 3>(g, mode, grid, s); else dispatch8cs_m<T, 4, 2>(g, mode, grid, s);
    }
}

template <class T, int NTW, int MT, bool FUSE>
void dispatch8m(const GemmArgs &g, int mode, dim3 grid, hipStream_t s)
{
    if constexpr (FUSE) {      // the modes that carry fused normalisation: generic conv (conv_in), fast conv, K % 64 == 0 linear
        if (mode == 0) launch8<T, 3, NTW, MT, true>(g, grid, s);
        else if (mode == 1) launch8<T, 1, NTW, MT, true>(g, grid, s);
        else launch8<T, 2, NTW, MT, true>(g, grid, s);
    } else {
        if (mode == 0) { if (g.K % 64 == 0) launch8<T, 3, NTW, MT, false>(g, grid, s); else launch8<T, 0, NTW, MT, false>(g, grid, s); }
        else if (mode == 1) launch8<T, 1, NTW, MT, false>(g, grid, s);
        else if (g.ups) launch8<T, 4, NTW, MT, false>(g, grid, s); else launch8<T, 2, NTW, MT, false>(g, grid, s);
    }
}
template <class T, bool FUSE>
void dispatch8(const GemmArgs &g, int mode, int ntw, int mt, dim3 grid, hipStream_t s)
{
    if constexpr (!FUSE) {
        // MT = 1 (64 x 128 tile, 72 KiB LDS, <= 128 VGPRs): TWO workgroups per CU -- for the K = N = C linears of the 32x32 / 16x16 levels,
        // whose few k-tiles leave a one-workgroup-per-CU kernel with its fill and epilogue latency fully exposed
        if (mt == 1 && ntw == 4 && mode == 0 && g.K % 64 == 0) { launch8<T, 3, 4, 1, false>(g, grid, s); return; }
        if (g.persist > 0 && ntw == 4 && mt == 4 && mode == 0) { launch8p<T, 4, 4>(g, g.persist, s); return; }
    }
    if (mt < 2) mt = 2;
    if (ntw == 5) {
        if (mt == 4) dispatch8m<T, 5, 4, FUSE>(g, mode, grid, s); else if (mt == 3) dispatch8m<T, 5, 3, FUSE>(g, mode, grid, s); else dispatch8m<T, 5, 2, FUSE>(g, mode, grid, s);
    } else {
        if (mt == 4) dispatch8m<T, 4, 4, FUSE>(g, mode, grid, s); else if (mt == 3) dispatch8m<T, 4, 3, FUSE>(g, mode, grid, s); else dispatch8m<T, 4, 2, FUSE>(g, mode, grid, s);
    }
}

template <class T, bool FUSE>
void dispatch4(const GemmArgs &g, int mode, int ntw, dim3 grid, hipStream_t s)
{
    if (ntw == 5) {
        if (mode == 0) launch4<T, 0, 5, FUSE>(g, grid, s); else if (mode == 1) launch4<T, 1, 5, FUSE>(g, grid, s); else launch4<T, 2, 5, FUSE>(g, grid, s);
    } else {
        if (mode == 0) launch4<T, 0, 4, FUSE>(g, grid, s); else if (mode == 1) launch4<T, 1, 4, FUSE>(g, grid, s); else launch4<T, 2, 4, FUSE>(g, grid, s);
    }
}

}  // namespace

// entry points of the three kernel translation units (dtype: DT_BF16 / DT_F16; mode as resolved by gc_dn_gemm: 0 linear, 1 generic conv,
// 2 fast conv; mt8 = 0: 4-wave kernel)
using dng::GemmArgs;
void dn_gemm_launch_plain(const GemmArgs &g, int dtype, int mode, int ntw, int mt8, dim3 grid, hipStream_t s);
void dn_gemm_launch_fuse(const GemmArgs &g, int dtype, int mode, int ntw, int mt8, dim3 grid, hipStream_t s);
void dn_gemm_launch_fp8(const GemmArgs &g, int dtype, int mode, int ntw, int mt, dim3 grid, hipStream_t s);
void dn_gemm_launch_splitk_epilogue(const GemmArgs &g, int dtype, hipStream_t s);
void dn_gemm_launch_cs(const GemmArgs &g, int dtype, int mode, int ntw, int mt8, dim3 grid, hipStream_t s);     // 8-wave kernel + channel partials
void dn_gemm_launch_splitk_epilogue_cs(const GemmArgs &g, int dtype, hipStream_t s);
void dn_gemm_launch_lean(const GemmArgs &g, int dtype, int mode, int ntw, int mt8, dim3 grid, hipStream_t s);   // 8-wave kernel, lean epilogue
void dn_gemm_launch_ln(const GemmArgs &g, int dtype, int lnv, bool lean, int ntw, int mt8, dim3 grid, hipStream_t s);   // lean LayerNorm fold (producer 1 / consumer 2)
