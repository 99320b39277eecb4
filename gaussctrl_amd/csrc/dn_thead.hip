// dn_thead.hip -- the "head" of a level-0 SD1.5 transformer block (C = 320) as ONE kernel: everything between the GroupNorm statistics
// and the cross-view self-attention,
//     xn = GroupNorm(x) (coefficients a, d per (image, channel) from gc_dn_groupnorm_coef)       h = proj_in(xn)
//     n1 = LayerNorm1(h)          Q | K = n1 Wq^T | n1 Wk^T  ->  qk [M][640]          V^T = Wv n1^T  ->  vt [B][320][L]
// (Transformer2DModel.forward: norm, proj_in; BasicTransformerBlock.forward: norm1, attn1.to_q / to_k / to_v -- the operands of the
// attention processor the reference installs in gaussctrl/gc_pipeline.py:224-227, utils.py:60-117).  Four launches of the per-op path.
//
// Same machine as dn_ttail.hip (read its header first): a wave owns 32 token rows in registers, every GEMM runs transposed on
// v_mfma_f32_32x32x16 with k-permuted weight tiles from a linear operand stream, 4 waves share the stream through a 15-slot LDS ring.
// New here: V^T falls out of the MFMA itself -- with the operands SWAPPED (A = the rows' fragment, B = the weight tile; both layouts are
// the same 8 k-values per lane) the accumulator holds D[token][channel], i.e. a lane owns one channel and 4 consecutive tokens per
// register quad: 8-byte stores into the token-contiguous V^T rows the attention kernel reads.
#include "dn_attn_common.h"

namespace {

constexpr int TC = 320, NB = TC / 32, KS = TC / 16;
constexpr int SLOT_BLK = 8, SLOT = SLOT_BLK * 1024, NSLOT = 15, RING_BLK = NSLOT * SLOT_BLK;
static_assert(NSLOT == 15 && KS * NB == 200, "the store windows in fetch() are written for 13 slots in flight and 200-block GEMMs");
constexpr int G_IN = 0, G_Q = KS * NB, G_K = 2 * KS * NB, G_V = 3 * KS * NB, BLK_TOTAL = 4 * KS * NB, NSLOTS_TOTAL = BLK_TOTAL / SLOT_BLK;
// parameter table (floats): bias of proj_in, LayerNorm1 gamma / beta in lane order (dn_ttail.hip), then this image's GroupNorm coefficients
// (a, d) per channel in natural order
constexpr int P_BIN = 0, P_G1 = 320, P_B1 = 640, P_TABLE = 960, P_COEF = 960, P_TOTAL = P_COEF + 2 * TC;
constexpr int LDS_BYTES = NSLOT * SLOT + P_TOTAL * 4;

struct HeadArgs {
    const unsigned short *x;               // [M][320] the block's input (before GroupNorm)
    const float *coef;                     // [B][320][2] GroupNorm coefficients: xn = x * a + d
    unsigned short *h, *qk, *vt;           // [M][320], [M][640] = Q | K, [B][320][ldvt]
    const unsigned char *w;                // operand stream: proj_in, to_q, to_k, to_v (BLK_TOTAL KB)
    const float *params;                   // [P_TABLE]
    int M, rows_per_frame, h_frags;
    int64_t ldvt, vt_bs;
    float eps;
};

#ifdef TTAIL_ABLATIONS
__device__ unsigned long long g_hstamps[16];
#define STAMP(i) do { if (blockIdx.x == 37 && tid == 0) g_hstamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

template <class T>
__global__ __launch_bounds__(256, 1) void k_thead(const HeadArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *prm = reinterpret_cast<float *>(smem + NSLOT * SLOT);
    const int64_t row0 = (int64_t)blockIdx.x * 128;
    const int img = (int)(row0 / a.rows_per_frame);

    // ---------------- the stream (dn_ttail.hip): one segment here
    int issue_slot = 0;
    const unsigned char *isrc = a.w + (size_t)wid * 2048;
    const unsigned voff = lane * 16;
    auto issue = [&](int ring_slot) __attribute__((always_inline)) {
        const unsigned dst = lds0 + ring_slot * SLOT + 2 * wid * 1024;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(isrc), "s"(dst) : "memory");
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(isrc + 1024), "s"(dst + 1024) : "memory");
        ++issue_slot;
        isrc += SLOT;
    };
    uint4 pre[SLOT_BLK];
    const unsigned char *my = smem + lane * 16, *my_hi = my + 65536;       // two bases: every block offset fits the 16-bit immediate
    auto fetch = [&](int g) __attribute__((always_inline)) {
        if ((g & (SLOT_BLK - 1)) == 0) {
            if (issue_slot < NSLOTS_TOTAL) {
                // vmcnt retires in order and counts stores too: the 20 row stores after a GEMM are YOUNGER than the DMA of the next 13 slots
                // to be entered, so those waits may leave them outstanding (else 20 stores + 4 DMAs fill the allowance: the stream starves)
                const int s = g / SLOT_BLK;
                const bool after_store = (s >= 26 && s <= 38) || (s >= 51 && s <= 63) || (s >= 76 && s <= 88);
                if (after_store) wait_vmcnt<2 * (NSLOT - 3) + KS>(); else wait_vmcnt<2 * (NSLOT - 3)>();
                __builtin_amdgcn_s_barrier();
                issue((g / SLOT_BLK + NSLOT - 2) % NSLOT);
            } else {
                wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
            }
        }
        const int o = (g % RING_BLK) * 1024;
        pre[g & (SLOT_BLK - 1)] = *reinterpret_cast<const uint4 *>(o < 65536 ? my + o : my_hi + (o - 65536));
    };
    auto blk = [&](int base, int i) __attribute__((always_inline)) -> uint4 {
        const uint4 v = pre[i & (SLOT_BLK - 1)];
        if (base + i + SLOT_BLK < BLK_TOTAL) fetch(base + i + SLOT_BLK);
        __builtin_amdgcn_sched_barrier(0);
        return v;
    };
#pragma unroll 1
    for (int i = 0; i < NSLOT - 2; ++i) issue(i);
#pragma unroll
    for (int i = 0; i < SLOT_BLK; ++i) fetch(i);

    for (int i = tid; i < P_TABLE / 4; i += 256) reinterpret_cast<float4 *>(prm)[i] = reinterpret_cast<const float4 *>(a.params)[i];
    for (int i = tid; i < 2 * TC / 4; i += 256)
        reinterpret_cast<float4 *>(prm + P_COEF)[i] = reinterpret_cast<const float4 *>(a.coef + (size_t)img * 2 * TC)[i];

    auto fresh_lane = [&]() __attribute__((always_inline)) -> unsigned {
        unsigned l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    };
    auto lo_f = [](unsigned w) __attribute__((always_inline)) { return T::to_f((unsigned short)(w & 0xffff)); };
    auto hi_f = [](unsigned w) __attribute__((always_inline)) { return T::to_f((unsigned short)(w >> 16)); };
    // rows in lane order (dn_ttail.hip): word w of k-step ks = channels 16 ks + {4 hg + 2 w', 8 + 4 hg + 2 w'}
    // one 16-byte store per k-step: the lane pair (m, 0) / (m, 1) first trades halves, so that hg = 0 owns channels 16 ks .. + 7 and hg = 1
    // channels 16 ks + 8 .. + 15 (stores are issue-bound here: 20 dwordx4 cost half of 40 dwordx2)
    auto store_rows = [&](const uint4 *src, unsigned short *p, int ld) __attribute__((always_inline)) {
        const unsigned l = fresh_lane();
        const bool up = (l >> 5) != 0;
        unsigned short *r = p + (row0 + wid * 32 + (l & 31)) * ld + 8 * (l >> 5);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const unsigned s0 = up ? src[ks].x : src[ks].z, s1 = up ? src[ks].y : src[ks].w;      // what the partner is missing
            const unsigned r0 = __shfl_xor(s0, 32, 64), r1 = __shfl_xor(s1, 32, 64);
            *reinterpret_cast<uint4 *>(r + 16 * ks) = up ? make_uint4(r0, r1, src[ks].z, src[ks].w) : make_uint4(src[ks].x, src[ks].y, r0, r1);
        }
    };
    // the residual stream h goes to the tail kernel only: stored as the fragments themselves ([32-row block][k-step][lane][16 bytes]),
    // 1 KB contiguous per store instruction
    auto store_frags = [&](const uint4 *src, unsigned short *p) __attribute__((always_inline)) {
        uint4 *r = reinterpret_cast<uint4 *>(p) + ((row0 >> 5) + wid) * (KS * 64) + fresh_lane();
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) r[ks * 64] = src[ks];
    };
    uint4 xf[KS], hf[KS];
    f32x16 acc[NB];
    {
        const unsigned l = fresh_lane();
        const unsigned short *r = a.x + (row0 + wid * 32 + (l & 31)) * TC + 4 * (l >> 5);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const uint2 lo = *reinterpret_cast<const uint2 *>(r + 16 * ks), hi = *reinterpret_cast<const uint2 *>(r + 16 * ks + 8);
            xf[ks] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
    }
    STAMP(0);
    __syncthreads();                       // tables visible
    STAMP(1);

    // ---------------- GroupNorm: xn = round(x * a + d), coefficient pairs of 4 consecutive channels = 32 contiguous bytes
    {
        const float *cf = prm + P_COEF + 8 * (fresh_lane() >> 5);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            float c0[8], c1[8];
            *reinterpret_cast<float4 *>(c0) = *reinterpret_cast<const float4 *>(cf + 32 * ks);
            *reinterpret_cast<float4 *>(c0 + 4) = *reinterpret_cast<const float4 *>(cf + 32 * ks + 4);
            *reinterpret_cast<float4 *>(c1) = *reinterpret_cast<const float4 *>(cf + 32 * ks + 16);
            *reinterpret_cast<float4 *>(c1 + 4) = *reinterpret_cast<const float4 *>(cf + 32 * ks + 20);
            const uint4 v = xf[ks];
            xf[ks] = make_uint4(pack2<T>(lo_f(v.x) * c0[0] + c0[1], hi_f(v.x) * c0[2] + c0[3]), pack2<T>(lo_f(v.y) * c0[4] + c0[5], hi_f(v.y) * c0[6] + c0[7]),
                                pack2<T>(lo_f(v.z) * c1[0] + c1[1], hi_f(v.z) * c1[2] + c1[3]), pack2<T>(lo_f(v.w) * c1[4] + c1[5], hi_f(v.w) * c1[6] + c1[7]));
            if (ks & 1) __builtin_amdgcn_sched_barrier(0);
        }
    }

    auto mma = [&](uint4 wv, uint4 xv, f32x16 c) __attribute__((always_inline)) -> f32x16 { return T::mfma32(wv, xv, c); };
    // acc[nb] += W[32 nb ..][k] x[k] (blocks in (ks, nb) order); epi(nb) two MFMAs after block nb is complete (dn_ttail.hip)
    // (static_for, not #pragma unroll: hipcc refuses to unroll 200 iterations of this size "as directed", and one dynamic index sends
    // the register arrays to scratch memory)
    auto gemm = [&](auto base_c, const uint4 *x, auto &&epi) __attribute__((always_inline)) {
        constexpr int base = decltype(base_c)::value;
        static_for<0, KS>([&](auto ksc) {
            constexpr int ks = decltype(ksc)::value;
            static_for<0, NB>([&](auto nbc) {
                constexpr int nb = decltype(nbc)::value;
                acc[nb] = mma(blk(base, ks * NB + nb), x[ks], acc[nb]);
                if constexpr (ks == KS - 1 && nb >= 2) epi(nb - 2);
            });
        });
        epi(NB - 2);
        epi(NB - 1);
    };
    auto frag_block = [&](uint4 *dst, int nb) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
            dst[2 * nb + j] = make_uint4(pack2<T>(acc[nb][8 * j], acc[nb][8 * j + 1]), pack2<T>(acc[nb][8 * j + 2], acc[nb][8 * j + 3]),
                                         pack2<T>(acc[nb][8 * j + 4], acc[nb][8 * j + 5]), pack2<T>(acc[nb][8 * j + 6], acc[nb][8 * j + 7]));
    };
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    };

    STAMP(2);
    // ---------------- h = proj_in(xn): the bias is the accumulators' initial value
    {
        const float *prm_l = prm + P_BIN + 16 * (fresh_lane() >> 5);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            float bv[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<float4 *>(bv + 4 * q) = *reinterpret_cast<const float4 *>(prm_l + 32 * nb + 4 * q);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = bv[r];
            asm volatile("" : "+a"(acc[nb]));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    gemm(std::integral_constant<int, G_IN>{}, xf, [&](int nb) { frag_block(hf, nb); });
    STAMP(3);
    if (a.h_frags) store_frags(hf, a.h); else store_rows(hf, a.h, TC);
    STAMP(4);

    // ---------------- n1 = LayerNorm1(h) (a row's 320 channels live in the lane pair (m, 0) / (m, 1))
    {
        const float *prm_l = prm + 16 * (fresh_lane() >> 5);
        float s = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const unsigned w[4] = {hf[ks].x, hf[ks].y, hf[ks].z, hf[ks].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) s += lo_f(w[i]) + hi_f(w[i]);
        }
        s += __shfl_xor(s, 32, 64);
        const float mean = s * (1.f / TC);
        float v = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            unsigned w[4] = {hf[ks].x, hf[ks].y, hf[ks].z, hf[ks].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(w[i]));        // unpack again (dn_ttail.hip)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float d0 = lo_f(w[i]) - mean, d1 = hi_f(w[i]) - mean;
                v += d0 * d0 + d1 * d1;
            }
        }
        v += __shfl_xor(v, 32, 64);
        const float rstd = __builtin_amdgcn_rsqf(v * (1.f / TC) + a.eps);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            float g[8], b[8];
            const int o = 32 * (ks >> 1) + 8 * (ks & 1);
            *reinterpret_cast<float4 *>(g) = *reinterpret_cast<const float4 *>(prm_l + (P_G1 + o));
            *reinterpret_cast<float4 *>(g + 4) = *reinterpret_cast<const float4 *>(prm_l + (P_G1 + o + 4));
            *reinterpret_cast<float4 *>(b) = *reinterpret_cast<const float4 *>(prm_l + (P_B1 + o));
            *reinterpret_cast<float4 *>(b + 4) = *reinterpret_cast<const float4 *>(prm_l + (P_B1 + o + 4));
            unsigned w[4] = {hf[ks].x, hf[ks].y, hf[ks].z, hf[ks].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(w[i]));
            unsigned ow[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                ow[i] = pack2<T>((lo_f(w[i]) - mean) * rstd * g[2 * i] + b[2 * i], (hi_f(w[i]) - mean) * rstd * g[2 * i + 1] + b[2 * i + 1]);
            xf[ks] = make_uint4(ow[0], ow[1], ow[2], ow[3]);
            if (ks & 1) __builtin_amdgcn_sched_barrier(0);
        }
    }

    STAMP(5);
    // ---------------- Q, K -> qk [M][640]
    zero_acc();
    gemm(std::integral_constant<int, G_Q>{}, xf, [&](int nb) { frag_block(hf, nb); });
    STAMP(6);
    store_rows(hf, a.qk, 2 * TC);
    STAMP(7);
    zero_acc();
    gemm(std::integral_constant<int, G_K>{}, xf, [&](int nb) { frag_block(hf, nb); });
    STAMP(8);
    store_rows(hf, a.qk + TC, 2 * TC);
    STAMP(9);

    // ---------------- V^T: operands swapped, D[token][channel]: lane (channel 32 nb + (lane & 31), hg), register 4 g + c = token 8 g + 4 hg + c
    zero_acc();
    auto vt_store = [&](int nb) __attribute__((always_inline)) {
        const unsigned l = fresh_lane();
        const int64_t tok0 = row0 + wid * 32 - (int64_t)img * a.rows_per_frame;
        unsigned short *r = a.vt + (int64_t)img * a.vt_bs + (int64_t)(32 * nb + (l & 31)) * a.ldvt + tok0 + 4 * (l >> 5);
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<uint2 *>(r + 8 * g) = make_uint2(pack2<T>(acc[nb][4 * g], acc[nb][4 * g + 1]), pack2<T>(acc[nb][4 * g + 2], acc[nb][4 * g + 3]));
    };
    static_for<0, KS>([&](auto ksc) {
        constexpr int ks = decltype(ksc)::value;
        static_for<0, NB>([&](auto nbc) {
            constexpr int nb = decltype(nbc)::value;
            acc[nb] = mma(xf[ks], blk(G_V, ks * NB + nb), acc[nb]);
            if constexpr (ks == KS - 1 && nb >= 2) vt_store(nb - 2);
        });
    });
    STAMP(10);
    vt_store(NB - 2);
    vt_store(NB - 1);
    STAMP(11);
}

template <class T> int launch(const HeadArgs &a, hipStream_t s)
{
    static gc::AttrOnce once;
    gc::ensure_dynamic_lds(once, (const void *)k_thead<T>, LDS_BYTES);
    hipLaunchKernelGGL((k_thead<T>), dim3((unsigned)(a.M / 128)), dim3(256), LDS_BYTES, s, a);
    return gc::check_launch("gc_dn_transformer_head");
}

}  // namespace

#ifdef TTAIL_ABLATIONS
extern "C" void gc_dn_transformer_head_stamps(unsigned long long *host16)
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_hstamps), sizeof(unsigned long long) * 16);
}
#endif

extern "C" int gc_dn_transformer_head(const gc_thead_desc *d, void *stream)
{
    GC_REQUIRE(d && d->x && d->gn_coef && d->h && d->qk && d->vt && d->w && d->params, "NULL argument");
    GC_REQUIRE(d->channels == TC, "the fused head is built for C = 320 (SD1.5 level 0)");
    GC_REQUIRE(d->M > 0 && d->rows_per_frame > 0 && d->rows_per_frame % 128 == 0 && d->M % d->rows_per_frame == 0, "rows_per_frame must be a multiple of 128 dividing M");
    GC_REQUIRE(d->ldvt >= d->rows_per_frame && d->ldvt % 4 == 0 && d->vt_batch_stride >= (int64_t)TC * d->ldvt, "V^T rows: ldvt >= rows_per_frame, 8-byte aligned");
    HeadArgs a;
    a.x = (const unsigned short *)d->x; a.coef = d->gn_coef; a.h = (unsigned short *)d->h; a.qk = (unsigned short *)d->qk; a.vt = (unsigned short *)d->vt;
    a.w = (const unsigned char *)d->w; a.params = d->params; a.M = (int)d->M; a.rows_per_frame = (int)d->rows_per_frame;
    a.ldvt = d->ldvt; a.vt_bs = d->vt_batch_stride; a.eps = d->ln_eps; a.h_frags = d->h_fragment_layout;
    if (d->dtype == DT_BF16) return launch<BF16>(a, (hipStream_t)stream);
    if (d->dtype == DT_F16) return launch<F16>(a, (hipStream_t)stream);
    GC_REQUIRE(false, "dtype");
}
