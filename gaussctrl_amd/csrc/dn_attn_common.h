// dn_attn_common.h -- pieces shared by the attention translation units (dn_attn.hip: k_attn / k_attn3 / k_attn4 and the C ABI entry;
// dn_attn5.hip: k_attn5): argument block, XCD-aware workgroup order, the online-softmax body (whole kernel for plain head sizes,
// in-kernel fallback of the static-offset kernels), LDS-DMA helpers.  Everything has internal linkage (anonymous namespace per TU).
#pragma once
#include "dn_common.h"
#include <cstdlib>
#include <type_traits>

namespace {
using namespace dn;

struct AttnArgs {
    const unsigned short *Q; int64_t ldq, q_bs;      // [B][Lq][ldq]
    const unsigned short *K; int64_t ldk, k_bs;      // [Bk][Lk][ldk]
    const unsigned short *Vt; int64_t ldvt, vt_bs;   // [Bk][H*D][ldvt]  (token-contiguous)
    const unsigned short *Kr; int64_t kr_bs;         // reference-frame bank (may alias K): [halves*ref_fph][Lk][ldk]
    const unsigned short *Vtr; int64_t vtr_bs;
    int ref_fph;                                     // frames per CFG half inside the reference bank
    unsigned short *O; int64_t ldo, o_bs;            // [B][Lq][ldo]
    int Lq, Lk, H, f;
    int nsets; int set_kind[5]; float set_w[5];      // kind -1: own frame; -2: frame b / f (shared text K/V); r >= 0: reference r of the half
    float scale_log2e;
    int nqb;                                         // query blocks per (batch, head)
    int abl;                                         // timing ablations of k_attn5's instrumented instantiation (kernel_variant >> 8; 0 in production)
    float *part;                                     // set-split launches (k_attn, gridDim.y = nsets): fp32 [nsets][B][Lq][H*D] weighted per-set outputs
};

// 1-D grid, XCD-aware: consecutive workgroup ids go round-robin over the 8 XCDs (each with its own L2), so the id is remapped
// bijectively to make the query blocks of one (batch, head) -- which stream the same K / V^T -- land on ONE XCD's L2.
__device__ __forceinline__ void block_coords(const AttnArgs &a, int QT, int &qblk, int &h, int &b)
{
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, qq = nwg >> 3, rr = nwg & 7;
    const int bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (blockIdx.x >> 3);
    qblk = bid % a.nqb;
    const int bh = bid / a.nqb;
    h = bh % a.H; b = bh / a.H;
}


template <class T> struct One;
template <> struct One<BF16> { static constexpr unsigned short v = 0x3F80; };
template <> struct One<F16> { static constexpr unsigned short v = 0x3C00; };

template <int D> struct SafeLds {
    static constexpr int DP = (D + 31) / 32 * 32, DV = (D + 15) / 16 * 16;
    static constexpr int KROW = DP == 64 ? 128 : (DP + 8) * 2, VROW = (64 + 8) * 2;
    static constexpr int KBYTES = 64 * KROW, VBYTES = DV * VROW;
};

// The "safe" form: online softmax with a running row maximum (no assumption on the logits).  It is the whole kernel for the
// head sizes without spare contraction columns and the in-kernel fallback of k_attn3.
// SPLIT_ONLY: the kernel is only launched in the set-split form (gridDim.y = nsets): the set loop is ONE iteration at compile time and the weighted
// total does not live beside the set's accumulator (k_attn_wide: 32 queries per wave at head size 160 would not fit otherwise).
template <class T, int D, int QT, int NW = 4, bool SPLIT_ONLY = false>
__device__ __forceinline__ void attn_safe_body(const AttnArgs &a, int qblk, int h, int b, unsigned char *sK, unsigned char *sV)
{
    constexpr int NT = NW * 64;                 // threads of the workgroup
    constexpr int DP = (D + 31) / 32 * 32;      // contraction length of QK^T, padded to the MFMA k = 32
    constexpr int DV = (D + 15) / 16 * 16;      // output rows of O^T, padded to the MFMA m = 16
    constexpr int KS = DP / 32, DT = DV / 16;
    constexpr bool SWZ = (DP == 64);            // 128-byte key rows: XOR-swizzled 16-byte chunks (conflict-free ds_read_b128)
    constexpr int KROW = SWZ ? 128 : (DP + 8) * 2;   // bytes per key row of the K tile
    constexpr int VROW = (64 + 8) * 2;          // bytes per channel row of the V^T tile
    // When D is not a multiple of 16 the padded V^T row D is filled with ones: O^T row D then accumulates the softmax
    // denominator sum_k P[q][k] inside the P V MFMAs (same rounding of P as the numerator) -- no VALU adds for l.
    constexpr bool ONES = (DV > D);
    constexpr int NKC = 64 * (D / 8), NVC = D * 8;                     // 16-byte chunks per K / V^T tile
    constexpr int KIT = (NKC + NT - 1) / NT, VIT = (NVC + NT - 1) / NT;
    static_assert(KROW == SafeLds<D>::KROW && VROW == SafeLds<D>::VROW && DV == SafeLds<D>::DV, "LDS plan");
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fr = lane & 15, g = lane >> 4;
    const int q_wave0 = (qblk * NW + wid) * (QT * 16);

    // zero the LDS once: pad columns / pad rows are never written again
    for (int i = tid; i < 64 * KROW / 16; i += NT) reinterpret_cast<uint4 *>(sK)[i] = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < DV * VROW / 16; i += NT) reinterpret_cast<uint4 *>(sV)[i] = make_uint4(0, 0, 0, 0);
    if (ONES) {
        __syncthreads();
        if (tid < 64) reinterpret_cast<unsigned short *>(sV + D * VROW)[tid] = One<T>::v;
    }

    // Q fragments (B operand): lane holds Q[q = fr][d = 32*ks + 8*g .. +8]
    uint4 qf[QT][KS];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int q = q_wave0 + qt * 16 + fr;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int d = ks * 32 + g * 8;
            qf[qt][ks] = (q < a.Lq && d + 8 <= D)
                             ? *reinterpret_cast<const uint4 *>(a.Q + (int64_t)b * a.q_bs + (int64_t)q * a.ldq + h * D + d)
                             : make_uint4(0, 0, 0, 0);
        }
    }

    // per-lane staging coordinates (constant over tiles): element offsets inside a tile, LDS byte offsets
    int k_go[KIT], k_lo[KIT], k_r[KIT], v_go[VIT], v_lo[VIT], v_c[VIT];
#pragma unroll
    for (int j = 0; j < KIT; ++j) {
        const int c = tid + NT * j, r = c / (D / 8), cc = c - r * (D / 8);
        k_r[j] = c < NKC ? r : -1;
        k_go[j] = c < NKC ? r * (int)a.ldk + cc * 8 : 0;
        k_lo[j] = SWZ ? r * 128 + ((cc ^ ((r >> 1) & 7)) << 4) : r * KROW + cc * 16;
    }
#pragma unroll
    for (int j = 0; j < VIT; ++j) {
        const int c = tid + NT * j, r = c >> 3, cc = c & 7;
        v_c[j] = c < NVC ? cc * 8 : -1;
        v_go[j] = c < NVC ? r * (int)a.ldvt + cc * 8 : 0;
        v_lo[j] = r * VROW + cc * 16;
    }
    // K fragment LDS offsets: row = 16 kt + fr, chunk = 4 ks + g
    int kfo[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) kfo[ks] = SWZ ? fr * 128 + (((ks * 4 + g) ^ ((fr >> 1) & 7)) << 4) : fr * KROW + (ks * 32 + g * 8) * 2;

    f32x4 otot[DT][QT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) otot[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ntiles = (a.Lk + 63) / 64;
    const int lk8 = (a.Lk + 7) / 8 * 8;
    const float c2 = a.scale_log2e;
    // set-split launch: this workgroup handles K/V set blockIdx.y only and leaves its weighted output in a.part (the sets are independent
    // attentions, out = sum_s w_s O_s: k_attn_combine adds them in a fixed order)
    const bool split = SPLIT_ONLY || gridDim.y > 1;
    const int s_begin = split ? (int)blockIdx.y : 0, s_end = split ? s_begin + 1 : a.nsets;
    for (int s = s_begin; SPLIT_ONLY ? s == s_begin : s < s_end; ++s) {
        const int kind = a.set_kind[s];
        const unsigned short *Kb, *Vb;
        if (kind >= 0) {
            const int kvb = (b / a.f) * a.ref_fph + kind;
            Kb = a.Kr + (int64_t)kvb * a.kr_bs + h * D;
            Vb = a.Vtr + (int64_t)kvb * a.vtr_bs + (int64_t)h * D * a.ldvt;
        } else {
            const int kvb = kind == -1 ? b : b / a.f;
            Kb = a.K + (int64_t)kvb * a.k_bs + h * D;
            Vb = a.Vt + (int64_t)kvb * a.vt_bs + (int64_t)h * D * a.ldvt;
        }
        f32x4 os[DT][QT];
        float mrow[QT], lrow[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            mrow[qt] = -1e30f; lrow[qt] = 0.f;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) os[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        for (int tile = 0; tile < ntiles; ++tile) {
            const int key0 = tile * 64;
            const bool full = key0 + 64 <= a.Lk;           // wave-uniform: interior tiles skip every bounds test
            __syncthreads();   // previous tile fully consumed (also orders the initial LDS fill)
            {
                const unsigned short *kp = Kb + (int64_t)key0 * a.ldk;
                const unsigned short *vp = Vb + key0;
                uint4 kv[KIT], vv[VIT];
                if (full) {   // interior tile: no bounds tests (lanes without a chunk re-read offset 0 and do not store)
#pragma unroll
                    for (int j = 0; j < KIT; ++j) kv[j] = *reinterpret_cast<const uint4 *>(kp + k_go[j]);
#pragma unroll
                    for (int j = 0; j < VIT; ++j) vv[j] = *reinterpret_cast<const uint4 *>(vp + v_go[j]);
                } else {
                    asm volatile("" ::: "memory");   // keep the wave-uniform branch
#pragma unroll
                    for (int j = 0; j < KIT; ++j) {
                        const bool ok = k_r[j] >= 0 && key0 + k_r[j] < a.Lk;
                        kv[j] = *reinterpret_cast<const uint4 *>(kp + (ok ? k_go[j] : 0));
                        if (!ok) kv[j] = make_uint4(0, 0, 0, 0);
                    }
#pragma unroll
                    for (int j = 0; j < VIT; ++j) {
                        const bool ok = v_c[j] >= 0 && key0 + v_c[j] < lk8;
                        vv[j] = *reinterpret_cast<const uint4 *>(vp + (ok ? v_go[j] : 0));
                        if (!ok) vv[j] = make_uint4(0, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int j = 0; j < KIT; ++j)
                    if (k_r[j] >= 0) *reinterpret_cast<uint4 *>(sK + k_lo[j]) = kv[j];
#pragma unroll
                for (int j = 0; j < VIT; ++j)
                    if (v_c[j] >= 0) *reinterpret_cast<uint4 *>(sV + v_lo[j]) = vv[j];
            }
            __syncthreads();

            // ---- S^T = K Q^T : st[kt][qt][reg] = S[q = fr][key = key0 + 16*kt + 4*g + reg]
            f32x4 st[4][QT];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) st[kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    const uint4 kf = *reinterpret_cast<const uint4 *>(sK + kfo[ks] + kt * 16 * KROW);
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) st[kt][qt] = T::mfma(kf, qf[qt][ks], st[kt][qt]);
                }
            if (!full) {   // last, partial tile: mask keys >= Lk (the asm keeps this a real wave-uniform branch)
                asm volatile("" ::: "memory");
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (key0 + kt * 16 + g * 4 + r >= a.Lk) {
#pragma unroll
                            for (int qt = 0; qt < QT; ++qt) st[kt][qt][r] = -1e30f;
                        }
            }
            // ---- online softmax per query row (lane-local + 2 cross-lane steps over g); exp2(c2*s - m) in one FMA + v_exp
            uint4 pf[QT][2];
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                float tmax = fmaxf(fmaxf(st[0][qt][0], st[0][qt][1]), st[0][qt][2]);      // v_max3_f32 chain
                tmax = fmaxf(fmaxf(tmax, st[0][qt][3]), st[1][qt][0]);
#pragma unroll
                for (int kt = 1; kt < 4; ++kt) {
                    if (kt > 1) tmax = fmaxf(fmaxf(tmax, st[kt - 1][qt][3]), st[kt][qt][0]);
                    tmax = fmaxf(fmaxf(tmax, st[kt][qt][1]), st[kt][qt][2]);
                }
                tmax = fmaxf(tmax, st[3][qt][3]);
                tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                const float mnew = fmaxf(mrow[qt], tmax * c2);
                const float alpha = __builtin_amdgcn_exp2f(mrow[qt] - mnew);
                mrow[qt] = mnew;
                float p[4][4];
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) p[kt][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kt][qt][r], c2, -mnew));
                if (!ONES) {
                    float psum = 0.f;
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt) psum += (p[kt][0] + p[kt][1]) + (p[kt][2] + p[kt][3]);
                    lrow[qt] = lrow[qt] * alpha + psum;   // per-lane partial; reduced over g at the end of the set
                }
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) os[dt][qt][r] *= alpha;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
                    pf[qt][kb] = make_uint4(pack2<T>(p[2 * kb][0], p[2 * kb][1]), pack2<T>(p[2 * kb][2], p[2 * kb][3]),
                                            pack2<T>(p[2 * kb + 1][0], p[2 * kb + 1][1]), pack2<T>(p[2 * kb + 1][2], p[2 * kb + 1][3]));
            }
            // ---- O^T += V^T P^T
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    const unsigned char *vr = sV + (dt * 16 + fr) * VROW;
                    const uint2 lo = *reinterpret_cast<const uint2 *>(vr + ((2 * kb) * 16 + g * 4) * 2);
                    const uint2 hi = *reinterpret_cast<const uint2 *>(vr + ((2 * kb + 1) * 16 + g * 4) * 2);
                    const uint4 vf = make_uint4(lo.x, lo.y, hi.x, hi.y);
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) os[dt][qt] = T::mfma(vf, pf[qt][kb], os[dt][qt]);
                }
        }
        // ---- fold this set into the weighted total
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            float l;
            if (ONES) {   // denominator = O^T row D: held by the lane with 16*dt + 4*g + r == D of the same query column
                constexpr int dt_l = D / 16, g_l = (D % 16) / 4, r_l = D % 4;
                l = __shfl(os[dt_l][qt][r_l], g_l * 16 + fr, 64);
            } else {
                l = lrow[qt];
                l += __shfl_xor(l, 16, 64);
                l += __shfl_xor(l, 32, 64);
            }
            const float inv = a.set_w[s] / l;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                // (explicit multiply THEN add, never an FMA: the set-split launch stores round(os * inv) per set and k_attn_combine adds the
                // sets in this order, so the two forms agree bit for bit whatever grid size picked between them)
                for (int r = 0; r < 4; ++r) otot[dt][qt][r] = __fadd_rn(otot[dt][qt][r], __fmul_rn(os[dt][qt][r], inv));
        }
    }
    // ---- store: lane owns O[q = fr][d = 16*dt + 4*g .. +4]
    if (split) {
        float *pb = a.part + (((int64_t)s_begin * gridDim.x / (a.nqb * a.H) + b) * a.Lq) * (int64_t)(a.H * D);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            const int q = q_wave0 + qt * 16 + fr;
            if (q >= a.Lq) continue;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int d = dt * 16 + g * 4;
                if (d + 4 > D) continue;
                *reinterpret_cast<float4 *>(pb + (int64_t)q * (a.H * D) + h * D + d) =
                    make_float4(otot[dt][qt][0], otot[dt][qt][1], otot[dt][qt][2], otot[dt][qt][3]);
            }
        }
        return;
    }
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int q = q_wave0 + qt * 16 + fr;
        if (q >= a.Lq) continue;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d = dt * 16 + g * 4;
            if (d + 4 > D) continue;
            *reinterpret_cast<uint2 *>(a.O + (int64_t)b * a.o_bs + (int64_t)q * a.ldo + h * D + d) =
                make_uint2(pack2<T>(otot[dt][qt][0], otot[dt][qt][1]), pack2<T>(otot[dt][qt][2], otot[dt][qt][3]));
        }
    }
}


template <class T> struct Pages;
static __device__ __attribute__((aligned(16))) unsigned short g_pg_e0_bf16[8] = {0x3F80, 0, 0, 0, 0, 0, 0, 0};
static __device__ __attribute__((aligned(16))) unsigned short g_pg_e1_bf16[8] = {0, 0x3F80, 0, 0, 0, 0, 0, 0};
static __device__ __attribute__((aligned(16))) unsigned short g_pg_e0_f16[8] = {0x3C00, 0, 0, 0, 0, 0, 0, 0};
static __device__ __attribute__((aligned(16))) unsigned short g_pg_e1_f16[8] = {0, 0x3C00, 0, 0, 0, 0, 0, 0};
template <> struct Pages<BF16> {
    static __device__ __forceinline__ const unsigned char *e0() { return (const unsigned char *)g_pg_e0_bf16; }
    static __device__ __forceinline__ const unsigned char *e1() { return (const unsigned char *)g_pg_e1_bf16; }
};
template <> struct Pages<F16> {
    static __device__ __forceinline__ const unsigned char *e0() { return (const unsigned char *)g_pg_e0_f16; }
    static __device__ __forceinline__ const unsigned char *e1() { return (const unsigned char *)g_pg_e1_f16; }
};

// LDS-DMA of 16 bytes per lane of `mask`: lane l writes LDS[lds_dst + 16 l] (M0 carries the LDS base; this kernel has no other
// M0 user, so it is not saved).  EXEC is set inside the asm (it is all-ones at every call site): no compiler-made branches, and
// every wave issues exactly the same number of loads.  Issued from inline asm so that hipcc's waitcnt pass does not drain the
// loads with vmcnt(0) before every ds_read; ordering is by explicit vmcnt + s_barrier.
__device__ __forceinline__ void glds16_v(const void *gsrc, unsigned lds_dst, unsigned long long mask)     // 64-bit address per lane
{
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);        // wave-uniform by construction; the "s" constraint needs the compiler to know it
    asm volatile("s_mov_b32 m0, %1\n\ts_mov_b64 exec, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\ts_mov_b64 exec, -1"
                 :: "v"(gsrc), "s"(lds_dst), "s"(mask) : "memory");
}
__device__ __forceinline__ void glds16_s(const void *sbase, unsigned voff, unsigned lds_dst, unsigned long long mask)   // uniform base + lane offset
{
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_mov_b32 m0, %2\n\ts_mov_b64 exec, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_mov_b64 exec, -1"
                 :: "v"(voff), "s"(sbase), "s"(lds_dst), "s"(mask) : "memory");
}

template <int I, int N, class F> __device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <int N> __device__ __forceinline__ void wait_vmcnt()
{
    static_assert(N >= 0 && N < 64, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ int swz4(int row) { return ((row >> 1) ^ (row << 1) ^ (row << 2)) & 7; }   // conflict-free for both tiles (search: scripts/lds_swizzle_search.py)
}  // namespace
