// dn_attn.hip -- fused multi-KV-set flash attention for GaussCtrl's cross-view attention (gfx950).
//
// Replaces the unfused baddbmm + softmax + bmm chain of the reference's attention processor,
// /root/reference/gaussctrl/utils.py:25-37 (compute_attn) and :86-117 (CrossViewAttnProcessor.__call__):
//     O = a * softmax(Q K_self^T s) V_self + (1 - a)/4 * sum_{r<4} softmax(Q K_ref_r^T s) V_ref_r
// where K_ref_r / V_ref_r are frame r of the SAME CFG half broadcast over all frames of that half.
// The reference materialises five [2f*8, L, L] probability tensors per layer; here one Q tile visits up to
// five K/V sets, each with its own online softmax, and the weighted combination happens in registers.
// With one set of weight 1 it is ordinary attention (text cross-attention utils.py:111-117, the DDIM
// inversion path gc_pipeline.py:136-137).
//
// CDNA4 mapping (wave64, v_mfma_f32_16x16x32): both GEMMs are issued "transposed" --
//     S^T[key][q] = K Q^T       (A operand = K rows from LDS, B operand = Q rows kept in VGPRs)
//     O^T[d][q]  += V^T P^T     (A operand = V^T rows from LDS, B operand = P straight from the S registers)
// so a lane owns ONE query row (q = lane & 15): softmax statistics, the online rescale and the final
// 1/l are lane-local, P never leaves registers (the k-slot order of an MFMA is free as long as A and B
// agree), and the output store is 8 contiguous bytes per accumulator.  V arrives already transposed
// ([B][C][L], written by the projection GEMM's epilogue) so no transpose happens here.
#include "dn_attn_common.h"

void gc_dn_launch_attn5(const void *args, int dtype, int B, int depth, hipStream_t s);      // dn_attn5.hip

namespace {
using namespace dn;
template <class T, int D, int QT>
__global__ __launch_bounds__(256, 2) void k_attn(const AttnArgs a)
{
    __shared__ __attribute__((aligned(16))) unsigned char sK[SafeLds<D>::KBYTES];
    __shared__ __attribute__((aligned(16))) unsigned char sV[SafeLds<D>::VBYTES];
    int qblk, h, b;
    block_coords(a, QT, qblk, h, b);
    attn_safe_body<T, D, QT>(a, qblk, h, b, sK, sV);
}

// The set-split form with ALL queries of a (frame, head) in one workgroup (head size 160 at L = 256: 8 waves x 32 queries): the K / V^T of a set
// cross the L2 -> LDS path once per (frame, head, set) instead of once per 64-query block (154 -> 38 MB per launch, the measured bound of the
// 64-query form: profiles/r05_attn160_prefetch_ab.txt), and every LDS fragment feeds two MFMAs.
template <class T, int D, int QT, int NW>
__global__ __launch_bounds__(NW * 64, 1) void k_attn_wide(const AttnArgs a)
{
    __shared__ __attribute__((aligned(16))) unsigned char sK[SafeLds<D>::KBYTES];
    __shared__ __attribute__((aligned(16))) unsigned char sV[SafeLds<D>::VBYTES];
    int qblk, h, b;
    block_coords(a, QT, qblk, h, b);
    attn_safe_body<T, D, QT, NW, true>(a, qblk, h, b, sK, sV);
}

// out = round(sum_s part[s]) in a fixed order: the second launch of a set-split attention (below)
template <class T>
__global__ __launch_bounds__(256) void k_attn_combine(const float *__restrict__ part, int nsets, int64_t n4, int64_t set_stride4, unsigned short *__restrict__ O,
                                                      int64_t ldo, int64_t o_bs, int64_t row4, int64_t rows_per_batch)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 v = reinterpret_cast<const float4 *>(part)[i];
        for (int s = 1; s < nsets; ++s) {
            const float4 w = reinterpret_cast<const float4 *>(part)[i + s * set_stride4];
            v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
        const int64_t row = i / row4, c4 = i - row * row4, b = row / rows_per_batch, q = row - b * rows_per_batch;
        *reinterpret_cast<uint2 *>(O + b * o_bs + q * ldo + c4 * 4) = make_uint2(pack2<T>(v.x, v.y), pack2<T>(v.z, v.w));
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// k_attn3: static-offset, software-pipelined form for head sizes with two spare contraction columns (D % 32 != 0: 40, 80).
//
// The SIMD issues MFMA and VALU work almost serially (scripts/ubench/issue_model.hip: a v_mfma_16x16x32 costs ~17 clk and
// hides only ~2 plain VALU; v_exp_f32 costs ~8 clk), and the online softmax of k_attn spends 2/3 of its time in VALU
// instructions.  Here the softmax bookkeeping moves INTO the QK^T MFMAs:
//   * Q is pre-multiplied by scale*log2(e) once; contraction column D of every Q row holds -m0 (the row's offset, = the row
//     maximum over the FIRST key tile of the K/V set) and column D+1 holds -30000.  Column D of a valid key row in LDS is 1,
//     column D+1 is 1 only for keys >= Lk.  The MFMA therefore returns S' = log2e*scale*q.k - m0 (or <= -30000 for masked keys)
//     and P = exp2(S') is ONE v_exp_f32 per element: no running maximum, no rescale of O, no masking VALU, no FMA.
//     (softmax is invariant to the offset; m0 only keeps exp2 in range.)  The softmax denominator comes out of the P V MFMAs
//     through a row of ones in V^T, as in k_attn.
//   * If a later tile exceeds m0 by more than the exponent range (P or its sum overflows: the denominator is not a positive
//     finite number), the workgroup recomputes its rows with the safe online-softmax body.  Results never depend on the data
//     being "nice"; only the speed does.
//   * K / V^T tiles travel HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, SGPR base + lane offset: no VGPR staging, no address
//     VALU) into an NST-deep ring, PD = NST-1 tiles ahead; ONE s_barrier per tile and counted s_waitcnt vmcnt(N).
//   * (set, tile) pairs are flattened into one sequence of steps; the QK^T MFMAs of step i+1 are issued between the exp /
//     convert slices of step i, the P V MFMAs of query tile 0 between those of query tile 1.
//   * the MFMA row -> key mapping inside a tile is permuted (key = 32(kt/2) + 8g + 4(kt&1) + r) so that a lane's eight P values
//     of a 32-key block are eight CONSECUTIVE keys: the V^T fragment is one ds_read_b128 (ds_read2_b64 runs at half rate).
//   * both tiles use 16-byte-chunk XOR swizzles that are conflict-free for the ds_read_b128 lane groups of gfx950
//     (exhaustive search: scripts/lds_swizzle_search.py; SQ_LDS_BANK_CONFLICT = 0 measured).
template <int CPR> __device__ __forceinline__ int swz_k(int row)
{
    if (CPR == 8) return (row ^ (row << 1) ^ ((row >> 3) << 2)) & 7;
    return (row & 1) | (((row >> 3) & 1) << 1);
}
__device__ __forceinline__ int swz_v(int row) { return (row ^ (row << 1) ^ (row << 2)) & 7; }

// RAG: Lk is not a multiple of the 64-key tile.  PRE: Q arrives already multiplied by scale*log2e (folded into the
// projection weights); otherwise the offset column is kept in raw-logit units and the multiply happens before v_exp_f32.
template <class T, int D, int QT, int NST, bool RAG, bool PRE>
__global__ __launch_bounds__(256, 2) void k_attn3(const AttnArgs a)
{
    static_assert(D % 8 == 0 && D % 32 != 0, "needs two spare contraction columns");
    constexpr int DP = (D + 31) / 32 * 32, KS = DP / 32, CPR = DP / 8;     // K row: CPR 16-byte chunks
    constexpr int DV = (D + 16) / 16 * 16, DT = DV / 16;                   // V^T rows: D channels, the ones row, zero rows
    constexpr int KBYTES = 64 * CPR * 16, VBYTES = DV * 128;
    constexpr int LCS = D / 8, KSS = LCS / 4, GS = LCS % 4;                // the special chunk: columns D (offset) and D+1 (mask)
    constexpr int KI = CPR / 4;                                            // K DMA instructions per wave per tile (8 rows each)
    constexpr int VSH = 2 * D, VI = (VSH + 63) / 64;                       // V^T chunks per wave per tile, DMA instructions
    constexpr int PD = NST - 1;                                            // prefetch distance (tiles)
    constexpr float BIG = 30000.f;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *sK = smem, *sV = smem + NST * KBYTES;
    const unsigned ldsK = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem, ldsV = ldsK + NST * KBYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    int qblk, h, b;
    block_coords(a, QT, qblk, h, b);
    const int q_wave0 = (qblk * 4 + wid) * (QT * 16);
    const int ntiles = (a.Lk + 63) / 64;
    const int nsteps = a.nsets * ntiles;

    // ---- LDS image, written once: zeros, column D of every key row = 1, the ones row of V^T
    for (int i = tid; i < NST * (KBYTES + VBYTES) / 16; i += 256) reinterpret_cast<uint4 *>(smem)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    for (int i = tid; i < NST * 64; i += 256) {
        const int st = i >> 6, row = i & 63;
        *reinterpret_cast<unsigned short *>(sK + st * KBYTES + row * (CPR * 16) + ((LCS ^ swz_k<CPR>(row)) << 4)) = One<T>::v;
    }
    for (int i = tid; i < NST * 64; i += 256) reinterpret_cast<unsigned short *>(sV + (i >> 6) * VBYTES + D * 128)[i & 63] = One<T>::v;

    // ---- Q fragments (B operand): lane holds Q[q = fr][d = 32*ks + 8*g .. +8]
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
    const float c2 = a.scale_log2e;       // PRE: 1
    float moff[QT];                       // current offset of each query row (exactly representable in T)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        moff[qt] = 0.f;
        if (g == GS) qf[qt][KSS].x = pack2<T>(0.f, -BIG);
    }

    // ---- LDS-DMA plan of this lane
    // K: instruction j of wave w covers chunk slots [(4j + w) * 64, +64) of the tile image (8 key rows); only data chunks load
    int k_off[KI], k_row[KI];
    unsigned long long k_msk[KI], k_smk[KI];
#pragma unroll
    for (int j = 0; j < KI; ++j) {
        const int p = (j * 4 + wid) * 64 + lane, row = p / CPR, lc = (p % CPR) ^ swz_k<CPR>(row);
        k_row[j] = row;
        k_msk[j] = __ballot(lc * 8 < D);
        k_smk[j] = __ballot(lc == LCS);
        k_off[j] = (row * (int)a.ldk + lc * 8) * 2;
    }
    // V^T: wave w owns chunk slots [VSH w, VSH (w+1)) of the D x 8 data chunks
    int v_off[VI], v_tok[VI];
    unsigned long long v_msk[VI];
#pragma unroll
    for (int j = 0; j < VI; ++j) {
        const int q = j * 64 + lane, p = VSH * wid + q, row = p >> 3, lc = (p & 7) ^ swz_v(row);
        v_msk[j] = __ballot(q < VSH);
        v_tok[j] = lc * 8;
        v_off[j] = (row * (int)a.ldvt + lc * 8) * 2;
    }
    const int lk8 = (a.Lk + 7) / 8 * 8;
    constexpr int GRP = KI * (RAG ? 2 : 1) + VI;          // DMA instructions per wave per step

    // per-set K / V^T base addresses, parked in lane s of two VGPR pairs (fetched with v_readlane at set changes)
    unsigned long long kb_tab = 0, vb_tab = 0;
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        if (s < a.nsets) {
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
            if (lane == s) { kb_tab = (unsigned long long)Kb; vb_tab = (unsigned long long)Vb; }
        }
    }
    auto tab = [&](unsigned long long t, int s) __attribute__((always_inline)) -> const unsigned char * {
        const unsigned lo = __builtin_amdgcn_readlane((unsigned)t, s), hi = __builtin_amdgcn_readlane((unsigned)(t >> 32), s);
        return (const unsigned char *)(((unsigned long long)hi << 32) | lo);
    };
    // DMA cursors: the K stream runs one tile ahead of the V^T stream; both walk the (set, tile) sequence and, past the end,
    // load the last set's first tile again (the counted waits need the same number of loads from every wave, every step)
    struct Cur { const unsigned char *p; int tile, s; unsigned dst; };
    Cur ck, cv;
    ck.p = tab(kb_tab, 0); ck.tile = 0; ck.s = 0; ck.dst = ldsK + wid * 1024;
    cv.p = tab(vb_tab, 0); cv.tile = 0; cv.s = 0; cv.dst = ldsV + wid * (VSH * 16);
    const int64_t kstride = (int64_t)128 * a.ldk;
    auto issue_k = [&]() __attribute__((always_inline)) {
        if (!RAG) {
#pragma unroll
            for (int j = 0; j < KI; ++j) glds16_s(ck.p, (unsigned)k_off[j], ck.dst + j * 4096, k_msk[j]);
        } else {       // keys >= Lk: the data chunks re-read row 0 of the tile (finite filler), column D+1 = 1 masks them
            const int lim = a.Lk - ck.tile * 64;
#pragma unroll
            for (int j = 0; j < KI; ++j) {
                const bool ok = k_row[j] < lim;
                glds16_s(ck.p, (unsigned)(ok ? k_off[j] : k_off[j] - k_row[j] * (int)a.ldk * 2), ck.dst + j * 4096, k_msk[j]);
                glds16_v(ok ? Pages<T>::e0() : Pages<T>::e1(), ck.dst + j * 4096, k_smk[j]);
            }
        }
        ck.dst = ck.dst + KBYTES == ldsK + wid * 1024 + NST * KBYTES ? ldsK + wid * 1024 : ck.dst + KBYTES;
        ck.p += kstride;
        if (++ck.tile == ntiles) { ck.tile = 0; ck.s = ck.s + 1 < a.nsets ? ck.s + 1 : ck.s; ck.p = tab(kb_tab, ck.s); }
    };
    auto issue_v = [&]() __attribute__((always_inline)) {
        const int lim = lk8 - cv.tile * 64;    // token chunks past round_up(Lk, 8) re-read chunk 0 (their P is exactly 0)
#pragma unroll
        for (int j = 0; j < VI; ++j)
            glds16_s(cv.p, (unsigned)(!RAG || v_tok[j] < lim ? v_off[j] : v_off[j] - v_tok[j] * 2), cv.dst + j * 1024, v_msk[j]);
        cv.dst = cv.dst + VBYTES == ldsV + wid * (VSH * 16) + NST * VBYTES ? ldsV + wid * (VSH * 16) : cv.dst + VBYTES;
        cv.p += 128;
        if (++cv.tile == ntiles) { cv.tile = 0; cv.s = cv.s + 1 < a.nsets ? cv.s + 1 : cv.s; cv.p = tab(vb_tab, cv.s); }
    };

    // fragment read offsets.  K: MFMA row fr of key-subtile kt is key 32(kt/2) + 4(kt&1) + 8(fr/4) + (fr&3); chunk 4ks + g
    int kfo[KS][2];
    {
        const int row0 = 8 * (fr >> 2) + (fr & 3);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int od = 0; od < 2; ++od) {
                const int row = row0 + 4 * od;
                kfo[ks][od] = row * (CPR * 16) + (((4 * ks + g) ^ swz_k<CPR>(row)) << 4);
            }
    }
    // V^T: row 16 dt + fr, chunk 4 kb + g (tokens 32 kb + 8 g .. +8)
    int vfo[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) vfo[kb] = fr * 128 + (((4 * kb + g) ^ swz_v(fr)) << 4);

    f32x4 otot[DT][QT], os[DT][QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) { otot[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f}; os[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    int bad = 0;

    constexpr int NKF = 4 * KS, NQ = NKF * QT, NPV = 2 * DT, NU = 8;      // K fragments, QK MFMAs, PV MFMAs / exp units per query tile

    // one pipeline step.  cur = S'(i) (already holds log2e*scale*q.k - offset); nxt receives S'(i+1)
    const unsigned char *rk = sK + (NST > 1 ? KBYTES : 0), *rv = sV;     // LDS stages step i reads: K(i+1), V(i)
    auto body = [&](f32x4(&cur)[4][QT], f32x4(&nxt)[4][QT], auto first_tag) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value;
        wait_vmcnt<(PD - 1) * GRP>();          // K(i+1), V(i) (and everything older) have landed for this wave
        __builtin_amdgcn_s_barrier();          // ... for every wave; every wave is done with step i-1's buffers
        issue_k();
        issue_v();
        const unsigned char *kb_ = rk, *vb_ = rv;
        rk = rk + KBYTES == sK + NST * KBYTES ? sK : rk + KBYTES;
        rv = rv + VBYTES == sV + NST * VBYTES ? sV : rv + VBYTES;
        if (FIRST) {
            // first tile of a K/V set: its row maximum becomes the set's offset.  cur was produced with the previous offset.
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                float t = fmaxf(fmaxf(cur[0][qt][0], cur[0][qt][1]), cur[0][qt][2]);
                t = fmaxf(fmaxf(t, cur[0][qt][3]), cur[1][qt][0]);
#pragma unroll
                for (int kt = 1; kt < 4; ++kt) {
                    if (kt > 1) t = fmaxf(fmaxf(t, cur[kt - 1][qt][3]), cur[kt][qt][0]);
                    t = fmaxf(fmaxf(t, cur[kt][qt][1]), cur[kt][qt][2]);
                }
                t = fmaxf(t, cur[3][qt][3]);
                const unsigned x = __float_as_uint(t);                       // reduce over g with the lane-swap instructions
                const auto r1 = __builtin_amdgcn_permlane32_swap(x, x, false, false);
                t = fmaxf(__uint_as_float(r1[0]), __uint_as_float(r1[1]));
                const unsigned y = __float_as_uint(t);
                const auto r2 = __builtin_amdgcn_permlane16_swap(y, y, false, false);
                t = fmaxf(__uint_as_float(r2[0]), __uint_as_float(r2[1]));
                const float mq = T::to_f(T::from_f(moff[qt] + t));           // new offset, rounded to what the MFMA will see
                const float dlt = mq - moff[qt];
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) cur[kt][qt][r] -= dlt;
                moff[qt] = mq;
                if (g == GS) qf[qt][KSS].x = pack2<T>(-mq, -BIG);
            }
        }
        uint4 pf[QT][2], kf[NKF], vf[NPV];
        auto rd_k = [&](int n) __attribute__((always_inline)) {       // n = ks * 4 + kt
            const int ks = n >> 2, kt = n & 3;
            kf[n] = *reinterpret_cast<const uint4 *>(kb_ + kfo[ks][kt & 1] + (32 * (kt >> 1)) * (CPR * 16));
        };
        auto rd_v = [&](int n) __attribute__((always_inline)) {       // n = kb * DT + dt
            const int kb = n / DT, dt = n - kb * DT;
            vf[n] = *reinterpret_cast<const uint4 *>(vb_ + vfo[kb] + dt * 2048);
        };
        auto mma_qk = [&](int m) __attribute__((always_inline)) {     // m = (ks * 4 + kt) * QT + qt
            const int n = m / QT, qt = m - n * QT, ks = n >> 2, kt = n & 3;
            if (ks == 0) nxt[kt][qt] = T::mfma(kf[n], qf[qt][ks], f32x4{0.f, 0.f, 0.f, 0.f});
            else nxt[kt][qt] = T::mfma(kf[n], qf[qt][ks], nxt[kt][qt]);
        };
        auto mma_pv = [&](int qt, int n) __attribute__((always_inline)) {
            const int kb = n / DT, dt = n - kb * DT;
            os[dt][qt] = T::mfma(vf[n], pf[qt][kb], os[dt][qt]);
        };
        // softmax numerators of one query tile in 8 units of 2 v_exp_f32 + 1 packed convert; word (kt&1)*2 + (j&1) of
        // pf[kt>>1] = keys 32(kt>>1) + 8g + 4(kt&1) + {r0, r0+1}
        auto unit = [&](auto qt_, auto j_) __attribute__((always_inline)) {
            constexpr int qt = decltype(qt_)::value, j = decltype(j_)::value, kt = j >> 1, r0 = 2 * (j & 1);
            const float x0 = PRE ? cur[kt][qt][r0] : cur[kt][qt][r0] * c2, x1 = PRE ? cur[kt][qt][r0 + 1] : cur[kt][qt][r0 + 1] * c2;
            const unsigned w = pack2<T>(__builtin_amdgcn_exp2f(x0), __builtin_amdgcn_exp2f(x1));
            constexpr int c = (kt & 1) * 2 + (j & 1);
            if constexpr (c == 0) pf[qt][kt >> 1].x = w;
            else if constexpr (c == 1) pf[qt][kt >> 1].y = w;
            else if constexpr (c == 2) pf[qt][kt >> 1].z = w;
            else pf[qt][kt >> 1].w = w;
        };
        auto units = [&](auto lo_, auto hi_) __attribute__((always_inline)) {
            constexpr int lo = decltype(lo_)::value, hi = decltype(hi_)::value;
            static_for<lo, hi>([&](auto x_) __attribute__((always_inline)) {
                constexpr int x = decltype(x_)::value;
                unit(std::integral_constant<int, x / NU>{}, std::integral_constant<int, x % NU>{});
            });
        };
        constexpr int UT = NU * QT;
        constexpr int M0 = NQ / QT;                               // QK MFMAs issued before P V of query tile 0 may start
        rd_k(0);
        if (NKF > 1) rd_k(1);
        if (NKF > 2) rd_k(2);
        if (NKF > 3) rd_k(3);
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, NQ>([&](auto m_) __attribute__((always_inline)) {
            constexpr int m = decltype(m_)::value;
            if constexpr (m % QT == 0 && m / QT + 4 < NKF) rd_k(m / QT + 4);
            if constexpr (QT > 1 && m >= 2 && m - 2 < NPV) rd_v(m - 2);      // V^T fragments: far ahead of their MFMAs (LDS latency)
            mma_qk(m);
            units(std::integral_constant<int, (UT * m) / NQ>{}, std::integral_constant<int, (UT * (m + 1)) / NQ>{});
            if constexpr (QT > 1 && m >= M0 && m - M0 < NPV) {    // P V of query tile 0 between the exps of tile 1
                __builtin_amdgcn_sched_barrier(0);
                mma_pv(0, m - M0);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (QT == 1) {
            static_for<0, NPV>([&](auto n_) __attribute__((always_inline)) { rd_v(decltype(n_)::value); });
        }
        static_for<0, QT>([&](auto qt_) __attribute__((always_inline)) {
            constexpr int qt = decltype(qt_)::value;
            constexpr int n0 = (qt == 0 && QT > 1) ? (NQ - M0 < NPV ? NQ - M0 : NPV) : 0;    // already issued above
            static_for<n0, NPV>([&](auto n_) __attribute__((always_inline)) {
                mma_pv(qt, decltype(n_)::value);
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    };

    // end of a K/V set: O_total += w / l * O_set; the denominator l is row D of O^T (the ones row of V^T)
    auto fold = [&](int s) __attribute__((always_inline)) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            constexpr int dt_l = D / 16, g_l = (D % 16) / 4, r_l = D % 4;
            const float l = __shfl(os[dt_l][qt][r_l], g_l * 16 + fr, 64);
            bad |= !(l > 0.f && l < 1e37f);
            const float inv = a.set_w[s] / l;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { otot[dt][qt][r] += os[dt][qt][r] * inv; os[dt][qt][r] = 0.f; }
        }
    };

    // ---- prologue: K(0) alone, then PD groups {K(j+1), V(j)}
    __syncthreads();                 // LDS image complete before the first DMA lands on it
    issue_k();
#pragma unroll
    for (int j = 0; j < PD; ++j) { issue_k(); issue_v(); }
    wait_vmcnt<PD * GRP>();
    __builtin_amdgcn_s_barrier();
    f32x4 sa[4][QT], sb[4][QT];
    {
        const unsigned char *kb_ = sK;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) sa[kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const uint4 kf = *reinterpret_cast<const uint4 *>(kb_ + kfo[ks][kt & 1] + (32 * (kt >> 1)) * (CPR * 16));
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) sa[kt][qt] = T::mfma(kf, qf[qt][ks], sa[kt][qt]);
            }
    }

    int tile = 0, s = 0;
    auto step = [&](f32x4(&cur)[4][QT], f32x4(&nxt)[4][QT]) __attribute__((always_inline)) {
        if (tile == 0) body(cur, nxt, std::true_type{});
        else body(cur, nxt, std::false_type{});
        if (++tile == ntiles) { fold(s); tile = 0; ++s; }
    };
    int i = 0;
    for (; i + 1 < nsteps; i += 2) { step(sa, sb); step(sb, sa); }
    if (i < nsteps) step(sa, sb);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // drain the tail prefetches before the LDS is reused / released

    if (__syncthreads_or(bad)) {      // some row left the exponent range of its first-tile offset: safe recomputation
        attn_safe_body<T, D, QT>(a, qblk, h, b, smem, smem + SafeLds<D>::KBYTES);
        return;
    }
    // ---- store: lane owns O[q = fr][d = 16*dt + 4*g .. +4]
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

// ------------------------------------------------------------------------------------------------------------------------
// k_attn4: the k_attn3 scheme (offset / mask columns inside the QK^T contraction, denominator from a ones row of V^T, LDS-DMA
// ring, flattened (set, tile) steps, in-kernel safe fallback) on v_mfma_f32_32x32x16.  Why: scripts/ubench/issue_model.hip (round 2,
// profiles/r02_issue_model_32x32.txt) -- a 32x32x16 MFMA (33.5 clk) hides ~3 v_exp_f32 / ~4 plain VALU in its shadow, a 16x16x32
// (18.5 clk, half the flops) hides only one v_exp_f32; per unit of matrix work the big tile carries 1.5x the softmax VALU for free,
// and D = 40 pads to 48 contraction columns (3 k-steps of 16) instead of 64.
// One wave owns 32 query rows (lane & 31; the two lane halves hold different key / channel rows).  Per 64-key step:
//   S'^T[key][q]: 2 key blocks x KS k-steps     A = K rows (LDS), B = Q (registers)                     6 MFMAs  (D = 40)
//   O^T[d][q]  += V^T P^T: DB row blocks x 4    A = V^T rows (LDS), B = P packed straight from S'^T     8 MFMAs
// MFMA row i of a key block is key pi(i) = i with bits 2 and 3 swapped, so that the eight S'^T registers a lane feeds to one P V k-step
// are eight CONSECUTIVE keys (V^T fragment = one ds_read_b128).

// QB: 32-query blocks per wave.  QB = 2 is the one-wave-per-SIMD form (512 registers): every K / V^T fragment read from LDS feeds two
// MFMAs and the workgroup streams K / V^T once for 256 queries -- half the LDS fragment traffic, LDS-DMA and barriers per unit of work.
template <class T, int D, int NST, int NW, bool RAG, bool PRE, int PF = 3, int QB = 1>
__global__ __launch_bounds__(NW * 64, NW == 4 && QB == 1 ? 2 : 1) void k_attn4(const AttnArgs a)
{
    static_assert(D % 8 == 0 && D % 16 != 0, "needs two spare contraction columns inside the 16-column padding");
    constexpr int DP = (D + 15) / 16 * 16, KS = DP / 16;                   // contraction length of QK^T, k-steps
    constexpr int DB = (D + 32) / 32;                                      // 32-row blocks of O^T: D channels, the ones row, zero rows
    static_assert(DP <= 64, "key rows keep a 128-byte LDS pitch");
    constexpr int KBYTES = 64 * 128, VBYTES = DB * 32 * 128;
    constexpr int LCS = D / 8, KSS = LCS / 2, HS = LCS & 1;                // the special chunk: columns D (offset) and D+1 (mask)
    constexpr int NT = NW * 64, KI = 8 / NW;                               // K DMA instructions per wave per tile (8 rows each)
    static_assert(NW == 4 || NW == 8, "4 or 8 waves");
    constexpr int VSH = 8 * D / NW, VI = (VSH + 63) / 64;                       // V^T chunks per wave per tile, DMA instructions
    constexpr int PD = NST - 1;
    constexpr float BIG = 30000.f;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *sK = smem, *sV = smem + NST * KBYTES;
    const unsigned ldsK = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem, ldsV = ldsK + NST * KBYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qi = lane & 31, hg = lane >> 5;
    int qblk, h, b;
    block_coords(a, 2, qblk, h, b);
    const int q_wave0 = (qblk * NW + wid) * (32 * QB);
    const int ntiles = (a.Lk + 63) / 64;
    const int nsteps = a.nsets * ntiles;

    // ---- LDS image, written once: zeros, column D of every key row = 1, the ones row of V^T
    for (int i = tid; i < NST * (KBYTES + VBYTES) / 16; i += NT) reinterpret_cast<uint4 *>(smem)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    for (int i = tid; i < NST * 64; i += NT) {
        const int st = i >> 6, row = i & 63;
        *reinterpret_cast<unsigned short *>(sK + st * KBYTES + row * 128 + ((LCS ^ swz4(row)) << 4)) = One<T>::v;
    }
    for (int i = tid; i < NST * 64; i += NT) reinterpret_cast<unsigned short *>(sV + (i >> 6) * VBYTES + D * 128)[i & 63] = One<T>::v;

    // ---- Q fragments (B operand): lane holds Q[q = qi][d = 16 ks + 8 hg .. +8]
    uint4 qf[QB][KS];
    float moff[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int q = q_wave0 + 32 * qb + qi;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int d = ks * 16 + hg * 8;
            qf[qb][ks] = (q < a.Lq && d + 8 <= D)
                             ? *reinterpret_cast<const uint4 *>(a.Q + (int64_t)b * a.q_bs + (int64_t)q * a.ldq + h * D + d)
                             : make_uint4(0, 0, 0, 0);
        }
        moff[qb] = 0.f;
        if (hg == HS) qf[qb][KSS].x = pack2<T>(0.f, -BIG);
    }
    const float c2 = a.scale_log2e;

    // ---- LDS-DMA plan of this lane (as k_attn3, 8 chunk slots per key row)
    int k_off[KI], k_row[KI];
    unsigned long long k_msk[KI], k_smk[KI];
#pragma unroll
    for (int j = 0; j < KI; ++j) {
        const int p = (j * NW + wid) * 64 + lane, row = p >> 3, lc = (p & 7) ^ swz4(row);
        k_row[j] = row;
        k_msk[j] = __ballot(lc * 8 < D);
        k_smk[j] = __ballot(lc == LCS);
        k_off[j] = (row * (int)a.ldk + lc * 8) * 2;
    }
    int v_off[VI], v_tok[VI];
    unsigned long long v_msk[VI];
#pragma unroll
    for (int j = 0; j < VI; ++j) {
        const int q = j * 64 + lane, p = VSH * wid + q, row = p >> 3, lc = (p & 7) ^ swz4(row);
        v_msk[j] = __ballot(q < VSH);
        v_tok[j] = lc * 8;
        v_off[j] = (row * (int)a.ldvt + lc * 8) * 2;
    }
    const int lk8 = (a.Lk + 7) / 8 * 8;
    constexpr int GRP = KI * (RAG ? 2 : 1) + VI;

    unsigned long long kb_tab = 0, vb_tab = 0;
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        if (s < a.nsets) {
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
            if (lane == s) { kb_tab = (unsigned long long)Kb; vb_tab = (unsigned long long)Vb; }
        }
    }
    auto tab = [&](unsigned long long t, int s) __attribute__((always_inline)) -> const unsigned char * {
        const unsigned lo = __builtin_amdgcn_readlane((unsigned)t, s), hi = __builtin_amdgcn_readlane((unsigned)(t >> 32), s);
        return (const unsigned char *)(((unsigned long long)hi << 32) | lo);
    };
    struct Cur { const unsigned char *p; int tile, s; unsigned dst; };
    Cur ck, cv;
    ck.p = tab(kb_tab, 0); ck.tile = 0; ck.s = 0; ck.dst = ldsK + wid * 1024;
    cv.p = tab(vb_tab, 0); cv.tile = 0; cv.s = 0; cv.dst = ldsV + wid * (VSH * 16);
    const int64_t kstride = (int64_t)128 * a.ldk;
    auto issue_k = [&]() __attribute__((always_inline)) {
        if (!RAG) {
#pragma unroll
            for (int j = 0; j < KI; ++j) glds16_s(ck.p, (unsigned)k_off[j], ck.dst + j * (NW * 1024), k_msk[j]);
        } else {
            const int lim = a.Lk - ck.tile * 64;
#pragma unroll
            for (int j = 0; j < KI; ++j) {
                const bool ok = k_row[j] < lim;
                glds16_s(ck.p, (unsigned)(ok ? k_off[j] : k_off[j] - k_row[j] * (int)a.ldk * 2), ck.dst + j * (NW * 1024), k_msk[j]);
                glds16_v(ok ? Pages<T>::e0() : Pages<T>::e1(), ck.dst + j * (NW * 1024), k_smk[j]);
            }
        }
        ck.dst = ck.dst + KBYTES == ldsK + wid * 1024 + NST * KBYTES ? ldsK + wid * 1024 : ck.dst + KBYTES;
        ck.p += kstride;
        if (++ck.tile == ntiles) { ck.tile = 0; ck.s = ck.s + 1 < a.nsets ? ck.s + 1 : ck.s; ck.p = tab(kb_tab, ck.s); }
    };
    auto issue_v = [&]() __attribute__((always_inline)) {
        const int lim = lk8 - cv.tile * 64;
#pragma unroll
        for (int j = 0; j < VI; ++j)
            glds16_s(cv.p, (unsigned)(!RAG || v_tok[j] < lim ? v_off[j] : v_off[j] - v_tok[j] * 2), cv.dst + j * 1024, v_msk[j]);
        cv.dst = cv.dst + VBYTES == ldsV + wid * (VSH * 16) + NST * VBYTES ? ldsV + wid * (VSH * 16) : cv.dst + VBYTES;
        cv.p += 128;
        if (++cv.tile == ntiles) { cv.tile = 0; cv.s = cv.s + 1 < a.nsets ? cv.s + 1 : cv.s; cv.p = tab(vb_tab, cv.s); }
    };

    // fragment read offsets.  K: MFMA row qi of key block kb is key 32 kb + pi(qi); chunk 2 ks + hg.  V^T: row 32 db + qi, chunk 2 t + hg
    int kfo[KS], vfo[4];
    {
        const int row = (qi & ~12) | (((qi >> 2) & 1) << 3) | (((qi >> 3) & 1) << 2);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kfo[ks] = row * 128 + (((2 * ks + hg) ^ swz4(row)) << 4);
#pragma unroll
        for (int t = 0; t < 4; ++t) vfo[t] = qi * 128 + (((2 * t + hg) ^ swz4(qi)) << 4);
    }

    f32x16 otot[QB][DB], os[QB][DB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) { otot[qb][db][r] = 0.f; os[qb][db][r] = 0.f; }
    int bad = 0;

    constexpr int NKF = 2 * KS, NVF = 4 * DB;                              // K / V^T fragments per step (each feeds QB MFMAs)
    constexpr int NQ = NKF * QB, NPV = NVF * QB, NM = NQ + NPV, NU = 16 * QB;   // QK MFMAs, PV MFMAs, exp units (2 v_exp + 1 cvt_pk)
    constexpr int NG = NM - DB * QB;                                      // MFMA gaps that carry exp units (the last k-step's P V needs all of them)

    const unsigned char *rk = sK + (NST > 1 ? KBYTES : 0), *rv = sV;
    auto body = [&](f32x16(&cur)[QB][2], f32x16(&nxt)[QB][2], auto first_tag) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value;
        wait_vmcnt<(PD - 1) * GRP>();          // K(i+1), V(i) (and everything older) have landed for this wave
        __builtin_amdgcn_s_barrier();          // ... for every wave; every wave is done with step i-1's buffers
        issue_k();
        issue_v();
        const unsigned char *kb_ = rk, *vb_ = rv;
        rk = rk + KBYTES == sK + NST * KBYTES ? sK : rk + KBYTES;
        rv = rv + VBYTES == sV + NST * VBYTES ? sV : rv + VBYTES;
        if (FIRST) {      // first tile of a K/V set: its row maximum becomes the set's offset (cur still carries the previous one)
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                float t = fmaxf(fmaxf(cur[qb][0][0], cur[qb][0][1]), cur[qb][0][2]);
#pragma unroll
                for (int r = 3; r + 1 < 16; r += 2) t = fmaxf(fmaxf(t, cur[qb][0][r]), cur[qb][0][r + 1]);
                t = fmaxf(fmaxf(t, cur[qb][0][15]), cur[qb][1][0]);
#pragma unroll
                for (int r = 1; r + 1 < 16; r += 2) t = fmaxf(fmaxf(t, cur[qb][1][r]), cur[qb][1][r + 1]);
                t = fmaxf(t, cur[qb][1][15]);
                const unsigned x = __float_as_uint(t);
                const auto r1 = __builtin_amdgcn_permlane32_swap(x, x, false, false);
                t = fmaxf(__uint_as_float(r1[0]), __uint_as_float(r1[1]));
                const float mq = T::to_f(T::from_f(moff[qb] + t));
                const float dlt = mq - moff[qb];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) cur[qb][kb][r] -= dlt;
                moff[qb] = mq;
                if (hg == HS) qf[qb][KSS].x = pack2<T>(-mq, -BIG);
            }
        }
        uint4 pf[QB][4], kf[NKF], vf[NVF];
        auto rd_k = [&](int n) __attribute__((always_inline)) {       // n = ks * 2 + kb
            const int ks = n >> 1, kb = n & 1;
            kf[n] = *reinterpret_cast<const uint4 *>(kb_ + kfo[ks] + kb * 4096);
        };
        auto rd_v = [&](int n) __attribute__((always_inline)) {       // n = t * DB + db
            const int t = n / DB, db = n - t * DB;
            vf[n] = *reinterpret_cast<const uint4 *>(vb_ + vfo[t] + db * 4096);
        };
        auto mma_qk = [&](int m) __attribute__((always_inline)) {     // m = fragment * QB + query block
            const int n = m / QB, qb = m - n * QB, ks = n >> 1, kb = n & 1;
            if (ks == 0) {
                f32x16 z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.f;
                nxt[qb][kb] = T::mfma32(kf[n], qf[qb][ks], z);
            } else nxt[qb][kb] = T::mfma32(kf[n], qf[qb][ks], nxt[qb][kb]);
        };
        auto mma_pv = [&](int m) __attribute__((always_inline)) {
            const int n = m / QB, qb = m - n * QB, t = n / DB, db = n - t * DB;
            os[qb][db] = T::mfma32(vf[n], pf[qb][t], os[qb][db]);
        };
        // exp unit x: k-step t = x / 4 (keys 32 (t/2) + 16 (t&1) + 8 hg ..+8), word w = x % 4 = registers 8 (t&1) + 2 w, +1 of cur[t/2]
        auto unit = [&](auto x_) __attribute__((always_inline)) {     // units ordered by (k-step, word, query block)
            constexpr int x = decltype(x_)::value, qb = x % QB, y = x / QB, t = y >> 2, w = y & 3, kb = t >> 1, r0 = 8 * (t & 1) + 2 * w;
            const float x0 = PRE ? cur[qb][kb][r0] : cur[qb][kb][r0] * c2, x1 = PRE ? cur[qb][kb][r0 + 1] : cur[qb][kb][r0 + 1] * c2;
            const unsigned v = pack2<T>(__builtin_amdgcn_exp2f(x0), __builtin_amdgcn_exp2f(x1));
            if constexpr (w == 0) pf[qb][t].x = v;
            else if constexpr (w == 1) pf[qb][t].y = v;
            else if constexpr (w == 2) pf[qb][t].z = v;
            else pf[qb][t].w = v;
        };
        static_for<0, PF>([&](auto n_) __attribute__((always_inline)) { rd_k(decltype(n_)::value); });
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, NM>([&](auto m_) __attribute__((always_inline)) {
            constexpr int m = decltype(m_)::value, fr = m / QB + PF;               // fragments PF fragment-uses ahead
            if constexpr (m % QB == 0 && fr < NKF) rd_k(fr);
            if constexpr (m % QB == 0 && fr >= NKF && fr - NKF < NVF) rd_v(fr - NKF);
            if constexpr (m < NQ) mma_qk(m);
            else mma_pv(m - NQ);
            if constexpr (m < NG) {
                static_for<(NU * m) / NG, (NU * (m + 1)) / NG>([&](auto x_) __attribute__((always_inline)) { unit(x_); });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    // end of a K/V set: O_total += w / l * O_set; the denominator l is row D of O^T (the ones row of V^T)
    auto fold = [&](int s) __attribute__((always_inline)) {
        constexpr int db_l = D / 32, dl = D % 32, r_l = (dl >> 3) * 4 + (dl & 3), hg_l = (dl >> 2) & 1;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const float l = __shfl(os[qb][db_l][r_l], qi + 32 * hg_l, 64);
            bad |= !(l > 0.f && l < 1e37f);
            const float inv = a.set_w[s] / l;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) { otot[qb][db][r] += os[qb][db][r] * inv; os[qb][db][r] = 0.f; }
        }
    };

    // ---- prologue: K(0) alone, then PD groups {K(j+1), V(j)}
    __syncthreads();
    issue_k();
#pragma unroll
    for (int j = 0; j < PD; ++j) { issue_k(); issue_v(); }
    wait_vmcnt<PD * GRP>();
    __builtin_amdgcn_s_barrier();
    f32x16 sa[QB][2], sb[QB][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int r = 0; r < 16; ++r) sa[qb][kb][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const uint4 kfr = *reinterpret_cast<const uint4 *>(sK + kfo[ks] + kb * 4096);
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) sa[qb][kb] = T::mfma32(kfr, qf[qb][ks], sa[qb][kb]);
        }
    }

    int tile = 0, s = 0;
    auto step = [&](f32x16(&cur)[QB][2], f32x16(&nxt)[QB][2]) __attribute__((always_inline)) {
        if (tile == 0) body(cur, nxt, std::true_type{});
        else body(cur, nxt, std::false_type{});
        if (++tile == ntiles) { fold(s); tile = 0; ++s; }
    };
    int i = 0;
    for (; i + 1 < nsteps; i += 2) { step(sa, sb); step(sb, sa); }
    if (i < nsteps) step(sa, sb);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    if (__syncthreads_or(bad)) {      // some row left the exponent range of its first-tile offset: safe recomputation
        attn_safe_body<T, D, 2 * QB, NW>(a, qblk, h, b, smem, smem + SafeLds<D>::KBYTES);
        return;
    }
    // ---- store: lane owns O[q = qi][d = 32 db + 8 (r / 4) + 4 hg .. +4]
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int q = q_wave0 + 32 * qb + qi;
        if (q >= a.Lq) continue;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int d = 32 * db + 8 * rq + 4 * hg;
                if (32 * db + 8 * rq + 8 > D) continue;      // D % 8 == 0: both halves of the 8-channel group are data, or neither
                *reinterpret_cast<uint2 *>(a.O + (int64_t)b * a.o_bs + (int64_t)q * a.ldo + h * D + d) =
                    make_uint2(pack2<T>(otot[qb][db][4 * rq], otot[qb][db][4 * rq + 1]), pack2<T>(otot[qb][db][4 * rq + 2], otot[qb][db][4 * rq + 3]));
            }
    }
}

template <class T, int D, int NST, int NW, bool RAG, bool PRE, int PF = 3, int QB = 1>
void launch_attn4_(const AttnArgs &a, int B, hipStream_t s)
{
    constexpr int DB = (D + 32) / 32;
    constexpr size_t ring = (size_t)NST * (64 * 128 + DB * 32 * 128), safe = SafeLds<D>::KBYTES + SafeLds<D>::VBYTES;
    constexpr size_t lds = ring > safe ? ring : safe;
    static gc::AttrOnce once;
    gc::ensure_dynamic_lds(once, (const void *)k_attn4<T, D, NST, NW, RAG, PRE, PF, QB>, (int)lds);
    AttnArgs aa = a;
    aa.nqb = (a.Lq + 32 * QB * NW - 1) / (32 * QB * NW);
    dim3 grid((unsigned)(aa.nqb * a.H * B));
    hipLaunchKernelGGL((k_attn4<T, D, NST, NW, RAG, PRE, PF, QB>), grid, dim3(NW * 64), lds, s, aa);
}
template <class T, int D, int NST, int NW, int PF = 3, int QB = 1>
void launch_attn4(const AttnArgs &a, int B, hipStream_t s)
{
    const bool pre = a.scale_log2e == 1.f;
    if (a.Lk & 63) { if (pre) launch_attn4_<T, D, NST, NW, true, true, PF, QB>(a, B, s); else launch_attn4_<T, D, NST, NW, true, false, PF, QB>(a, B, s); }
    else { if (pre) launch_attn4_<T, D, NST, NW, false, true, PF, QB>(a, B, s); else launch_attn4_<T, D, NST, NW, false, false, PF, QB>(a, B, s); }
}

template <class T, int D, int QT, int NST, bool RAG, bool PRE>
void launch_attn3_(const AttnArgs &a, int B, hipStream_t s)
{
    constexpr int DP = (D + 31) / 32 * 32, CPR = DP / 8, DV = (D + 16) / 16 * 16;
    constexpr size_t ring = (size_t)NST * (64 * CPR * 16 + DV * 128), safe = SafeLds<D>::KBYTES + SafeLds<D>::VBYTES;
    constexpr size_t lds = ring > safe ? ring : safe;
    static gc::AttrOnce once;
    gc::ensure_dynamic_lds(once, (const void *)k_attn3<T, D, QT, NST, RAG, PRE>, (int)lds);
    AttnArgs aa = a;
    aa.nqb = (a.Lq + 64 * QT - 1) / (64 * QT);
    dim3 grid((unsigned)(aa.nqb * a.H * B));
    hipLaunchKernelGGL((k_attn3<T, D, QT, NST, RAG, PRE>), grid, dim3(256), lds, s, aa);
}
template <class T, int D, int QT, int NST>
void launch_attn3(const AttnArgs &a, int B, hipStream_t s)
{
    const bool pre = a.scale_log2e == 1.f;
    if (a.Lk & 63) { if (pre) launch_attn3_<T, D, QT, NST, true, true>(a, B, s); else launch_attn3_<T, D, QT, NST, true, false>(a, B, s); }
    else { if (pre) launch_attn3_<T, D, QT, NST, false, true>(a, B, s); else launch_attn3_<T, D, QT, NST, false, false>(a, B, s); }
}

#ifndef ATT80_QT
#define ATT80_QT 1
#endif
template <class T>
int launch_attn(const AttnArgs &a, int D, int B, bool fast, int variant, hipStream_t s)
{
    if (fast && (int64_t)a.nsets * ((a.Lk + 63) / 64) >= 4) {   // short key streams: the pipeline's fill / LDS set-up does not amortise
        switch (D) {
        case 40:
            // default: k_attn5 (key-split 8-wave form, dn_attn5.hip) when the tile shapes fit; kernel_variant bit 4 keeps k_attn4 (A/B)
            if (!(variant & 30) && (a.Lk & 63) == 0 && (a.Lq & 255) == 0)
                gc_dn_launch_attn5(&a, std::is_same<T, BF16>::value ? DT_BF16 : DT_F16, B, (variant & 32) ? 4 : (variant & 64) ? 8 : 6, s);
            else if (variant & 2) launch_attn3<T, 40, 2, 3>(a, B, s);     // kernel_variant bit 1: the 16x16x32 form (A/B measurements)
            else if (variant & 4) launch_attn4<T, 40, 3, 8>(a, B, s);      // bit 2: 8 waves, one workgroup per CU
            else if (variant & 8) launch_attn4<T, 40, 4, 4, 3, 2>(a, B, s); // bit 3: 64 queries per wave (one wave per SIMD)
            else launch_attn4<T, 40, 3, 4>(a, B, s);
            return GC_OK;
        case 80: launch_attn3<T, 80, ATT80_QT, 3>(a, B, s); return GC_OK;
        default: break;
        }
    }
    // Few workgroups and several K/V sets (D = 160 at 16x16 / 8x8: 192 / 48 workgroups of one wave per SIMD, nobody to hide the
    // S -> max -> exp -> P V dependency chain): one workgroup per (query block, set) + a fixed-order fp32 combine
    if (D == 160 && a.part && a.nsets > 1 && (a.Lq & 255) == 0 && !(variant & 128)) {      // kernel_variant bit 7: the 64-query form (A/B, tests)
        AttnArgs aa = a;
        aa.nqb = a.Lq / 256;
        const unsigned nwg = (unsigned)(aa.nqb * a.H * B);
        hipLaunchKernelGGL((k_attn_wide<T, 160, 2, 8>), dim3(nwg, (unsigned)a.nsets), dim3(512), 0, s, aa);
        const int64_t n4 = (int64_t)B * a.Lq * (a.H * 160) / 4;
        hipLaunchKernelGGL((k_attn_combine<T>), dim3((unsigned)std::min<int64_t>((n4 + 255) / 256, 2048)), dim3(256), 0, s, a.part, a.nsets,
                           n4, n4, a.O, a.ldo, a.o_bs, (int64_t)(a.H * 160) / 4, (int64_t)a.Lq);
        return GC_OK;
    }
#define GC_ATT(DD, QQ)                                                                                  \
    do {                                                                                                \
        AttnArgs aa = a;                                                                                \
        aa.nqb = (a.Lq + 64 * QQ - 1) / (64 * QQ);                                                      \
        const unsigned nwg = (unsigned)(aa.nqb * a.H * B);                                              \
        if (a.part && a.nsets > 1 && nwg < 512) {                                                       \
            hipLaunchKernelGGL((k_attn<T, DD, QQ>), dim3(nwg, (unsigned)a.nsets), dim3(256), 0, s, aa); \
            const int64_t n4 = (int64_t)B * a.Lq * (a.H * DD) / 4;                                      \
            hipLaunchKernelGGL((k_attn_combine<T>), dim3((unsigned)std::min<int64_t>((n4 + 255) / 256, 2048)), dim3(256), 0, s, a.part, a.nsets, \
                               n4, n4, a.O, a.ldo, a.o_bs, (int64_t)(a.H * DD) / 4, (int64_t)a.Lq);     \
        } else                                                                                          \
            hipLaunchKernelGGL((k_attn<T, DD, QQ>), dim3(nwg), dim3(256), 0, s, aa);                    \
    } while (0)
    switch (D) {
    case 8: GC_ATT(8, 2); break;
    case 16: GC_ATT(16, 2); break;
    case 32: GC_ATT(32, 2); break;
    case 40: GC_ATT(40, 2); break;
    case 64: GC_ATT(64, 2); break;
    case 80: GC_ATT(80, 2); break;
    case 160: GC_ATT(160, 1); break;
    default: gc::set_error("gc_dn_attention: unsupported head dim %d", D); return GC_EINVAL;
    }
#undef GC_ATT
    return GC_OK;
}

}  // namespace

extern "C" size_t gc_dn_attention_workspace_bytes(const gc_attn_desc *d)
{
    if (!d || d->nsets <= 1 || d->head_dim != 160) return 0;        // the set-split form serves the head size that has no static-offset kernel
    return sizeof(float) * (size_t)d->nsets * (size_t)d->batch * (size_t)d->Lq * (size_t)d->heads * (size_t)d->head_dim;
}

extern "C" int gc_dn_attention(const gc_attn_desc *d, void *stream)
{
    GC_REQUIRE(d && d->Q && d->K && d->Vt && d->O, "null operand");
    GC_REQUIRE(d->nsets >= 1 && d->nsets <= 5, "1..5 K/V sets");
    GC_REQUIRE(d->ldq % 8 == 0 && d->ldk % 8 == 0 && d->ldvt % 8 == 0 && d->ldo % 4 == 0 && d->head_dim % 8 == 0,
               "leading dimensions must keep 16-byte alignment");
    GC_REQUIRE(d->ldvt >= (d->Lk + 7) / 8 * 8, "Vt rows must hold round_up(Lk, 8) tokens (zero padded)");
    AttnArgs a;
    a.Q = (const unsigned short *)d->Q; a.ldq = d->ldq; a.q_bs = d->q_batch_stride;
    a.K = (const unsigned short *)d->K; a.ldk = d->ldk; a.k_bs = d->k_batch_stride;
    a.Vt = (const unsigned short *)d->Vt; a.ldvt = d->ldvt; a.vt_bs = d->vt_batch_stride;
    a.O = (unsigned short *)d->O; a.ldo = d->ldo; a.o_bs = d->o_batch_stride;
    a.Lq = d->Lq; a.Lk = d->Lk; a.H = d->heads; a.f = d->frames_per_half > 0 ? d->frames_per_half : 1;
    a.nsets = d->nsets;
    for (int i = 0; i < 5; ++i) { a.set_kind[i] = d->set_kind[i]; a.set_w[i] = d->set_weight[i]; }
    a.Kr = d->Kref ? (const unsigned short *)d->Kref : a.K; a.kr_bs = d->Kref ? d->kref_batch_stride : a.k_bs;
    a.Vtr = d->Vtref ? (const unsigned short *)d->Vtref : a.Vt; a.vtr_bs = d->Vtref ? d->vtref_batch_stride : a.vt_bs;
    a.ref_fph = d->Kref ? d->ref_frames_per_half : a.f;
    GC_REQUIRE((d->Kref == nullptr) == (d->Vtref == nullptr), "Kref and Vtref must be given together");
    for (int i = 0; i < d->nsets; ++i) GC_REQUIRE(d->set_kind[i] >= -2 && d->set_kind[i] < a.ref_fph, "bad set_kind");
    a.scale_log2e = d->q_prescaled ? 1.f : d->scale * 1.4426950408889634f;
    a.abl = d->kernel_variant >> 8;
    const size_t ws_need = gc_dn_attention_workspace_bytes(d);          // 0: this shape never takes the set-split form, whatever is passed
    a.part = (d->workspace && ws_need > 0 && d->workspace_bytes >= ws_need) ? (float *)d->workspace : nullptr;
    const bool fast = !(d->kernel_variant & 1);      // kernel_variant bit 0: online-softmax kernel everywhere (tests)
    int rc = d->dtype == DT_BF16 ? launch_attn<BF16>(a, d->head_dim, d->batch, fast, d->kernel_variant, gc::S(stream))
             : d->dtype == DT_F16 ? launch_attn<F16>(a, d->head_dim, d->batch, fast, d->kernel_variant, gc::S(stream)) : GC_EINVAL;
    if (rc != GC_OK) return rc;
    return gc::check_launch("gc_dn_attention");
}
