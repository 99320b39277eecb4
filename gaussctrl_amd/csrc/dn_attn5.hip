// dn_attn5.hip -- k_attn5, the key-split 8-wave form of the D = 40 cross-view attention (its own translation unit: the kernel is
// register-tight and is iterated on separately; shared pieces in dn_attn_common.h).  Reference semantics: dn_attn.hip header
// (/root/reference/gaussctrl/utils.py:25-37,86-117).
#include "dn_attn_common.h"

namespace {
#ifdef ATTN5_DEBUG
__device__ unsigned g_attn5_dbg = 0;
#endif
// ------------------------------------------------------------------------------------------------------------------------
// k_attn5 (round 3; D = 40, Lk % 64 == 0, Lq % 256 == 0): the k_attn4 arithmetic with the work of a 64-key tile split so that every
// LDS fragment feeds TWO MFMAs.  Why: per 64-key tile k_attn4's eight co-resident waves (two workgroups of four) issue 8 x 14
// ds_read_b128 (448 LDS cycles) and DMA two 14 KB tiles (224-448 cycles) for 938 MFMA cycles per SIMD -- the LDS array, not the matrix
// pipe, is what saturates (profiles/r02_attn4_ablation.txt: no ds_read -26 %, no DMA -17 %).  Here ONE workgroup of eight waves per CU
// shares one K / V^T stream for 256 queries, and wave w = (query group w >> 1: 64 queries = two 32-query blocks, key half w & 1: 32 of
// the tile's 64 keys).  Per tile and wave: S'^T = 3 k-steps x 2 query blocks (6 MFMAs, 3 K fragments), O^T += 2 row blocks x 2
// k-steps x 2 query blocks (8 MFMAs, 4 V^T fragments): the same 14 MFMAs per (32 queries x 64 keys) with 7 fragment reads instead of
// 14 and one DMA'd tile instead of two -- 36-48 % LDS occupancy instead of 72-96 %.
// (Tried and removed, round 3: deferring the P V MFMAs by a whole tile -- P and V^T fragments double-buffered, the cross-set sums parked
// in LDS to pay for the registers -- is bit-identical and 1.7 % SLOWER (712 vs 700 us): the cost the ablations attribute to the
// S -> exp -> P -> PV chain is not the latency of late exp units.)
// The two key-half waves of a pair use the SAME offset (both evaluate the first key block of a set's first tile), so their partial
// numerators and denominators simply add: the denominators are exchanged through LDS at the end of each set (w_s / (l_a + l_b)
// scales both partial O^T), the partial weighted sums once at the end of the kernel.
// ABL: instrumented instantiation for timing ablations (results wrong by construction): a.abl bit 0 no v_exp, 1 no exp units at
// all, 2 no s_barrier, 3 no LDS-DMA, 4 no LDS fragment reads
template <class T, bool PRE, int NST = 4, bool ABL = false, bool PV16 = !ABL>
__global__ __launch_bounds__(512, 1) void k_attn5(const AttnArgs a)
{
    constexpr int D = 40, NW = 8, KS = 3, DB = 2, QB = 2;
    constexpr int KBYTES = 64 * 128, VBYTES = DB * 32 * 128;
    constexpr int LCS = D / 8, KSS = LCS / 2, HS = LCS & 1;
    constexpr int NT = NW * 64;
    constexpr int VSH = 8 * D / NW;                                        // V^T chunks per wave per tile (40): one DMA instruction
    constexpr int PD = NST - 1, GRP = 2;                                   // DMA instructions per wave per tile: one K, one V^T
    constexpr float BIG = 30000.f;
    constexpr int XL = NST * (KBYTES + VBYTES);                            // byte offset of the denominator exchange area (2 KB)
    constexpr int XS = XL + 2048, NSAMP = 2;                               // byte offset of the key SAMPLE tiles (8 KB each; sets s and s + 1: ping-pong)
    constexpr bool SAMPLED = std::is_same<T, F16>::value;                  // bf16's 8-bit exponent needs no careful offset: it keeps the first key block
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *sK = smem, *sV = smem + NST * KBYTES;
    float *xl = reinterpret_cast<float *>(smem + XL);
    const unsigned ldsK = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem, ldsV = ldsK + NST * KBYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = wid & 1;                                                // key block of every tile this wave owns
    const int qi = lane & 31, hg = lane >> 5;
    int qblk, h, b;
    block_coords(a, 2, qblk, h, b);
    const int q_wave0 = qblk * 256 + (wid >> 1) * 64;
    const int ntiles = a.Lk >> 6;

    // ---- LDS image, written once: zeros, column D of every key row = 1, the ones row of V^T
    for (int i = tid; i < (XS + (SAMPLED ? NSAMP * KBYTES + 16 : 0)) / 16; i += NT) reinterpret_cast<uint4 *>(smem)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    for (int i = tid; i < NST * 64; i += NT) {
        const int st = i >> 6, row = i & 63;
        *reinterpret_cast<unsigned short *>(sK + st * KBYTES + row * 128 + ((LCS ^ swz4(row)) << 4)) = One<T>::v;
    }
    for (int i = tid; i < NST * 64; i += NT) reinterpret_cast<unsigned short *>(sV + (i >> 6) * VBYTES + D * 128)[i & 63] = One<T>::v;

    // ---- Q fragments (B operand): lane holds Q[q = qi][d = 16 ks + 8 hg .. +8] of both query blocks
    uint4 qf[QB][KS];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int q = q_wave0 + 32 * qb + qi;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int d = ks * 16 + hg * 8;
            qf[qb][ks] = (d + 8 <= D) ? *reinterpret_cast<const uint4 *>(a.Q + (int64_t)b * a.q_bs + (int64_t)q * a.ldq + h * D + d)
                                      : make_uint4(0, 0, 0, 0);
        }
        if (hg == HS) qf[qb][KSS].x = pack2<T>(0.f, -BIG);
    }
    const float c2 = a.scale_log2e;
    // Experiment hook (kernel_variant bits 16..20, default 0): P = exp2(s - m0 - cshift).  In f16 P must stay below 2^16, so a row whose
    // later keys beat the first block's maximum by 16 binades sends its workgroup to the safe body; a positive shift widens that margin
    // but P below 2^-14 is FLUSHED on this path (measured, profiles/r03_attn5_f16_window.txt: relative L2 error 3e-4 -> 5e-3 -> 8e-2 at
    // shift 0 / 4 / 8), so f16 keeps shift 0 and its data-dependent fallback rate; bf16's 8-bit exponent never gets there.
    const float cshift = (float)((a.abl >> 8) & 31) / (PRE ? 1.f : c2);
    const int abl = ABL ? __builtin_amdgcn_readfirstlane(a.abl) : 0;
    float abl_x = -(float)(lane & 7);
    unsigned abl_sink = 0;

    // ---- LDS-DMA plan of this lane (k_attn4's, 8 waves: one K and one V^T instruction per wave and tile)
    int k_off, v_off;
    unsigned long long k_msk, v_msk;
    {
        const int p = wid * 64 + lane, row = p >> 3, lc = (p & 7) ^ swz4(row);
        k_msk = __ballot(lc * 8 < D);
        k_off = (row * (int)a.ldk + lc * 8) * 2;
        // The exponent offset of a K/V set is the row maximum over a SAMPLE of 64 of its keys (round 4; before: its first 32 keys).  Sample
        // row j is key j * (Lk / 64) + ((8 j + (j >> 3)) mod (Lk / 64)): on a 64-wide token map one key in every 8 x 8 block of the map, so a
        // query whose large logits sit anywhere in the image has a sample near them -- in f16 P = exp2(s - offset) must stay below 2^16, and
        // the first keys (the top-left corner of the image) say little about a query at the bottom (profiles/r04_attn5_f16_sample.txt).
    }
    {
        const int p = VSH * wid + lane, row = p >> 3, lc = (p & 7) ^ swz4(row);
        v_msk = __ballot(lane < VSH);
        v_off = (row * (int)a.ldvt + lc * 8) * 2;
    }
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
    // (recomputed where it is used -- twice per set at most -- instead of held in a register through the tile loop)
    auto sample_off = [&]() __attribute__((always_inline)) -> unsigned {
        const int p = wid * 64 + lane, row = p >> 3, lc = (p & 7) ^ swz4(row);
        const int kj = row * ntiles + ((8 * row + (row >> 3)) % ntiles);
        return (unsigned)((kj * (int)a.ldk + lc * 8) * 2);
    };
    auto issue_sample = [&](int s) __attribute__((always_inline)) {       // the sample tile of set s -> slot s & 1
        glds16_s(tab(kb_tab, s), sample_off(), ldsK + XS + (s & 1) * KBYTES + wid * 1024, k_msk);
    };
    struct Cur { const unsigned char *p; int tile, s; unsigned dst; };
    Cur ck, cv;
    ck.p = tab(kb_tab, 0); ck.tile = 0; ck.s = 0; ck.dst = ldsK + wid * 1024;
    cv.p = tab(vb_tab, 0); cv.tile = 0; cv.s = 0; cv.dst = ldsV + wid * (VSH * 16);
    const int64_t kstride = (int64_t)128 * a.ldk;
    // dslot >= 0: the ring slot is a compile-time constant (the tile loop is unrolled NST times when ntiles % NST == 0: every LDS
    // address is then an immediate and the per-tile pointer-wrap SALU disappears -- the kernel is sensitive to issue slots: the
    // instrumented build's extra branches alone cost 30 %)
    auto issue_kv = [&](auto dslot_) __attribute__((always_inline)) {
        constexpr int DS = decltype(dslot_)::value;
        glds16_s(ck.p, (unsigned)k_off, DS >= 0 ? ldsK + wid * 1024 + DS * KBYTES : ck.dst, k_msk);
        if (DS < 0) ck.dst = ck.dst + KBYTES == ldsK + wid * 1024 + NST * KBYTES ? ldsK + wid * 1024 : ck.dst + KBYTES;
        ck.p += kstride;
        if (++ck.tile == ntiles) { ck.tile = 0; ck.s = ck.s + 1 < a.nsets ? ck.s + 1 : ck.s; ck.p = tab(kb_tab, ck.s); }
        glds16_s(cv.p, (unsigned)v_off, DS >= 0 ? ldsV + wid * (VSH * 16) + DS * VBYTES : cv.dst, v_msk);
        if (DS < 0) cv.dst = cv.dst + VBYTES == ldsV + wid * (VSH * 16) + NST * VBYTES ? ldsV + wid * (VSH * 16) : cv.dst + VBYTES;
        cv.p += 128;
        if (++cv.tile == ntiles) { cv.tile = 0; cv.s = cv.s + 1 < a.nsets ? cv.s + 1 : cv.s; cv.p = tab(vb_tab, cv.s); }
    };

    // fragment read offsets.  K: MFMA row qi is key pi(qi) of the wave's key block; chunk 2 ks + hg.  V^T: row 32 db + qi, chunk 2 t + hg
    // with t = 2 kh + t' the two 16-key k-steps of the wave's key block
    int kfo[KS], k0o[KS], vfo[2];
    {
        const int row = (qi & ~12) | (((qi >> 2) & 1) << 3) | (((qi >> 3) & 1) << 2);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            k0o[ks] = row * 128 + (((2 * ks + hg) ^ swz4(row)) << 4);
            kfo[ks] = k0o[ks] + kh * 4096;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) vfo[t] = qi * 128 + (((2 * (2 * kh + t) + hg) ^ swz4(qi)) << 4);
    }
    // PV16 (round 4): O^T rows 32..47 (channels 32..39 and the ones row; 48..63 are padding) come from 16x16x32 MFMAs -- two per query block
    // and 32 keys (16 passes) instead of two 32x32x16 (32 passes): -14 % MFMA passes per tile.  Their A operand is V^T rows 32 + (lane & 15) with
    // the 8 keys of chunk c(g) = {0, 2, 1, 3}[g = lane >> 4] of the wave's key block (ONE ds_read_b128 per tile); their B operand is made from
    // the two P fragments of the 32x32x16 form by v_permlane16_swap: swap(P(t=0), P(t=1)) = (queries 0..15, queries 16..31) x the four
    // 8-key chunks in exactly that order.
    int vfo16;
    {
        const int r16 = lane & 15, g = lane >> 4, c = ((g & 1) << 1) | (g >> 1);
        vfo16 = (32 + r16) * 128 + (((4 * kh + c) ^ swz4(r16)) << 4);
    }

    // O^T rows 40..63 are never stored: of row block 1 only rows 32..39 (registers 0..3) are carried across K/V sets
    f32x16 os[QB][PV16 ? 1 : DB], otot0[QB];
    f32x4 otot1[QB][PV16 ? 2 : 1];                   // PV16: per 16-query half (16x16 accumulator layout: lane (query & 15, rows 4 (lane >> 4) + r))
    f32x4 o1[QB][2];                                 // PV16: O^T rows 32..47 of the current set, per 16-query half
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            otot0[qb][r] = 0.f; os[qb][0][r] = 0.f;
            if constexpr (!PV16) os[qb][DB - 1][r] = 0.f;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            otot1[qb][0][r] = 0.f; o1[qb][0][r] = 0.f; o1[qb][1][r] = 0.f;
            if constexpr (PV16) otot1[qb][1][r] = 0.f;
        }
    }
    int bad = 0, bail = 0;
    volatile unsigned *ovf = reinterpret_cast<volatile unsigned *>(smem + XS + NSAMP * KBYTES);      // f16: "a partial denominator overflowed" flag of the workgroup
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;

    const unsigned char *rk = sK, *rv = sV;
    const unsigned char *samp = smem + XS;           // sample tile of the current set
    f32x16 S0, S1;                       // S'^T of the wave's key block for query block 0 / 1 (single-buffered: see the schedule below)
    uint4 pf[QB][2], kf[KS], vf[2][PV16 ? 1 : DB], vf16 = make_uint4(0, 0, 0, 0);
    // exp unit w of k-step t: registers 8 t + 2 w, + 1 of S -> one packed word of the P fragment (2 v_exp_f32 + 1 v_cvt_pk)
    auto unit = [&](const f32x16 &S, uint4 &p, int w, int t) __attribute__((always_inline)) {
        const int r0 = 8 * t + 2 * w;
        float x0 = PRE ? S[r0] : S[r0] * c2, x1 = PRE ? S[r0 + 1] : S[r0 + 1] * c2;
        if (ABL && (abl & 2)) return;
        if (ABL && (abl & 32)) { x0 = abl_x; x1 = abl_x; }          // inputs that do not come from an MFMA
        const unsigned v = (ABL && (abl & 1)) ? pack2<T>(x0, x1) : pack2<T>(__builtin_amdgcn_exp2f(x0), __builtin_amdgcn_exp2f(x1));
        if (ABL && (abl & 64)) { abl_sink ^= v; return; }           // outputs that no MFMA reads
        if (w == 0) p.x = v;
        else if (w == 1) p.y = v;
        else if (w == 2) p.z = v;
        else p.w = v;
    };
    // The same work at OP granularity (round 6): op k of an S tile = (unit u = k / 3: registers 2 u, 2 u + 1; k % 3 = 0 / 1: v_exp_f32 of one register IN PLACE,
    // 2: v_cvt_pk of the pair into word u & 3 of P fragment u >> 2).  ops [a, b) of the 24: lets the tile schedule put ~4 VALU under every MFMA.
    auto uops = [&](f32x16 &S, uint4 &p0, uint4 &p1, auto a_, auto b_) __attribute__((always_inline)) {
        static_for<decltype(a_)::value, decltype(b_)::value>([&](auto k_) __attribute__((always_inline)) {
            constexpr int k = decltype(k_)::value, u = k / 3, sub = k % 3, r0 = 2 * u;
            if constexpr (sub == 0) S[r0] = __builtin_amdgcn_exp2f(PRE ? S[r0] : S[r0] * c2);
            else if constexpr (sub == 1) S[r0 + 1] = __builtin_amdgcn_exp2f(PRE ? S[r0 + 1] : S[r0 + 1] * c2);
            else {
                const unsigned v = pack2<T>(S[r0], S[r0 + 1]);
                uint4 &p = u < 4 ? p0 : p1;
                if constexpr ((u & 3) == 0) p.x = v;
                else if constexpr ((u & 3) == 1) p.y = v;
                else if constexpr ((u & 3) == 2) p.z = v;
                else p.w = v;
            }
        });
    };
    // PV16: the two 16x16x32 B operands (queries 0..15 / 16..31 x 32 keys) from the 32x32x16 P fragments of the two 16-key k-steps
    auto p16 = [&](const uint4 &p0, const uint4 &p1, uint4 &qa, uint4 &qb_) __attribute__((always_inline)) {
        const auto x = __builtin_amdgcn_permlane16_swap(p0.x, p1.x, false, false), y = __builtin_amdgcn_permlane16_swap(p0.y, p1.y, false, false);
        const auto z = __builtin_amdgcn_permlane16_swap(p0.z, p1.z, false, false), w = __builtin_amdgcn_permlane16_swap(p0.w, p1.w, false, false);
        qa = make_uint4(x[0], y[0], z[0], w[0]); qb_ = make_uint4(x[1], y[1], z[1], w[1]);
    };
    // Software pipeline of one tile, in issue order (MFMA groups and the exp units that run in their shadow):
    //   A  wait + barrier (tiles i and i+1 landed, tile i-1's slot free), DMA of tile i+PD
    //   C  S0(i)   = K Q0^T            3 MFMAs   + units 4..7 of S1(i-1)        (K fragments were read one tile ahead)
    //   D  O1 += V^T(i-1) P1(i-1)^T    4 MFMAs   + units 0..2 of S0(i)          (the first MFMA bare: S0 is still in the pipe)
    //   E  read the V^T fragments of tile i (after D: they replace tile i-1's)
    //   F  S1(i)   = K Q1^T            3 MFMAs   + units 3..7 of S0(i);  then read the K fragments of tile i+1
    //   G  O0 += V^T(i) P0(i)^T        4 MFMAs   + units 0..3 of S1(i)          (the first MFMA bare)
    // Every exp batch has a 7-MFMA window and every MFMA carries ~1.2 units (2.4 v_exp + 1.2 v_cvt_pk: inside what a 32x32x16 MFMA
    // hides, profiles/r02_issue_model_32x32.txt); S0 / S1 / P0 / P1 / the fragments need no second copy.
    // (one loop body for every tile -- a peeled first-tile variant makes the register allocator shuttle all four accumulators
    // between two homes, 96 v_mov per tile; at the start of a set the pipeline is primed with P1 = 0 instead, so C and D add nothing)
    auto rd_kf = [&](const unsigned char *kb_) __attribute__((always_inline)) {
        if (ABL && (abl & 16)) return;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kf[ks] = *reinterpret_cast<const uint4 *>(kb_ + kfo[ks]);
    };
    auto tile_step = [&](bool first, auto slot_) __attribute__((always_inline)) {
        constexpr int SLOT = decltype(slot_)::value;                       // ring slot of this tile, or -1: dynamic ring pointers
        wait_vmcnt<(PD - 2) * GRP>();
        if (!(ABL && (abl & 4))) __builtin_amdgcn_s_barrier();
        if (!(ABL && (abl & 8))) issue_kv(std::integral_constant<int, SLOT < 0 ? -1 : (SLOT + PD) % NST>{});
        const unsigned char *kb_ = SLOT < 0 ? rk : sK + SLOT * KBYTES, *vb_ = SLOT < 0 ? rv : sV + SLOT * VBYTES;
        (void)kb_;
        if (SLOT < 0) {
            rk = rk + KBYTES == sK + NST * KBYTES ? sK : rk + KBYTES;
            rv = rv + VBYTES == sV + NST * VBYTES ? sV : rv + VBYTES;
        }
        if (first) {
            // first tile of a K/V set: the row maximum over the set's SAMPLE tile (f16: 64 keys spread over the whole set, DMA'd in the
            // prologue; bf16: the set's first 32 keys) becomes the set's offset -- evaluated by both waves of a pair on the same data, so they agree bit for bit without an exchange
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                if (hg == HS) qf[qb][KSS].x = pack2<T>(0.f, -BIG);
                float t = -BIG;
                const unsigned char *src = SAMPLED ? samp : kb_;
#pragma unroll 1
                for (int hb = 0; hb < (SAMPLED ? 2 : 1); ++hb) {   // f16: the sample tile's two 32-key blocks, one accumulator (rolled: the kernel is register-tight)
                    f32x16 m0 = zero16;
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) m0 = T::mfma32(*reinterpret_cast<const uint4 *>(src + k0o[ks] + hb * 4096), qf[qb][ks], m0);
#pragma unroll
                    for (int r = 0; r + 1 < 16; r += 2) t = fmaxf(fmaxf(t, m0[r]), m0[r + 1]);
                }
                const unsigned x = __float_as_uint(t);
                const auto r1 = __builtin_amdgcn_permlane32_swap(x, x, false, false);
                t = fmaxf(__uint_as_float(r1[0]), __uint_as_float(r1[1]));
                const float mq = T::to_f(T::from_f(t + cshift));
                if (hg == HS) qf[qb][KSS].x = pack2<T>(-mq, -BIG);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PV16) {
            // Issue packing (round 6).  The loop is ISSUE-bound, not dependency-bound (scripts/ubench/attn_pingpong.hip: the same instruction stream with the exp units
            // detached from the MFMAs takes the same time), and a gfx950 SIMD hides ~24 cycles of VALU under a 32x32x16 MFMA and ~8 under a 16x16x32 one, the rest
            // adds to the loop (scripts/ubench/attn_pack.hip, profiles/r06_attn_issue_packing_ubench.txt: the round-3 distribution -- two bare MFMAs, bursts of 6-10 ops
            // behind others -- 1 180 ticks per tile, ~4 ops under every MFMA 1 100).  Ops of S1 (P1, read by D of the NEXT tile): G2 1, g1 3, g2 3 | C1 4, C2 4, C3 5,
            // D1 4; ops of S0 (P0, read by G): d1 3, d2 3, F1 5, F2 5, F3 5, G1 3; the lane swaps stay alone under D2 / G2.  Every P word is complete before the MFMA that
            // reads it issues (P1 words 0..3 by C2, 4..7 by D1 < D2; P0 words 0..3 by F2, 4..7 by G1 < G2); S0 / S1 are exponentiated in place.
            using std::integral_constant;
            uint4 qa, qc;
            S0 = T::mfma32(kf[0], qf[0][0], zero16);                                                                                  // C
            __builtin_amdgcn_sched_barrier(0);          // (the MFMA issues first: its ops run in its shadow)
            uops(S1, pf[1][0], pf[1][1], integral_constant<int, 7>{}, integral_constant<int, 11>{});
            __builtin_amdgcn_sched_barrier(0);
            S0 = T::mfma32(kf[1], qf[0][1], S0);
            __builtin_amdgcn_sched_barrier(0);          // (the MFMA issues first: its ops run in its shadow)
            uops(S1, pf[1][0], pf[1][1], integral_constant<int, 11>{}, integral_constant<int, 15>{});
            __builtin_amdgcn_sched_barrier(0);
            S0 = T::mfma32(kf[2], qf[0][2], S0);
            __builtin_amdgcn_sched_barrier(0);          // (the MFMA issues first: its ops run in its shadow)
            uops(S1, pf[1][0], pf[1][1], integral_constant<int, 15>{}, integral_constant<int, 20>{});
            __builtin_amdgcn_sched_barrier(0);
            os[1][0] = T::mfma32(vf[0][0], pf[1][0], os[1][0]);                                                                     // D
            __builtin_amdgcn_sched_barrier(0);          // (the MFMA issues first: its ops run in its shadow)
            uops(S1, pf[1][0], pf[1][1], integral_constant<int, 20>{}, integral_constant<int, 24>{});
            __builtin_amdgcn_sched_barrier(0);
            os[1][0] = T::mfma32(vf[1][0], pf[1][1], os[1][0]);
            __builtin_amdgcn_sched_barrier(0);          // (the MFMA issues first: its ops run in its shadow)
            p16(pf[1][0], pf[1][1], qa, qc);
            __builtin_amdgcn_sched_barrier(0);
            o1[1][0] = T::mfma(vf16, qa, o1[1][0]);
            __builtin_amdgcn_sched_barrier(0);          // (the MFMA issues first: its ops run in its shadow)
            uops(S0, pf[0][0], pf[0][1], integral_constant<int, 0>{}, integral_constant<int, 3>{});
            __builtin_amdgcn_sched_barrier(0);
            o1[1][1] = T::mfma(vf16, qc, o1[1][1]);
            __builtin_amdgcn_sched_barrier(0);          // (the MFMA issues first: its ops run in its shadow)
            uops(S0, pf[0][0], pf[0][1], integral_constant<int, 3>{}, integral_constant<int, 6>{});
            __builtin_amdgcn_sched_barrier(0);
            vf[0][0] = *reinterpret_cast<const uint4 *>(vb_ + vfo[0]); vf[1][0] = *reinterpret_cast<const uint4 *>(vb_ + vfo[1]);     // E
            vf16 = *reinterpret_cast<const uint4 *>(vb_ + vfo16);
            S1 = T::mfma32(kf[0], qf[1][0], zero16);                                                                                  // F
            __builtin_amdgcn_sched_barrier(0);          // (the MFMA issues first: its ops run in its shadow)
            uops(S0, pf[0][0], pf[0][1], integral_constant<int, 6>{}, integral_constant<int, 11>{});
            __builtin_amdgcn_sched_barrier(0);
            S1 = T::mfma32(kf[1], qf[1][1], S1);
            __builtin_amdgcn_sched_barrier(0);          // (the MFMA issues first: its ops run in its shadow)
            uops(S0, pf[0][0], pf[0][1], integral_constant<int, 11>{}, integral_constant<int, 16>{});
            __builtin_amdgcn_sched_barrier(0);
            S1 = T::mfma32(kf[2], qf[1][2], S1);
            __builtin_amdgcn_sched_barrier(0);          // (the MFMA issues first: its ops run in its shadow)
            uops(S0, pf[0][0], pf[0][1], integral_constant<int, 16>{}, integral_constant<int, 21>{});
            __builtin_amdgcn_sched_barrier(0);
            rd_kf(SLOT < 0 ? rk : sK + ((SLOT + 1) % NST) * KBYTES);      // tile i+1 (landed: the barrier above waited for it)
            os[0][0] = T::mfma32(vf[0][0], pf[0][0], os[0][0]);                                                                     // G
            __builtin_amdgcn_sched_barrier(0);          // (the MFMA issues first: its ops run in its shadow)
            uops(S0, pf[0][0], pf[0][1], integral_constant<int, 21>{}, integral_constant<int, 24>{});
            __builtin_amdgcn_sched_barrier(0);
            os[0][0] = T::mfma32(vf[1][0], pf[0][1], os[0][0]);
            __builtin_amdgcn_sched_barrier(0);          // (the MFMA issues first: its ops run in its shadow)
            p16(pf[0][0], pf[0][1], qa, qc);
            uops(S1, pf[1][0], pf[1][1], integral_constant<int, 0>{}, integral_constant<int, 1>{});
            __builtin_amdgcn_sched_barrier(0);
            o1[0][0] = T::mfma(vf16, qa, o1[0][0]);
            __builtin_amdgcn_sched_barrier(0);          // (the MFMA issues first: its ops run in its shadow)
            uops(S1, pf[1][0], pf[1][1], integral_constant<int, 1>{}, integral_constant<int, 4>{});
            __builtin_amdgcn_sched_barrier(0);
            o1[0][1] = T::mfma(vf16, qc, o1[0][1]);
            __builtin_amdgcn_sched_barrier(0);          // (the MFMA issues first: its ops run in its shadow)
            uops(S1, pf[1][0], pf[1][1], integral_constant<int, 4>{}, integral_constant<int, 7>{});
            __builtin_amdgcn_sched_barrier(0);
            return;
        }
        S0 = T::mfma32(kf[0], qf[0][0], zero16);        // C
        unit(S1, pf[1][1], 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        S0 = T::mfma32(kf[1], qf[0][1], S0);
        unit(S1, pf[1][1], 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        S0 = T::mfma32(kf[2], qf[0][2], S0);
        unit(S1, pf[1][1], 2, 1); unit(S1, pf[1][1], 3, 1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PV16) {
            uint4 qa, qc;
            os[1][0] = T::mfma32(vf[0][0], pf[1][0], os[1][0]);      // D
            __builtin_amdgcn_sched_barrier(0);
            os[1][0] = T::mfma32(vf[1][0], pf[1][1], os[1][0]);
            p16(pf[1][0], pf[1][1], qa, qc);      // after BOTH 32x32x16 MFMAs have read P: the lane swaps run in place (no copies of the 8 P registers)
            unit(S0, pf[0][0], 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            o1[1][0] = T::mfma(vf16, qa, o1[1][0]);
            unit(S0, pf[0][0], 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            o1[1][1] = T::mfma(vf16, qc, o1[1][1]);
            unit(S0, pf[0][0], 2, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (!(ABL && (abl & 16))) {     // E
                vf[0][0] = *reinterpret_cast<const uint4 *>(vb_ + vfo[0]); vf[1][0] = *reinterpret_cast<const uint4 *>(vb_ + vfo[1]);
                vf16 = *reinterpret_cast<const uint4 *>(vb_ + vfo16);
            }
        } else {
        os[1][0] = T::mfma32(vf[0][0], pf[1][0], os[1][0]);      // D
        __builtin_amdgcn_sched_barrier(0);
        os[1][DB - 1] = T::mfma32(vf[0][DB - 1], pf[1][0], os[1][DB - 1]);
        unit(S0, pf[0][0], 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        os[1][0] = T::mfma32(vf[1][0], pf[1][1], os[1][0]);
        unit(S0, pf[0][0], 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        os[1][DB - 1] = T::mfma32(vf[1][DB - 1], pf[1][1], os[1][DB - 1]);
        unit(S0, pf[0][0], 2, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (!(ABL && (abl & 16))) {
#pragma unroll
            for (int t = 0; t < 2; ++t)     // E
#pragma unroll
                for (int db = 0; db < (PV16 ? 1 : DB); ++db) vf[t][db] = *reinterpret_cast<const uint4 *>(vb_ + vfo[t] + db * 4096);
        }
        }
        S1 = T::mfma32(kf[0], qf[1][0], zero16);        // F
        unit(S0, pf[0][0], 3, 0); unit(S0, pf[0][1], 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        S1 = T::mfma32(kf[1], qf[1][1], S1);
        unit(S0, pf[0][1], 1, 1); unit(S0, pf[0][1], 2, 1);
        __builtin_amdgcn_sched_barrier(0);
        S1 = T::mfma32(kf[2], qf[1][2], S1);
        unit(S0, pf[0][1], 3, 1);
        __builtin_amdgcn_sched_barrier(0);
        rd_kf(SLOT < 0 ? rk : sK + ((SLOT + 1) % NST) * KBYTES);      // tile i+1 (landed: the barrier above waited for it)
        if constexpr (PV16) {
            uint4 qa, qc;
            os[0][0] = T::mfma32(vf[0][0], pf[0][0], os[0][0]);      // G
            __builtin_amdgcn_sched_barrier(0);
            os[0][0] = T::mfma32(vf[1][0], pf[0][1], os[0][0]);
            p16(pf[0][0], pf[0][1], qa, qc);
            unit(S1, pf[1][0], 0, 0); unit(S1, pf[1][0], 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            o1[0][0] = T::mfma(vf16, qa, o1[0][0]);
            unit(S1, pf[1][0], 2, 0);
            __builtin_amdgcn_sched_barrier(0);
            o1[0][1] = T::mfma(vf16, qc, o1[0][1]);
            unit(S1, pf[1][0], 3, 0);
            __builtin_amdgcn_sched_barrier(0);
        } else {
        os[0][0] = T::mfma32(vf[0][0], pf[0][0], os[0][0]);      // G
        __builtin_amdgcn_sched_barrier(0);
        os[0][DB - 1] = T::mfma32(vf[0][DB - 1], pf[0][0], os[0][DB - 1]);
        unit(S1, pf[1][0], 0, 0); unit(S1, pf[1][0], 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        os[0][0] = T::mfma32(vf[1][0], pf[0][1], os[0][0]);
        unit(S1, pf[1][0], 2, 0);
        __builtin_amdgcn_sched_barrier(0);
        os[0][DB - 1] = T::mfma32(vf[1][DB - 1], pf[0][1], os[0][DB - 1]);
        unit(S1, pf[1][0], 3, 0);
        __builtin_amdgcn_sched_barrier(0);
        }
    };
    // end of a K/V set: finish the pipeline (B and D of the last tile), then O_total += w / (l_a + l_b) * O_set; a wave's partial
    // denominator is row D of its O^T (the ones row of V^T)
    auto fold = [&](int s) __attribute__((always_inline)) {
        if constexpr (PV16) uops(S1, pf[1][0], pf[1][1], std::integral_constant<int, 7>{}, std::integral_constant<int, 24>{});      // (ops 0..6 ran under the last tile's G)
        else { unit(S1, pf[1][1], 0, 1); unit(S1, pf[1][1], 1, 1); unit(S1, pf[1][1], 2, 1); unit(S1, pf[1][1], 3, 1); }
        constexpr int db_l = D / 32, dl = D % 32, r_l = (dl >> 3) * 4 + (dl & 3), hg_l = (dl >> 2) & 1;
        static_assert(!PV16 || (db_l == 1 && dl == 8), "PV16: the ones row is row 8 of the 16-row block (D = 40)");
        const bool lden = PV16 ? (lane >> 4) == 2 : hg == hg_l;       // lanes that hold partial denominators (PV16: rows 8..11 of the 16x16 tiles)
        if constexpr (PV16) {
            uint4 qa, qc;
            os[1][0] = T::mfma32(vf[0][0], pf[1][0], os[1][0]);
            p16(pf[1][0], pf[1][1], qa, qc);
            os[1][0] = T::mfma32(vf[1][0], pf[1][1], os[1][0]);
            o1[1][0] = T::mfma(vf16, qa, o1[1][0]);
            o1[1][1] = T::mfma(vf16, qc, o1[1][1]);
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
                if (lden) { xl[(wid * QB + qb) * 32 + (lane & 15)] = o1[qb][0][0]; xl[(wid * QB + qb) * 32 + 16 + (lane & 15)] = o1[qb][1][0]; }
        } else {
            os[1][0] = T::mfma32(vf[0][0], pf[1][0], os[1][0]);
            os[1][DB - 1] = T::mfma32(vf[0][DB - 1], pf[1][0], os[1][DB - 1]);
            os[1][0] = T::mfma32(vf[1][0], pf[1][1], os[1][0]);
            os[1][DB - 1] = T::mfma32(vf[1][DB - 1], pf[1][1], os[1][DB - 1]);
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
                if (lden) xl[(wid * QB + qb) * 32 + qi] = os[qb][DB - 1][r_l];
        }
        if (SAMPLED && !ABL) {          // f16: an overflowed partial denominator (inf / NaN) is known before the exchange -- flag it through the same barrier
            bool ov = false;
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                if constexpr (PV16) ov |= lden && !(o1[qb][0][0] < 1e37f && o1[qb][1][0] < 1e37f);
                else ov |= lden && !(os[qb][DB - 1][r_l] < 1e37f);
            }
            if (__ballot(ov) != 0ull && lane == 0) *ovf = 1u;
        }
        __syncthreads();
        if (SAMPLED && !ABL) bail = (int)*ovf;
        // the sample tile of set s + 2 replaces set s's (read at the set's first tile, long ago); it has a whole set to land.  One more load
        // in the in-order queue only makes the counted waits of the next tiles wait for older loads, never for fewer.
        if (SAMPLED && s + 2 < a.nsets) issue_sample(s + 2);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const float l = xl[(wid * QB + qb) * 32 + qi] + xl[((wid ^ 1) * QB + qb) * 32 + qi];
            bad |= !(l > 0.f && l < 1e37f);
#ifdef ATTN5_DEBUG
            if (!(l > 0.f && l < 1e37f) && hg == 0 && atomicAdd(&g_attn5_dbg, 1u) < 24u)
                printf("bad: wg %d wave %d qb %d q %d set %d: l_mine %g l_partner %g\n", (int)blockIdx.x, wid, qb, qi, s, xl[(wid * QB + qb) * 32 + qi], xl[((wid ^ 1) * QB + qb) * 32 + qi]);
#endif
            const float inv = a.set_w[s] / l;
#pragma unroll
            for (int r = 0; r < 16; ++r) { otot0[qb][r] += os[qb][0][r] * inv; os[qb][0][r] = 0.f; }
            if constexpr (PV16) {
                // the 16x16 tiles: lane (query & 15) of half hh holds query 16 hh + (lane & 15) -- this lane's own query qi when hh == (qi >> 4), else qi ^ 16
                const float l2 = xl[(wid * QB + qb) * 32 + (qi ^ 16)] + xl[((wid ^ 1) * QB + qb) * 32 + (qi ^ 16)];
                bad |= !(l2 > 0.f && l2 < 1e37f);
                const float inv2 = a.set_w[s] / l2;
                const bool up = (qi >> 4) & 1;
                const float i0 = up ? inv2 : inv, i1 = up ? inv : inv2;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    otot1[qb][0][r] += o1[qb][0][r] * i0; otot1[qb][1][r] += o1[qb][1][r] * i1;
                    o1[qb][0][r] = 0.f; o1[qb][1][r] = 0.f;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) otot1[qb][0][r] += os[qb][DB - 1][r] * inv;
#pragma unroll
                for (int r = 0; r < 16; ++r) os[qb][DB - 1][r] = 0.f;
            }
        }
    };

    // ---- prologue: the sample tiles of every set (they land first: loads complete in order), then PD tiles in flight
    __syncthreads();
    if (SAMPLED) {
        issue_sample(0);
        if (a.nsets > 1) issue_sample(1);
    }
    static_for<0, PD>([&](auto j_) __attribute__((always_inline)) { issue_kv(std::integral_constant<int, decltype(j_)::value>{}); });
    ck.dst = ldsK + wid * 1024 + PD * KBYTES; cv.dst = ldsV + wid * (VSH * 16) + PD * VBYTES;      // (dynamic form: next slot)
    wait_vmcnt<(PD - 1) * GRP>();
    __builtin_amdgcn_s_barrier();
    rd_kf(rk);                          // K fragments run one tile ahead of the loop
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int db = 0; db < (PV16 ? 1 : DB); ++db) vf[t][db] = make_uint4(0, 0, 0, 0);
    for (int s = 0; s < a.nsets; ++s) {
        samp = smem + XS + (s & 1) * KBYTES;
#pragma unroll
        for (int r = 0; r < 16; ++r) S1[r] = (PV16 && r < 5) ? 0.f : -BIG;      // exp2 -> 0: the pipeline starts with P1 = 0 (PV16: ops 0..6 of the op schedule
                                                                                // "already ran": registers 0..4 hold exponentials, words 0, 1 of P1 are packed)
        pf[1][0] = make_uint4(0, 0, 0, 0);
        if (ABL && (abl & 128)) {       // P fragments hold non-trivial constants instead of zeros (with bit 6: is the cost the dependency or the data?)
            const uint4 c = make_uint4(0x3F2A3E91u + lane, 0x3DD73F11u ^ (lane << 3), 0x3E4C3F60u, 0x3F053D9Au + 7 * lane);
            pf[0][0] = c; pf[0][1] = c; pf[1][0] = c; pf[1][1] = c;
        }
        if (ntiles % NST == 0) {        // every set starts at ring slot 0: unrolled, static slots
            for (int t = 0; t < ntiles; t += NST)
                static_for<0, NST>([&](auto k_) __attribute__((always_inline)) { tile_step(t + decltype(k_)::value == 0, k_); });
        } else {
            for (int t = 0; t < ntiles; ++t) tile_step(t == 0, std::integral_constant<int, -1>{});
        }
        fold(s);
        if (SAMPLED && !ABL && bail) break;      // f16: leave for the safe body as soon as a set has overflowed instead of finishing the other sets first
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    if (ABL && abl_sink == 0x12345u) bad = 1;
    if (bail || __syncthreads_or(ABL ? (bad & 2) : bad)) {      // some row left the exponent range of its set's offset: safe recomputation
        __syncthreads();                                        // (every wave is past its last LDS read; the DMA was drained above)
        attn_safe_body<T, D, 2, NW>(a, qblk, h, b, smem, smem + SafeLds<D>::KBYTES);
        return;
    }
    // ---- combine the two key halves: wave kh hands its partial of query block 1 - kh to its partner and stores block kh
    constexpr int XR = PV16 ? 24 : 20;               // exchanged registers per lane
    float *xo = reinterpret_cast<float *>(smem) + (size_t)wid * (XR * 64);
#pragma unroll
    for (int r = 0; r < 16; ++r) xo[r * 64 + lane] = kh ? otot0[0][r] : otot0[1][r];
#pragma unroll
    for (int hh = 0; hh < (PV16 ? 2 : 1); ++hh)
#pragma unroll
        for (int r = 0; r < 4; ++r) xo[(16 + 4 * hh + r) * 64 + lane] = kh ? otot1[0][hh][r] : otot1[1][hh][r];
    __syncthreads();
    const float *xp = reinterpret_cast<const float *>(smem) + (size_t)(wid ^ 1) * (XR * 64);
    const int q = q_wave0 + 32 * kh + qi;
    unsigned short *orow = a.O + (int64_t)b * a.o_bs + (int64_t)q * a.ldo + h * D;
    {
        float o[XR];
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = (kh ? otot0[1][r] : otot0[0][r]) + xp[r * 64 + lane];
#pragma unroll
        for (int hh = 0; hh < (PV16 ? 2 : 1); ++hh)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[16 + 4 * hh + r] = (kh ? otot1[1][hh][r] : otot1[0][hh][r]) + xp[(16 + 4 * hh + r) * 64 + lane];
#pragma unroll
        for (int rq = 0; rq < (PV16 ? 4 : 5); ++rq)      // channels 8 rq + 4 hg .. + 4 (not PV16: rq = 4 = rows 32..39 of row block 1)
            *reinterpret_cast<uint2 *>(orow + 8 * rq + 4 * hg) =
                make_uint2(pack2<T>(o[4 * rq], o[4 * rq + 1]), pack2<T>(o[4 * rq + 2], o[4 * rq + 3]));
        if constexpr (PV16) {
            // channels 32..39 from the 16x16 tiles: lanes 0..31 hold rows 4 (lane >> 4) + r of query 16 hh + (lane & 15) of the block
            if (hg == 0) {
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int q16 = q_wave0 + 32 * kh + 16 * hh + (lane & 15);
                    unsigned short *orow16 = a.O + (int64_t)b * a.o_bs + (int64_t)q16 * a.ldo + h * D;
                    *reinterpret_cast<uint2 *>(orow16 + 32 + 4 * (lane >> 4)) =
                        make_uint2(pack2<T>(o[16 + 4 * hh], o[16 + 4 * hh + 1]), pack2<T>(o[16 + 4 * hh + 2], o[16 + 4 * hh + 3]));
                }
            }
        }
    }
}

template <class T, bool PRE, int NST, bool ABL = false>
void launch_attn5_(const AttnArgs &a, int B, hipStream_t s)
{
    constexpr size_t ring = (size_t)NST * (64 * 128 + 2 * 32 * 128) + 2048 + 2 * (64 * 128) + 16, xchg = (size_t)8 * 24 * 64 * 4;   // ring + exchange + 2 sample tiles + flag
    constexpr size_t safe = SafeLds<40>::KBYTES + SafeLds<40>::VBYTES;
    constexpr size_t lds = ring > xchg ? (ring > safe ? ring : safe) : (xchg > safe ? xchg : safe);
    static gc::AttrOnce once;
    gc::ensure_dynamic_lds(once, (const void *)k_attn5<T, PRE, NST, ABL>, (int)lds);
    AttnArgs aa = a;
    aa.nqb = a.Lq / 256;
    dim3 grid((unsigned)(aa.nqb * a.H * B));
    hipLaunchKernelGGL((k_attn5<T, PRE, NST, ABL>), grid, dim3(512), lds, s, aa);
}
template <class T, int NST>
void launch_attn5(const AttnArgs &a, int B, hipStream_t s)
{
    if (a.scale_log2e == 1.f) launch_attn5_<T, true, NST>(a, B, s);
    else launch_attn5_<T, false, NST>(a, B, s);
}
}  // namespace

// AttnArgs has internal linkage per translation unit (same definition in both): the entry point takes it through a void pointer.
// depth: stages of the K / V^T LDS ring (4, 6 or 8: the DMA of a tile is issued depth - 1 tiles before its barrier)
void gc_dn_launch_attn5(const void *args, int dtype, int B, int depth, hipStream_t s)
{
    (void)depth;       // ring depths 4 / 6 / 8 measured equal (profiles/r03_attn5_ablation.txt): 4 stages, which divides the 64 tiles of L = 4096
    const AttnArgs &a = *static_cast<const AttnArgs *>(args);
    if (a.abl & 0xff) { launch_attn5_<BF16, true, 4, true>(a, B, s); return; }
    if (dtype == DT_BF16) launch_attn5<BF16, 4>(a, B, s);
    else launch_attn5<F16, 4>(a, B, s);
}
