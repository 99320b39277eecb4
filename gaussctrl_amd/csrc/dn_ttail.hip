// dn_ttail.hip -- the "tail" of a level-0 SD1.5 transformer block (C = 320, 8 heads of 40) as ONE kernel: everything that follows
// the cross-view self-attention until the block's output,
//     h1 = attn1.to_out(o1) + h            h2 = attn2.to_out(softmax(q2 Kt^T) Vt) + h1,  q2 = attn2.to_q(LN2(h1))
//     h3 = ff.net.2(GEGLU(ff.net.0.proj(LN3(h2)))) + h2          out = proj_out(h3) + x
// (diffusers BasicTransformerBlock.forward + Transformer2DModel's proj_out, reached from the reference at gaussctrl/gc_pipeline.py:224-227
// through the UNet it patches in gc_pipeline.py:76-83; nine launches of the per-op path: DESIGN.md 7.0).
//
// ROWS STAY IN REGISTERS.  A wave owns 32 token rows for the whole kernel; every GEMM runs "transposed" on v_mfma_f32_32x32x16:
// D[n][m] += W[n][k] x[m][k], A operand = a 32 x 16 weight tile read from LDS, B operand = the rows' activations held by the lane.
// The 32x32 accumulator leaves lane (m, hg) with channels n = 32 nb + 8 g + 4 hg + c in register 4 g + c -- and a B operand wants 8
// k-values per lane: with the k-order of every weight tile permuted on the host (PERM16 below) the registers 8 j .. 8 j + 7 of
// accumulator block nb ARE the B fragment of k-step 2 nb + j of the next GEMM.  No activation ever goes through LDS or HBM between
// the nine operations; LayerNorm / softmax / GEGLU are lane-local plus one exchange with the partner lane (m, 1 - hg).
//
// WEIGHTS ARE ONE LINEAR STREAM.  Each MFMA of a wave consumes one 1 KB block (lane l's 16 bytes at offset 16 l); the host lays all
// blocks out in exactly the order the kernel consumes them (weights, then the text K / V^T of the row's CFG half, then weights), so
// operand delivery is a ring of 8 KB slots filled by global_load_lds_dwordx4 (2 instructions per wave per slot), one s_barrier per slot,
// and conflict-free ds_read_b128.  A workgroup is 4 waves (one per SIMD, up to 512 registers each) = 128 rows; all four consume the
// same stream.  What bounds it: profiles/r03_weight_stream_ubench.txt (delivery) and the 3368 MFMAs per wave (45 us at 2.4 GHz).
#include "dn_attn_common.h"

namespace {

constexpr int TC = 320, TH = 8, TD = 40;                 // channels, heads, head size
constexpr int NB = TC / 32, KS = TC / 16;                // accumulator blocks / k-steps of a C-wide GEMM
constexpr int FF = 4 * TC, FF_IT = FF / 64;              // GEGLU inner width; 64 inner channels (= 4 up-blocks = 4 down k-steps) per iteration
// GELU by table: (gelu(x_i), gelu(x_i+1) - gelu(x_i)) at x_i = (i - GELU_N / 2) / GELU_STEP, exact erf on the host, linear interpolation
// (|error| <= h^2 / 8 max|gelu''| = 8.6e-6; beyond +-8 the end segments extrapolate gelu's asymptotes x and 0): 7 VALU + one ds_read_b64
// per value instead of the 14 of erf by Abramowitz-Stegun -- at one wave per SIMD every VALU instruction is matrix-core idle time
constexpr int GELU_N = 2048;
constexpr float GELU_STEP = 128.f;
constexpr int SLOT_BLK = 8, SLOT = SLOT_BLK * 1024, NSLOT = 15, RING_BLK = NSLOT * SLOT_BLK;
constexpr int BLK_A = 2 * KS * NB, BLK_KV = TH * 21, BLK_B = KS * NB + FF_IT * (KS * 4 + 4 * NB) + KS * NB;
constexpr int NSLOTS_TOTAL = (BLK_A + BLK_KV + BLK_B) / SLOT_BLK;
constexpr int G_O1 = 0, G_Q2 = KS * NB, G_KV = BLK_A, G_O2 = BLK_A + BLK_KV, G_FF = G_O2 + KS * NB, FF_BLK = KS * 4 + 4 * NB, G_PO = G_FF + FF_IT * FF_BLK;
static_assert(FF_BLK == RING_BLK, "one feed-forward iteration = one revolution of the ring");
static_assert(BLK_A % SLOT_BLK == 0 && BLK_KV % SLOT_BLK == 0 && BLK_B % SLOT_BLK == 0, "segments are whole slots");
// parameter table (floats, "lane order": index 32 nb + 16 hg + r <-> channel 32 nb + 8 (r >> 2) + 4 hg + (r & 3))
constexpr int P_BO1 = 0, P_G2 = 320, P_B2 = 640, P_BO2 = 960, P_G3 = 1280, P_B3 = 1600, P_BDN = 1920, P_BPO = 2240, P_BUP = 2560, P_LUT = 2560 + 2 * FF, P_TOTAL = P_LUT + 2 * GELU_N;
constexpr int LDS_BYTES = NSLOT * SLOT + P_TOTAL * 4;

struct TailArgs {
    const unsigned short *o1, *h, *x;      // [M][320]: attn1 output, residual stream (proj_in output), block input (proj_out residual)
    unsigned short *out;                   // [M][320]
    const unsigned char *wa, *wkv, *wb;    // stream segments; wkv: [2 halves][BLK_KV KB]
    const float *params;                   // [P_TOTAL]
    int M, rows_per_frame, f, Lt, h_frags;
    int in_rows;                           // > 0: the three inputs hold rows [0, in_rows) only; output row m reads input row m % in_rows (CFG-shared prefix)
    float eps;
    int stop;                              // tests: 0 = whole tail; 1..5 = write the intermediate after that many stages to `out` and leave
};

#ifdef TTAIL_ABLATIONS
__device__ unsigned long long g_stamps[64];
#define STAMP(i) do { if (blockIdx.x == 37 && tid == 0) g_stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

// development builds (-DTTAIL_ABLATIONS) only: ABL 1 no DMA, 2 no barriers, 4 no LDS reads, 8 no MFMA; STOP 1..5: `out` receives the
// intermediate after that stage (compile-time: a run-time early exit costs spills around every branch)
template <class T, int ABL = 0, int STOP = 0>
__global__ __launch_bounds__(256, 1) void k_ttail(const TailArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *prm = reinterpret_cast<float *>(smem + NSLOT * SLOT);
    const int m = lane & 31, hg = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * 128;
    const int half = (int)(row0 / a.rows_per_frame) / a.f;
    const unsigned char *wkv = a.wkv + (size_t)half * BLK_KV * 1024;

    // ---------------- the stream.  Block g of the stream lives at ring position g % RING_BLK.  A wave keeps the next 8 blocks (one slot) in
    // registers (`pre`): blk(base, i) hands out block base + i and immediately requests block base + i + 8 into the same registers, so
    // the LDS latency never meets an MFMA (one wave per SIMD: nobody else would hide it).  When that look-ahead enters slot s: the wave's
    // own DMA for s has landed (vmcnt), everyone's has (s_barrier) -- and everyone has CONSUMED slot s - 2, whose ring position is
    // refilled.  Every LDS offset is a compile-time constant: the stages are unrolled and one feed-forward iteration (120 blocks) is
    // exactly one revolution of the ring.
    int issue_slot = 0;                    // next slot to request (uniform)
    const unsigned char *isrc = a.wa + (size_t)wid * 2048;      // this wave's 2 KB of it
    const unsigned char *isrc_kv = wkv + (size_t)wid * 2048, *isrc_b = a.wb + (size_t)wid * 2048;
    const unsigned voff = lane * 16;
    auto issue = [&](int ring_slot, bool mid = false) __attribute__((always_inline)) {      // mid: inside segment B for sure (the feed-forward loop)
        const unsigned dst = lds0 + ring_slot * SLOT + 2 * wid * 1024;
        if (!(ABL & 1)) {          // EXEC is all ones here: no mask juggling around the DMA (dn_attn_common.h's glds16_s writes EXEC twice per call)
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(isrc), "s"(dst) : "memory");
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(isrc + 1024), "s"(dst + 1024) : "memory");
        }
        ++issue_slot;
        isrc += SLOT;
        if (!mid) {
            if (issue_slot == BLK_A / SLOT_BLK) isrc = isrc_kv;
            if (issue_slot == (BLK_A + BLK_KV) / SLOT_BLK) isrc = isrc_b;
        }
    };
    uint4 pre[SLOT_BLK];
    const unsigned char *my = smem + lane * 16, *my_hi = my + 65536;       // two bases: every block offset fits the 16-bit immediate
    auto fetch = [&](int g, bool mid = false) __attribute__((always_inline)) {        // g: stream index of the block to read (compile-time after unrolling, up to a ring revolution)
        if ((g & (SLOT_BLK - 1)) == 0) {   // entering slot g / 8; mid: far from the end of the stream (the feed-forward loop)
            if (mid || issue_slot < NSLOTS_TOTAL) {
                wait_vmcnt<2 * (NSLOT - 3)>();
                if (!(ABL & 2)) __builtin_amdgcn_s_barrier();
                issue((g / SLOT_BLK + NSLOT - 2) % NSLOT, mid);
            } else {
                wait_vmcnt<0>();
                if (!(ABL & 2)) __builtin_amdgcn_s_barrier();
            }
        }
        const int o = (g % RING_BLK) * 1024;
        if (!(ABL & 4) || g < SLOT_BLK) pre[g & (SLOT_BLK - 1)] = *reinterpret_cast<const uint4 *>(o < 65536 ? my + o : my_hi + (o - 65536));
    };
    auto blk = [&](int base, int i, bool mid = false) __attribute__((always_inline)) -> uint4 {       // base: the stage's first block (a multiple of 8), i: block inside the stage
        const uint4 v = pre[i & (SLOT_BLK - 1)];
        if (base + i + SLOT_BLK < NSLOTS_TOTAL * SLOT_BLK) fetch(base + i + SLOT_BLK, mid);
        __builtin_amdgcn_sched_barrier(0);           // keep the look-ahead where it is: hipcc would sink every read next to its MFMA
        return v;
    };
#pragma unroll 1
    for (int i = 0; i < NSLOT - 2; ++i) issue(i);
#pragma unroll
    for (int i = 0; i < SLOT_BLK; ++i) fetch(i);

    for (int i = tid; i < P_TOTAL / 4; i += 256) reinterpret_cast<float4 *>(prm)[i] = reinterpret_cast<const float4 *>(a.params)[i];

    // ---------------- activations: uint4[KS] in lane order (word w of k-step ks = channels 16 ks + {4 hg + 2 w', 8 + 4 hg + 2 w'})
    // Long-lived per-lane values (row pointers, table offsets) are recomputed where they are used: kept in a register across a stage
    // they get spilled, and a scratch reload's s_waitcnt vmcnt(0) drains the whole DMA queue
    auto fresh_lane = [&]() __attribute__((always_inline)) -> unsigned {
        unsigned l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    };
    const int64_t row0_in = a.in_rows > 0 ? row0 % a.in_rows : row0;     // (in_rows % 128 == 0: a workgroup's rows never straddle the wrap)
    auto row_off = [&](int64_t base = -1) __attribute__((always_inline)) -> int64_t {      // element offset of this lane's 4-channel group 0 in a [M][320] tensor
        const unsigned l = fresh_lane();
        return ((base < 0 ? row0 : base) + wid * 32 + (l & 31)) * TC + 4 * (l >> 5);
    };
    auto load_rows = [&](const unsigned short *p, uint4 *dst) __attribute__((always_inline)) {
        const unsigned short *r = p + row_off(row0_in);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const uint2 lo = *reinterpret_cast<const uint2 *>(r + 16 * ks), hi = *reinterpret_cast<const uint2 *>(r + 16 * ks + 8);
            dst[ks] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
    };
    auto store_rows = [&](const uint4 *src) __attribute__((always_inline)) {
        unsigned short *r = a.out + row_off();
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            *reinterpret_cast<uint2 *>(r + 16 * ks) = make_uint2(src[ks].x, src[ks].y);
            *reinterpret_cast<uint2 *>(r + 16 * ks + 8) = make_uint2(src[ks].z, src[ks].w);
        }
    };
    auto lo_f = [](unsigned w) __attribute__((always_inline)) { return T::to_f((unsigned short)(w & 0xffff)); };
    auto hi_f = [](unsigned w) __attribute__((always_inline)) { return T::to_f((unsigned short)(w >> 16)); };

    uint4 xf[KS], hres[KS];
    f32x16 acc[NB];
    load_rows(a.o1, xf);
    if (a.h_frags) {                       // written by the head kernel as the fragments themselves: 1 KB per load instruction
        const uint4 *r = reinterpret_cast<const uint4 *>(a.h) + ((row0_in >> 5) + wid) * (KS * 64) + fresh_lane();
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) hres[ks] = r[ks * 64];
    } else load_rows(a.h, hres);
    __syncthreads();                       // parameter table visible

    auto mma = [&](uint4 wv, uint4 xv, f32x16 c) __attribute__((always_inline)) -> f32x16 {
        if (ABL & 8) { c[0] += __uint_as_float(wv.x ^ xv.x); return c; }
        return T::mfma32(wv, xv, c);
    };
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    };
    // acc[nb] += W[32 nb ..][k] x[k], blocks in (ks, nb) order.  epi(nb) is called as soon as block nb is complete (two MFMAs later, so
    // that its result is out of the pipeline): the epilogue's VALU work runs in the shadow of the last MFMAs, and the accumulators are
    // read out of the AGPRs block by block (read all at once they would need 160 more registers than there are)
    auto gemm = [&](int base, const uint4 *x, auto &&epi) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                acc[nb] = mma(blk(base, ks * NB + nb), x[ks], acc[nb]);
                if (ks == KS - 1 && nb >= 2) epi(nb - 2);
            }
        epi(NB - 2);
        epi(NB - 1);
    };
    // acc = bias + residual (lane order): every residual connection enters as the accumulators' initial value -- fp32, added before the
    // products like the per-op epilogue adds it after them, one rounding at the end -- so the epilogues only round, and no residual has
    // to wait anywhere (parked in memory its reload would cost a vmcnt(0), i.e. a drained DMA queue)
    auto init_acc = [&](int pbias, const uint4 *res) __attribute__((always_inline)) {
        const float *prm_l = prm + 16 * (fresh_lane() >> 5);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            float bv[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<float4 *>(bv + 4 * q) = *reinterpret_cast<const float4 *>(prm_l + (pbias + 32 * nb + 4 * q));
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint4 rr = res[2 * nb + j];
                unsigned rw[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
                for (int w = 0; w < 4; ++w) asm volatile("" : "+v"(rw[w]));      // unpack here (not CSE'd with a LayerNorm pass 160 registers ago)
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    acc[nb][8 * j + 2 * w] = bv[8 * j + 2 * w] + lo_f(rw[w]);
                    acc[nb][8 * j + 2 * w + 1] = bv[8 * j + 2 * w + 1] + hi_f(rw[w]);
                }
            }
            asm volatile("" : "+a"(acc[nb]));          // into the AGPRs now (left to itself hipcc keeps VGPR copies and spills them)
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto frag_block = [&](uint4 *dst, int nb) __attribute__((always_inline)) {        // round the accumulators (no bias) into lane order
#pragma unroll
        for (int j = 0; j < 2; ++j)
            dst[2 * nb + j] = make_uint4(pack2<T>(acc[nb][8 * j], acc[nb][8 * j + 1]), pack2<T>(acc[nb][8 * j + 2], acc[nb][8 * j + 3]),
                                         pack2<T>(acc[nb][8 * j + 4], acc[nb][8 * j + 5]), pack2<T>(acc[nb][8 * j + 6], acc[nb][8 * j + 7]));
    };
    // dst = round(LayerNorm(src) * gamma + beta); a row's 320 channels live in the lane pair (m, 0) / (m, 1)
    auto layernorm = [&](const uint4 *src, int pg, int pb, uint4 *dst) __attribute__((always_inline)) {
        const float *prm_l = prm + 16 * (fresh_lane() >> 5);
        float s = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const unsigned w[4] = {src[ks].x, src[ks].y, src[ks].z, src[ks].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) s += lo_f(w[i]) + hi_f(w[i]);
        }
        s += __shfl_xor(s, 32, 64);
        const float mean = s * (1.f / TC);
        float v = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            unsigned w[4] = {src[ks].x, src[ks].y, src[ks].z, src[ks].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(w[i]));        // unpack again: keeping 160 floats alive over the three passes spills
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
            *reinterpret_cast<float4 *>(g) = *reinterpret_cast<const float4 *>(prm_l + (pg + o));
            *reinterpret_cast<float4 *>(g + 4) = *reinterpret_cast<const float4 *>(prm_l + (pg + o + 4));
            *reinterpret_cast<float4 *>(b) = *reinterpret_cast<const float4 *>(prm_l + (pb + o));
            *reinterpret_cast<float4 *>(b + 4) = *reinterpret_cast<const float4 *>(prm_l + (pb + o + 4));
            unsigned w[4] = {src[ks].x, src[ks].y, src[ks].z, src[ks].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(w[i]));
            unsigned ow[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                ow[i] = pack2<T>((lo_f(w[i]) - mean) * rstd * g[2 * i] + b[2 * i], (hi_f(w[i]) - mean) * rstd * g[2 * i + 1] + b[2 * i + 1]);
            dst[ks] = make_uint4(ow[0], ow[1], ow[2], ow[3]);
            if (ks & 1) __builtin_amdgcn_sched_barrier(0);
        }
    };
    // ---------------- 1: h1 = to_out1(o1) + h
    STAMP(0);
    init_acc(P_BO1, hres);
    gemm(G_O1, xf, [&](int nb) { frag_block(hres, nb); });
    STAMP(1);
    STAMP(2);
    if constexpr (STOP == 1) { store_rows(hres); wait_vmcnt<0>(); return; }

    // ---------------- 2: q2 = to_q2(LN2(h1));  o2 = softmax(q2 Kt^T) Vt per head;  h2 = to_out2(o2) + h1
    layernorm(hres, P_G2, P_B2, xf);
    STAMP(3);
    uint4 qf[KS];                          // q2 (already scaled by D^-1/2 log2 e on the host), lane order
    zero_acc();
    gemm(G_Q2, xf, [&](int nb) { frag_block(qf, nb); });
    STAMP(4);
    STAMP(5);
    if constexpr (STOP == 2) { store_rows(qf); wait_vmcnt<0>(); return; }
    // h1 waits in 80 AGPRs while the attention needs the VGPRs (explicitly: hipcc would spill it to scratch instead)
    unsigned park[4 * KS];
#pragma unroll
    for (int i = 0; i < KS; ++i) {
        asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(park[4 * i]) : "v"(hres[i].x));
        asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(park[4 * i + 1]) : "v"(hres[i].y));
        asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(park[4 * i + 2]) : "v"(hres[i].z));
        asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(park[4 * i + 3]) : "v"(hres[i].w));
    }
    {
        uint4 *of = xf;                    // LN2's output is dead: the attention output takes its registers
        // 8-channel group `grp` (channels 8 grp .. 8 grp + 7): this lane's 4 of them are 2 words of a lane-order array
        auto grp_get = [&](const uint4 *p, int grp, unsigned &w0, unsigned &w1) __attribute__((always_inline)) {
            const uint4 v = p[grp >> 1];
            if (grp & 1) { w0 = v.z; w1 = v.w; } else { w0 = v.x; w1 = v.y; }
        };
        auto grp_set = [&](uint4 *p, int grp, unsigned w0, unsigned w1) __attribute__((always_inline)) {
            if (grp & 1) { p[grp >> 1].z = w0; p[grp >> 1].w = w1; } else { p[grp >> 1].x = w0; p[grp >> 1].y = w1; }
        };
        static_for<0, TH>([&](auto hc) {
            constexpr int h = decltype(hc)::value;
            // S^T[kb] = K[32 kb ..][d] q[d], d padded 40 -> 48 (3 k-steps; the last one half empty)
            f32x16 st[3];
#pragma unroll
            for (int kb = 0; kb < 3; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
            static_for<0, 3>([&](auto kc) {
                constexpr int ks = decltype(kc)::value;
                uint4 qb;
                grp_get(qf, 5 * h + 2 * ks, qb.x, qb.y);
                if constexpr (2 * ks + 1 < 5) grp_get(qf, 5 * h + 2 * ks + 1, qb.z, qb.w);
                else { qb.z = 0; qb.w = 0; }
#pragma unroll
                for (int kb = 0; kb < 3; ++kb) st[kb] = mma(blk(G_KV, h * 21 + ks * 3 + kb), qb, st[kb]);
            });
            float mx = -3.0e38f;
#pragma unroll
            for (int kb = 0; kb < 3; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = 32 * kb + 8 * (r >> 2) + 4 * hg + (r & 3);
                    st[kb][r] = key < a.Lt ? st[kb][r] : -3.0e38f;
                    mx = fmaxf(mx, st[kb][r]);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            uint4 pf[6];
#pragma unroll
            for (int kb = 0; kb < 3; ++kb) {
                float p[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) p[r] = __builtin_amdgcn_exp2f(st[kb][r] - mx);
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    pf[2 * kb + j] = make_uint4(pack2<T>(p[8 * j], p[8 * j + 1]), pack2<T>(p[8 * j + 2], p[8 * j + 3]),
                                                pack2<T>(p[8 * j + 4], p[8 * j + 5]), pack2<T>(p[8 * j + 6], p[8 * j + 7]));
            }
            // O^T[rb] = Vt[32 rb ..][key] P^T; row 40 of Vt is ones: O^T[40] = the denominator with P's rounding
            f32x16 ot[2];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) ot[rb][r] = 0.f;
#pragma unroll
            for (int kv = 0; kv < 6; ++kv)
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) ot[rb] = mma(blk(G_KV, h * 21 + 9 + kv * 2 + rb), pf[kv], ot[rb]);
            // d = 40 = 32 + 8: block 1, g = 1, hg = 0, c = 0 -> register 4 of the hg = 0 lane
            const float l = __shfl(ot[1][4], m, 64);
            const float inv = 1.f / l;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                grp_set(of, 5 * h + g, pack2<T>(ot[0][4 * g] * inv, ot[0][4 * g + 1] * inv), pack2<T>(ot[0][4 * g + 2] * inv, ot[0][4 * g + 3] * inv));
            grp_set(of, 5 * h + 4, pack2<T>(ot[1][0] * inv, ot[1][1] * inv), pack2<T>(ot[1][2] * inv, ot[1][3] * inv));
        });
        if constexpr (STOP == 3) { store_rows(of); wait_vmcnt<0>(); return; }
        STAMP(6);
#pragma unroll
        for (int i = 0; i < KS; ++i) {
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(hres[i].x) : "a"(park[4 * i]));
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(hres[i].y) : "a"(park[4 * i + 1]));
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(hres[i].z) : "a"(park[4 * i + 2]));
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(hres[i].w) : "a"(park[4 * i + 3]));
        }
        init_acc(P_BO2, hres);
        gemm(G_O2, of, [&](int nb) { frag_block(hres, nb); });
    }
    STAMP(7);
    if constexpr (STOP == 4) { store_rows(hres); wait_vmcnt<0>(); return; }

    // ---------------- 3: h3 = down(GEGLU(up(LN3(h2)))) + h2
    layernorm(hres, P_G3, P_B3, xf);
    STAMP(8);
    init_acc(P_BDN, hres);                 // h2 + bias: the down projection accumulates on top of them
    // one iteration = 64 inner channels: 4 up-blocks (80 MFMAs), GEGLU, 4 k-steps of the down projection (40 MFMAs).  The last one is
    // peeled: h2 comes back from `out` while its MFMAs run, and the epilogue of every finished accumulator block runs in their shadow.
    auto ff_iter = [&](int it, auto last_c) __attribute__((always_inline)) {
        constexpr bool LAST = decltype(last_c)::value;
        STAMP(10 + 3 * it);
        f32x16 up[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) up[j][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < 4; ++j) up[j] = mma(blk(G_FF, ks * 4 + j, !LAST), xf[ks], up[j]);
        STAMP(11 + 3 * it);
        // the lane's table offset is recomputed here: as a loop-invariant register it gets spilled, and the reload's vmcnt(0) drains the DMA queue
        unsigned ln;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
        const float *prm_up = prm + P_BUP + it * 128 + ((ln >> 5) << 4);
        uint4 ff[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // rows of an up-block: groups g = 0, 2 hold 8 hidden channels each, g = 1, 3 their gates (host permutation)
            float bv[16];
            const float *bp = prm_up + j * 32;
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<float4 *>(bv + 4 * q) = *reinterpret_cast<const float4 *>(bp + 4 * q);
            float o[8];
#pragma unroll
            for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float hid = up[j][8 * pr + c] + bv[8 * pr + c], gate = up[j][8 * pr + 4 + c] + bv[8 * pr + 4 + c];
                    // segment index clamped, the fraction NOT: beyond +-8 the end segments extrapolate (slopes 1 and 0 to 1e-14: gelu's asymptotes)
                    const float t = __builtin_fmaf(gate, GELU_STEP, 0.5f * GELU_N);
                    const float ti = __builtin_floorf(__builtin_amdgcn_fmed3f(t, 0.f, (float)(GELU_N - 1)));
                    const float2 e = reinterpret_cast<const float2 *>(prm + P_LUT)[(int)ti];
                    o[4 * pr + c] = hid * __builtin_fmaf(t - ti, e.y, e.x);
                }
            ff[j] = pack8<T>(o);
            __builtin_amdgcn_sched_barrier(0);         // one block at a time (16 bias + 16 accumulator registers, not 128)
        }
        STAMP(12 + 3 * it);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                acc[nb] = mma(blk(G_FF, KS * 4 + j * NB + nb, !LAST), ff[j], acc[nb]);
                if constexpr (LAST)
                    if (j == 3 && nb >= 2) frag_block(hres, nb - 2);
            }
        if constexpr (LAST) {
            frag_block(hres, NB - 2);
            frag_block(hres, NB - 1);
        }
    };
#pragma unroll 1
    for (int it = 0; it < FF_IT - 1; ++it) ff_iter(it, std::false_type{});
    ff_iter(FF_IT - 1, std::true_type{});
    STAMP(9);
    if constexpr (STOP == 5) { store_rows(hres); wait_vmcnt<0>(); return; }

    // ---------------- 4: out = proj_out(h3) + x
    STAMP(60);
    load_rows(a.x, xf);                    // the block's input (proj_out's residual); loaded here, not under the last MFMAs: 80 more live
                                           // registers there spill, and a spill reload drains the DMA queue
    init_acc(P_BPO, xf);
    gemm(G_PO, hres, [&](int nb) { frag_block(xf, nb); });
    STAMP(61);
    store_rows(xf);
    STAMP(62);
}

template <class T, int ABL = 0, int STOP = 0> int launch(const TailArgs &a, hipStream_t s)
{
    static gc::AttrOnce once;
    gc::ensure_dynamic_lds(once, (const void *)k_ttail<T, ABL, STOP>, LDS_BYTES);
    hipLaunchKernelGGL((k_ttail<T, ABL, STOP>), dim3((unsigned)((a.M + 127) / 128)), dim3(256), LDS_BYTES, s, a);
    return gc::check_launch("gc_dn_transformer_tail");
}

}  // namespace

extern "C" int gc_dn_transformer_tail(const gc_ttail_desc *d, void *stream)
{
    GC_REQUIRE(d && d->attn_out && d->resid && d->x_in && d->out && d->w_a && d->w_kv && d->w_b && d->params, "NULL argument");
    GC_REQUIRE(d->channels == TC && d->heads == TH, "the fused tail is built for C = 320, 8 heads (SD1.5 level 0)");
    GC_REQUIRE(d->M > 0 && d->rows_per_frame > 0 && d->rows_per_frame % 128 == 0 && d->M % d->rows_per_frame == 0, "rows_per_frame must be a multiple of 128 dividing M");
    GC_REQUIRE(d->frames_per_half > 0 && (d->M / d->rows_per_frame) % d->frames_per_half == 0 && (d->M / d->rows_per_frame) / d->frames_per_half <= 2, "at most two CFG halves");
    GC_REQUIRE(d->text_len > 0 && d->text_len <= 96, "text length <= 96");
    TailArgs a;
    a.o1 = (const unsigned short *)d->attn_out; a.h = (const unsigned short *)d->resid; a.x = (const unsigned short *)d->x_in;
    a.out = (unsigned short *)d->out;
    a.wa = (const unsigned char *)d->w_a; a.wkv = (const unsigned char *)d->w_kv; a.wb = (const unsigned char *)d->w_b;
    a.params = d->params; a.M = (int)d->M; a.rows_per_frame = (int)d->rows_per_frame; a.f = d->frames_per_half; a.Lt = d->text_len;
    a.eps = d->ln_eps; a.stop = d->stop_after & 7; a.h_frags = d->resid_fragment_layout;
    GC_REQUIRE(d->in_rows >= 0 && (d->in_rows == 0 || (d->in_rows % 128 == 0 && d->M % d->in_rows == 0)), "in_rows must be 0 or a multiple of 128 dividing M");
    a.in_rows = (int)d->in_rows;
#ifdef TTAIL_ABLATIONS
    if (d->dtype == DT_BF16 && (d->stop_after & 7)) {
        switch (d->stop_after & 7) {
        case 1: return launch<BF16, 0, 1>(a, (hipStream_t)stream);
        case 2: return launch<BF16, 0, 2>(a, (hipStream_t)stream);
        case 3: return launch<BF16, 0, 3>(a, (hipStream_t)stream);
        case 4: return launch<BF16, 0, 4>(a, (hipStream_t)stream);
        case 5: return launch<BF16, 0, 5>(a, (hipStream_t)stream);
        }
    }
    if (d->dtype == DT_BF16 && (d->stop_after >> 3)) {
        switch (d->stop_after >> 3) {
        case 1: return launch<BF16, 1>(a, (hipStream_t)stream);
        case 2: return launch<BF16, 2>(a, (hipStream_t)stream);
        case 3: return launch<BF16, 3>(a, (hipStream_t)stream);
        case 4: return launch<BF16, 4>(a, (hipStream_t)stream);
        case 7: return launch<BF16, 7>(a, (hipStream_t)stream);
        case 8: return launch<BF16, 8>(a, (hipStream_t)stream);
        }
    }
#else
    GC_REQUIRE(d->stop_after == 0, "stage outputs / ablations need a development build (make TTAIL_FLAGS=-DTTAIL_ABLATIONS)");
#endif
    if (d->dtype == DT_BF16) return launch<BF16>(a, (hipStream_t)stream);
    if (d->dtype == DT_F16) return launch<F16>(a, (hipStream_t)stream);
    GC_REQUIRE(false, "dtype");
}

#ifdef TTAIL_ABLATIONS
extern "C" void gc_dn_transformer_tail_stamps(unsigned long long *host64)
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(host64, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 64);
}
#endif

extern "C" void gc_dn_transformer_tail_layout(int64_t *blocks_a, int64_t *blocks_kv, int64_t *blocks_b, int64_t *param_floats)
{
    *blocks_a = BLK_A; *blocks_kv = BLK_KV; *blocks_b = BLK_B; *param_floats = P_TOTAL;
}
