// dn_gemm.hip -- bf16/f16 MFMA GEMM and implicit-GEMM 3x3 convolution for the SD1.5 UNet / ControlNet /
// VAE blocks (gfx950).  Replaces the cuBLAS / cuDNN calls diffusers issues for every Linear and Conv2d
// of UNet2DConditionModel / ControlNetModel / AutoencoderKL reached from
// /root/reference/gaussctrl/gc_pipeline.py:142-145,209-219.
//
// out[m][n] = epilogue( sum_k Act[m][k] * W[n][k] ),  Act = row-major matrix (Linear / 1x1 conv on NHWC)
// or the on-the-fly im2col of an NHWC tensor (3x3, pad 1, stride 1|2, optional fused nearest x2 upsample).
//
// CDNA4 mapping: 128(m) x 128(n) x 64(k) workgroup tile, 256 lanes = 4 wave64 in 2x2, each wave a
// 64x64 sub-tile = 4x4 v_mfma_f32_16x16x32 accumulators (fp32).  Operands are staged global -> VGPR ->
// LDS (16-byte chunks, XOR-swizzled so every ds_read_b128 fragment read is bank-conflict free), double
// buffered with the next tile's global loads in flight under the MFMAs (one barrier per k-tile).
// The MFMA is issued "swapped" (A operand = weights, B operand = activations) so each lane ends up
// with 4 CONSECUTIVE output channels of one output row: bias / residual / GEGLU are lane-local and the
// store is one 8-byte write per accumulator.
#include "dn_common.h"

namespace {
using namespace dn;

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int NT = 256;

struct GemmArgs {
    int64_t M, N, K;
    const void *A; int64_t lda;
    int mode;   // 0 linear, 1 conv3x3
    int B, Hi, Wi, Cin, Ho, Wo, stride, ups, pad;
    const void *W;
    const float *bias;
    const float *rowvec; int64_t ld_rowvec; int64_t rows_per_batch;
    const void *residual; int64_t ldr;
    float out_scale;
    int act, geglu;
    void *out; int64_t ldc; int out_f32;
    void *out_t; int64_t ldt; int64_t t_batch_stride;
};

// byte offset of 16-byte chunk `c` (0..7) of row `r` inside a [rows][64] 2-byte tile
__device__ __forceinline__ int lds_off(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

template <class T, bool CONV>
__global__ __launch_bounds__(NT) void k_gemm(const GemmArgs g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // layout: [buf][Act 16 KiB | W 16 KiB]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    // XCD-aware tile order: consecutive workgroups on one XCD share the activation panel (same m-block)
    const int nbn = (int)((g.N + BN - 1) / BN);
    // workgroup b runs on XCD b % 8: give each XCD a contiguous run of logical tiles (bijective remap)
    const int64_t nwg = gridDim.x, xcd = blockIdx.x & 7, qq = nwg >> 3, rr = nwg & 7;
    const int64_t bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (blockIdx.x >> 3);
    const int64_t mblk = bid / nbn, nblk = bid % nbn;
    const int64_t m_base = mblk * BM, n_base = nblk * BN;

    // ---- per-thread staging coordinates: 4 chunks of Act and 4 chunks of W per k-tile
    int a_row[4], a_chunk[4];
    const unsigned char *a_ptr[4];      // linear: row base pointer (or null if row >= M)
    int a_b[4], a_y[4], a_x[4];          // conv: output pixel
    bool a_ok[4];
    const unsigned char *w_ptr[4];
    bool w_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = tid + NT * i;
        a_row[i] = q >> 3; a_chunk[i] = q & 7;
        const int64_t m = m_base + a_row[i];
        a_ok[i] = m < g.M;
        if (CONV) {
            const int64_t mm = a_ok[i] ? m : 0;
            const int hw = g.Ho * g.Wo;
            a_b[i] = (int)(mm / hw);
            const int rem = (int)(mm - (int64_t)a_b[i] * hw);
            a_y[i] = rem / g.Wo; a_x[i] = rem - a_y[i] * g.Wo;
            a_ptr[i] = nullptr;
        } else {
            a_ptr[i] = (const unsigned char *)g.A + (a_ok[i] ? m : 0) * g.lda * 2;
        }
        const int64_t n = n_base + a_row[i];
        w_ok[i] = n < g.N;
        w_ptr[i] = (const unsigned char *)g.W + (w_ok[i] ? n : 0) * g.K * 2;
    }
    const int Hin = g.ups ? g.Hi * 2 : g.Hi, Win = g.ups ? g.Wi * 2 : g.Wi;

    uint4 ra[4], rw[4];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t k0 = (int64_t)kt * BK + a_chunk[i] * 8;
            uint4 va = make_uint4(0, 0, 0, 0), vw = make_uint4(0, 0, 0, 0);
            if (k0 < g.K) {
                if (CONV) {
                    const int tap = (int)(k0 / g.Cin);
                    const int ci = (int)(k0 - (int64_t)tap * g.Cin);
                    const int dy = tap / 3, dx = tap - dy * 3;
                    int yi = a_y[i] * g.stride + dy - g.pad, xi = a_x[i] * g.stride + dx - g.pad;
                    if (a_ok[i] && yi >= 0 && yi < Hin && xi >= 0 && xi < Win) {
                        if (g.ups) { yi >>= 1; xi >>= 1; }
                        const int64_t off = (((int64_t)a_b[i] * g.Hi + yi) * g.Wi + xi) * g.Cin + ci;
                        va = *reinterpret_cast<const uint4 *>((const unsigned char *)g.A + off * 2);
                    }
                } else if (a_ok[i]) {
                    va = *reinterpret_cast<const uint4 *>(a_ptr[i] + k0 * 2);
                }
                if (w_ok[i]) vw = *reinterpret_cast<const uint4 *>(w_ptr[i] + k0 * 2);
            }
            ra[i] = va; rw[i] = vw;
        }
    };
    auto store_tile = [&](int buf) {
        unsigned char *sa = smem + buf * 32768, *sw = sa + 16384;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<uint4 *>(sa + lds_off(a_row[i], a_chunk[i])) = ra[i];
            *reinterpret_cast<uint4 *>(sw + lds_off(a_row[i], a_chunk[i])) = rw[i];
        }
    };

    f32x4 acc[4][4];   // [nt][mt]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (int)((g.K + BK - 1) / BK);
    load_tile(0);
    store_tile(0);
    __syncthreads();
    const int fr = lane & 15, fc = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const unsigned char *sa = smem + buf * 32768, *sw = sa + 16384;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 fw[4], fa[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                fw[t] = *reinterpret_cast<const uint4 *>(sw + lds_off(wn * 64 + t * 16 + fr, ks * 4 + fc));
                fa[t] = *reinterpret_cast<const uint4 *>(sa + lds_off(wm * 64 + t * 16 + fr, ks * 4 + fc));
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = T::mfma(fw[nt], fa[mt], acc[nt][mt]);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds out[m = m0 + (lane&15)][n = n0 + 4*(lane>>4) + r], r = 0..3
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int64_t m = m_base + wm * 64 + mt * 16 + fr;
        if (m >= g.M) continue;
        const int64_t bidx = g.rowvec ? m / g.rows_per_batch : 0;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            if (g.geglu && (nt & 1)) continue;
            const int64_t n = n_base + wn * 64 + nt * 16 + fc * 4;
            if (n >= g.N) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = acc[nt][mt][r];
                if (g.bias) x += g.bias[n + r];
                if (g.rowvec) x += g.rowvec[bidx * g.ld_rowvec + n + r];
                v[r] = x;
            }
            int64_t on = n;
            if (g.geglu) {   // weights are row-permuted in 16-blocks [x | gate]; partner tile = nt + 1
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float gt = acc[nt + 1][mt][r];
                    if (g.bias) gt += g.bias[n + 16 + r];
                    v[r] = v[r] * gelu_erf(gt);
                }
                on = (n_base + wn * 64 + nt * 16) / 2 + fc * 4;
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
                const uint2 rr = *reinterpret_cast<const uint2 *>((const unsigned char *)g.residual + (m * g.ldr + on) * 2);
                v[0] += T::to_f((unsigned short)(rr.x & 0xffff)); v[1] += T::to_f((unsigned short)(rr.x >> 16));
                v[2] += T::to_f((unsigned short)(rr.y & 0xffff)); v[3] += T::to_f((unsigned short)(rr.y >> 16));
            }
            if (g.out) {
                if (g.out_f32) {
                    *reinterpret_cast<float4 *>((float *)g.out + m * g.ldc + on) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    *reinterpret_cast<uint2 *>((unsigned char *)g.out + (m * g.ldc + on) * 2) =
                        make_uint2(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]));
                }
            }
            if (g.out_t) {   // transposed copy out_t[b][n][tok] (V operand of the attention kernel)
                const int64_t b = m / g.rows_per_batch, tok = m - b * g.rows_per_batch;
                unsigned short *o = (unsigned short *)g.out_t + b * g.t_batch_stride + tok;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[(on + r) * g.ldt] = T::from_f(v[r]);
            }
        }
    }
}

}  // namespace

extern "C" int gc_dn_gemm(const gc_gemm_desc *d, void *stream)
{
    GC_REQUIRE(d && d->W && d->A, "null descriptor / operand");
    GC_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "empty problem");
    GC_REQUIRE(d->K % 8 == 0, "K must be a multiple of 8 (pad channels on the host)");
    GC_REQUIRE(d->N % 4 == 0, "N must be a multiple of 4 (pad output channels on the host)");
    GC_REQUIRE(d->dtype == DT_BF16 || d->dtype == DT_F16, "dtype must be 0 (bf16) or 1 (f16)");
    GemmArgs g;
    g.M = d->M; g.N = d->N; g.K = d->K; g.A = d->A; g.lda = d->lda; g.mode = d->mode;
    g.B = d->B; g.Hi = d->Hi; g.Wi = d->Wi; g.Cin = d->Cin; g.Ho = d->Ho; g.Wo = d->Wo; g.stride = d->stride; g.ups = d->upsample; g.pad = d->pad_lo;
    g.W = d->W; g.bias = d->bias; g.rowvec = d->rowvec; g.ld_rowvec = d->ld_rowvec;
    g.rows_per_batch = d->rows_per_batch > 0 ? d->rows_per_batch : 1;
    g.residual = d->residual; g.ldr = d->ldr; g.out_scale = d->out_scale; g.act = d->act; g.geglu = d->geglu;
    g.out = d->out; g.ldc = d->ldc; g.out_f32 = d->out_f32; g.out_t = d->out_t; g.ldt = d->ldt; g.t_batch_stride = d->t_batch_stride;
    if (d->mode == 1) {
        GC_REQUIRE(d->Cin % 8 == 0 && d->K == 9 * (int64_t)d->Cin, "conv3x3: K must be 9*Cin with Cin % 8 == 0");
        GC_REQUIRE(d->M == (int64_t)d->B * d->Ho * d->Wo, "conv3x3: M must be B*Ho*Wo");
        GC_REQUIRE(d->pad_lo == 0 || d->pad_lo == 1, "conv3x3: pad_lo must be 0 or 1");
    } else {
        GC_REQUIRE(d->lda >= d->K && d->lda % 8 == 0, "linear: lda must be >= K and a multiple of 8");
    }
    if (d->geglu) GC_REQUIRE(d->N % 32 == 0 && !d->out_t, "geglu needs N % 32 == 0");
    const int64_t nbm = (d->M + BM - 1) / BM, nbn = (d->N + BN - 1) / BN;
    const dim3 grid((unsigned)(nbm * nbn)), block(NT);
    const size_t lds = 65536;
    hipStream_t s = gc::S(stream);
#define GC_LAUNCH(T, C)                                                                            \
    do {                                                                                           \
        static bool attr_set = false;                                                              \
        if (!attr_set) {                                                                           \
            (void)hipFuncSetAttribute((const void *)k_gemm<T, C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            attr_set = true;                                                                       \
        }                                                                                          \
        hipLaunchKernelGGL((k_gemm<T, C>), grid, block, lds, s, g);                                \
    } while (0)
    if (d->dtype == DT_BF16) { if (d->mode == 1) GC_LAUNCH(BF16, true); else GC_LAUNCH(BF16, false); }
    else { if (d->mode == 1) GC_LAUNCH(F16, true); else GC_LAUNCH(F16, false); }
#undef GC_LAUNCH
    return gc::check_launch("gc_dn_gemm");
}
