// Issue-packing microbenchmark for the k_attn5 tile (round 6): the tile's 10 v_mfma_f32_32x32x16_bf16 + 4 v_mfma_f32_16x16x32_bf16 in the product order
// (C: 3 big, D: 2 big + 2 small, F: 3 big, G: 2 big + 2 small) with the 32 v_exp_f32 / 16 v_cvt_pk_bf16_f32 / 8 v_permlane16_swap of a wave DISTRIBUTED over
// the 14 MFMA shadows by a table (VALU detached from the MFMAs: attn_pingpong.hip shows the real dependencies cost nothing, the loop is issue-bound).
// Which distribution is fastest?  build: hipcc --offload-arch=gfx950 -O3 -o attn_pack attn_pack.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
struct Tab { int e[14], c[14], s[14]; };
// slot kinds: 0 = 32x32x16, 1 = 16x16x32
__device__ constexpr int KIND[14] = {0, 0, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 1, 1};

template <int I, int N, class F> __device__ __forceinline__ void sfor(F &&f) { if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); } }

template <int V> struct T;
// V0: the product's distribution (after the round-6 in-place swaps)
template <> struct T<0> { static constexpr Tab t = {{2, 2, 4, 0, 2, 2, 2, 4, 4, 2, 0, 4, 2, 2}, {1, 1, 2, 0, 1, 1, 1, 2, 2, 1, 0, 2, 1, 1}, {0, 0, 0, 0, 4, 0, 0, 0, 0, 0, 0, 4, 0, 0}}; };
// V1: 3 exp under every big MFMA, the rest (2 exp, cvts, swaps) under the small ones and the last big ones
template <> struct T<1> { static constexpr Tab t = {{3, 3, 3, 3, 3, 1, 0, 3, 3, 3, 3, 3, 1, 0}, {1, 1, 1, 1, 1, 2, 2, 1, 1, 1, 1, 1, 1, 1}, {0, 0, 0, 0, 0, 2, 2, 0, 0, 0, 0, 0, 2, 2}}; };
// V2: 3 exp + 1 cvt under every big MFMA, small ones: 1 cvt + 2 swaps (+ the 2 left-over exp)
template <> struct T<2> { static constexpr Tab t = {{3, 3, 3, 3, 3, 1, 0, 3, 3, 3, 3, 3, 1, 0}, {1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 2}, {1, 0, 1, 0, 1, 1, 0, 1, 0, 1, 0, 1, 1, 0}}; };
// V3: uniform 2 exp + 1 cvt under 12 of the MFMAs (incl. the small ones), 4 exp + 2 cvt under two, swaps spread
template <> struct T<3> { static constexpr Tab t = {{2, 2, 2, 4, 2, 2, 2, 2, 2, 2, 4, 2, 2, 2}, {1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1}, {1, 1, 0, 0, 1, 0, 1, 1, 1, 0, 0, 1, 0, 1}}; };
// V4: everything under the big MFMAs, small ones bare
template <> struct T<4> { static constexpr Tab t = {{3, 3, 3, 4, 3, 0, 0, 3, 3, 3, 4, 3, 0, 0}, {2, 1, 2, 1, 2, 0, 0, 2, 1, 2, 1, 2, 0, 0}, {1, 1, 1, 0, 1, 0, 0, 1, 1, 1, 0, 1, 0, 0}}; };
// V5: the product's before round 6 (swaps right after the first P V MFMA + 8 v_mov)
template <> struct T<5> { static constexpr Tab t = {{2, 2, 4, 0, 2, 2, 2, 4, 4, 2, 0, 4, 2, 2}, {1, 1, 2, 0, 1, 1, 1, 2, 2, 1, 0, 2, 1, 1}, {0, 0, 0, 8, 0, 0, 0, 0, 0, 0, 8, 0, 0, 0}}; };
// V6: no VALU at all
template <> struct T<6> { static constexpr Tab t = {{0}, {0}, {0}}; };
// V7: 2 exp + 1 cvt under each big, 3 exp + 1.5 cvt under small?  (small ones carry more)
template <> struct T<7> { static constexpr Tab t = {{2, 2, 2, 2, 2, 3, 3, 2, 2, 2, 2, 2, 3, 3}, {1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 2, 1}, {1, 0, 1, 0, 1, 0, 1, 1, 0, 1, 0, 1, 0, 1}}; };

template <int V>
__global__ __launch_bounds__(512, 1) void k(float *out, long long *cyc, int iters)
{
    constexpr Tab t = T<V>::t;
    uint4 a = make_uint4(0x3c003c00u + threadIdx.x, 0x3c003c00u, 0x3c003c00u, 0x3c003c01u), b = make_uint4(a.y, a.x, a.w, a.z);
    f32x16 big[4];
    f32x4 sm[4];
    for (int i = 0; i < 4; ++i) { for (int r = 0; r < 16; ++r) big[i][r] = 0.f; sm[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    float v[16];
    unsigned w[8];
    for (int i = 0; i < 16; ++i) v[i] = -1.f - 0.01f * i - 1e-4f * threadIdx.x;
    for (int i = 0; i < 8; ++i) w[i] = threadIdx.x * 7 + i;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        __builtin_amdgcn_s_barrier();
        sfor<0, 14>([&](auto j_) __attribute__((always_inline)) {
            constexpr int j = decltype(j_)::value;
            if constexpr (KIND[j] == 0) big[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), big[j & 3], 0, 0, 0);
            else sm[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), sm[j & 3], 0, 0, 0);
            constexpr int ne = t.e[j], nc = t.c[j], ns = t.s[j];
#pragma unroll
            for (int i = 0; i < ne; ++i) { const int r = (j * 3 + i) & 15; asm volatile("v_exp_f32 %0, %1" : "=v"(v[r]) : "v"(v[(r + 5) & 15])); }
#pragma unroll
            for (int i = 0; i < nc; ++i) { const int r = (j + i) & 7; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[r]) : "v"(v[(2 * r) & 15]), "v"(v[(2 * r + 1) & 15])); }
#pragma unroll
            for (int i = 0; i < ns; ++i) { const int r = (2 * (j + i)) & 6; asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(w[r]), "+v"(w[r + 1])); }
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) { for (int r = 0; r < 16; ++r) s += big[i][r]; s += sm[i][0] + sm[i][1] + sm[i][2] + sm[i][3]; }
    for (int i = 0; i < 16; ++i) s += v[i];
    for (int i = 0; i < 8; ++i) s += __uint_as_float(w[i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int V> void run(const char *name)
{
    constexpr Tab t = T<V>::t;
    int ne = 0, nc = 0, ns = 0;
    for (int j = 0; j < 14; ++j) { ne += t.e[j]; nc += t.c[j]; ns += t.s[j]; }
    float *out; long long *cyc;
    hipMalloc(&out, sizeof(float) * 512 * 256); hipMalloc(&cyc, 8);
    const int iters = 20000;
    hipLaunchKernelGGL(k<V>, dim3(256), dim3(512), 0, 0, out, cyc, 200);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<V>, dim3(256), dim3(512), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-58s exp %2d cvt %2d swap %2d: %7.1f ns per tile %7.1f ticks per tile\n", name, ne, nc, ns, ms * 1e6 / iters, (double)c / iters);
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<6>("MFMAs alone");
    run<5>("product before round 6 (8 swap-slot ops after the first PV MFMA)");
    run<0>("product now (swaps after the second PV MFMA)");
    run<1>("3 exp under every big MFMA, rest under the small ones");
    run<2>("3 exp + 1 cvt under every big MFMA, swaps spread");
    run<3>("2 exp + 1 cvt almost everywhere, swaps spread");
    run<4>("everything under the big MFMAs, small ones bare");
    run<7>("2 exp + 1 cvt under big, 3 exp under small");
    return 0;
}
