// MFMA block microbenchmark: the register pattern of k_gemm8 (NTW x MT accumulators, NTW + MT operand fragments), no memory.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_block mfma_block.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NTW, int MT, int ORDER>
__global__ __launch_bounds__(512) void k(float *out, int iters)
{
    f32x4 acc[NTW][MT];
    bf16x8 w[NTW], a[MT];
    for (int i = 0; i < NTW; ++i) for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < NTW; ++i) for (int e = 0; e < 8; ++e) w[i][e] = (__bf16)(float)(threadIdx.x + i + e);
    for (int i = 0; i < MT; ++i) for (int e = 0; e < 8; ++e) a[i][e] = (__bf16)(float)(i * 3 + e);
    for (int it = 0; it < iters; ++it) {
        if (ORDER == 0) {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[nt], a[mt], acc[nt][mt], 0, 0, 0);
        } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[nt], a[mt], acc[nt][mt], 0, 0, 0);
        }
        // keep the operands "live and changing" without VALU work: nothing (registers are loop-invariant)
        asm volatile("" ::: "memory");
    }
    float s = 0;
    for (int i = 0; i < NTW; ++i) for (int j = 0; j < MT; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NT2, int MT2>
__global__ __launch_bounds__(512) void k32(float *out, int iters)
{
    f32x16 acc[NT2][MT2];
    bf16x8 w[NT2], a[MT2];
    for (int i = 0; i < NT2; ++i) for (int j = 0; j < MT2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int i = 0; i < NT2; ++i) for (int e = 0; e < 8; ++e) w[i][e] = (__bf16)(float)(threadIdx.x + i + e);
    for (int i = 0; i < MT2; ++i) for (int e = 0; e < 8; ++e) a[i][e] = (__bf16)(float)(i * 3 + e);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[nt], a[mt], acc[nt][mt], 0, 0, 0);
        asm volatile("" ::: "memory");
    }
    float s = 0;
    for (int i = 0; i < NT2; ++i) for (int j = 0; j < MT2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F> void timeit(const char *name, F launch, double flop_per_iter_per_wave, int waves, int blocks)
{
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(200);
    hipEventRecord(e0); launch(iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tf = flop_per_iter_per_wave * iters * waves * blocks / (ms * 1e-3) / 1e12;
    printf("%-44s blocks=%3d waves/SIMD=%d : %8.3f ms  %8.1f TF/s  (%.1f%% of 2.5 PF if all 256 CUs did the same: %.0f TF)\n", name, blocks, waves / 4, ms, tf,
           100.0 * tf * 256 / blocks / 2500.0, tf * 256 / blocks);
}

int main()
{
    float *out; hipMalloc(&out, 4 * 512 * 256);
    for (int blocks : {1, 256}) {
        for (int th : {256, 512}) {
            timeit("16x16x32 5x3 nt-major", [&](int it) { k<5, 3, 0><<<blocks, th>>>(out, it); }, 15 * 16384.0, th / 64, blocks);
            timeit("16x16x32 5x3 mt-major", [&](int it) { k<5, 3, 1><<<blocks, th>>>(out, it); }, 15 * 16384.0, th / 64, blocks);
            timeit("16x16x32 4x4", [&](int it) { k<4, 4, 0><<<blocks, th>>>(out, it); }, 16 * 16384.0, th / 64, blocks);
            timeit("16x16x32 5x4", [&](int it) { k<5, 4, 0><<<blocks, th>>>(out, it); }, 20 * 16384.0, th / 64, blocks);
            timeit("32x32x16 2x2", [&](int it) { k32<2, 2><<<blocks, th>>>(out, it); }, 4 * 32768.0, th / 64, blocks);
        }
    }
    return 0;
}
