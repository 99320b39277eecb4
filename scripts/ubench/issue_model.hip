// Issue-model microbenchmark for gfx950: how do v_mfma_f32_16x16x32_bf16 and VALU instructions of ONE wave (and of two
// co-resident waves of a SIMD) overlap?  Prints cycles per MFMA group for several fillers.
// build: hipcc --offload-arch=gfx950 -O3 -o issue_model issue_model.hip ; run: ./issue_model
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NM, int NV, int KIND>
__global__ __launch_bounds__(1024) void k(float *out, long long *cyc, int iters)
{
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i * 3 + 1); }
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.001f + i;
    typedef __attribute__((ext_vector_type(16))) float f32x16;
    f32x16 big[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) big[i][j] = 0.f;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (NM == 1) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[g], 0, 0, 0);
            if (NM == 3) big[g & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, big[g & 3], 0, 0, 0);
            if (NM == 2) {   // 16x16x16 (4 bf16 per lane)
                typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
                bf16x4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
                acc[g] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(__attribute__((ext_vector_type(4))) short, a4), __builtin_bit_cast(__attribute__((ext_vector_type(4))) short, b4), acc[g], 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int r = (g * NV + j) & 15;
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[r]) : "v"(v[(r + 1) & 15]));
                if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[r]));
                if (KIND == 5) asm volatile("v_exp_legacy_f32 %0, %0" : "+v"(v[r]));
                if (KIND == 6) asm volatile("v_exp_f16 %0, %0" : "+v"(v[r]));
                if (KIND == 7) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(v[(r + 1) & 15]), "v"(v[(r + 2) & 15]));
                if (KIND == 8) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(v[r]) : "v"(v[(r + 1) & 15]));
                if (KIND == 2) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(v[(r + 1) & 15]), "v"(v[(r + 2) & 15]));
                if (KIND == 3) { unsigned t; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(t) : "v"(v[r]), "v"(v[(r + 1) & 15])); v[r] = __uint_as_float(t); }
                if (KIND == 9) {   // the attention mix: exp, exp, cvt_pk, exp, exp, cvt_pk ...
                    if (j % 3 == 2) { unsigned t; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(t) : "v"(v[r]), "v"(v[(r + 1) & 15])); v[(r + 2) & 15] = __uint_as_float(t); }
                    else asm volatile("v_exp_f32 %0, %0" : "+v"(v[r]));
                }
                if (KIND == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double *)&v[(2 * r) & 14]) : "v"(*(double *)&v[(2 * r + 2) & 14]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += big[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NM, int NV, int KIND>
void run(const char *name, int threads)
{
    float *out; long long *cyc;
    hipMalloc(&out, 4 * 512 * 256); hipMalloc(&cyc, 8);
    const int iters = 200000;
    k<NM, NV, KIND><<<1, threads>>>(out, cyc, 2000);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<NM, NV, KIND><<<1, threads>>>(out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-20s waves/SIMD=%d : %7.2f ticks, %7.2f ns per group (%d MFMA + %d VALU)\n", name, threads / 256, (double)h / (iters * 8.0), ms * 1e6 / (iters * 8.0), NM, NV);
    hipFree(out); hipFree(cyc);
}

int main(int argc, char **argv)
{
    if (argc > 1) {      // round 2: 32x32x16 vs 16x16x32 beside the attention filler mix (2 v_exp_f32 + 1 v_cvt_pk per 3 fillers)
        for (int th : {256, 512}) {
            run<1, 0, 0>("16x16x32 only", th);
            run<3, 0, 0>("32x32x16 only", th);
            run<1, 3, 9>("16x16x32 + 3 mix", th);
            run<1, 2, 1>("16x16x32 + 2 exp", th);
            run<1, 1, 1>("16x16x32 + 1 exp", th);
            run<3, 2, 1>("32x32x16 + 2 exp", th);
            run<3, 3, 1>("32x32x16 + 3 exp", th);
            run<3, 4, 1>("32x32x16 + 4 exp", th);
            run<3, 3, 9>("32x32x16 + 3 mix", th);
            run<3, 6, 9>("32x32x16 + 6 mix", th);
            run<3, 9, 9>("32x32x16 + 9 mix", th);
            run<3, 4, 0>("32x32x16 + 4 fma", th);
            run<3, 6, 0>("32x32x16 + 6 fma", th);
            run<3, 8, 0>("32x32x16 + 8 fma", th);
            run<3, 4, 3>("32x32x16 + 4 cvt_pk", th);
        }
        return 0;
    }
    // s_memtime counts at 100 MHz on gfx9?  report raw counter units; calibrate with the MFMA-only line (16 clk expected)
    for (int th : {256, 512, 1024}) {
        run<1, 0, 0>("mfma only", th);
        run<2, 0, 0>("mfma 16x16x16 only", th);
        run<2, 2, 1>("mfma16 + 2 exp", th);
        run<0, 4, 0>("4 fma only", th);
        run<0, 4, 1>("4 exp only", th);
        run<0, 4, 5>("4 exp_legacy only", th);
        run<0, 4, 6>("4 exp_f16 only", th);
        run<0, 4, 7>("4 perm_b32 only", th);
        run<0, 4, 8>("4 ldexp only", th);
        run<1, 2, 5>("mfma + 2 exp_legacy", th);
        run<1, 2, 6>("mfma + 2 exp_f16", th);
        run<0, 4, 2>("4 max3 only", th);
        run<0, 4, 3>("4 cvt_pk only", th);
        run<0, 4, 4>("4 pk_mul only", th);
        run<1, 2, 0>("mfma + 2 fma", th);
        run<1, 3, 0>("mfma + 3 fma", th);
        run<1, 4, 0>("mfma + 4 fma", th);
        run<1, 6, 0>("mfma + 6 fma", th);
        run<1, 8, 0>("mfma + 8 fma", th);
        run<1, 2, 1>("mfma + 2 exp", th);
        run<1, 4, 1>("mfma + 4 exp", th);
        run<1, 4, 2>("mfma + 4 max3", th);
        run<1, 4, 3>("mfma + 4 cvt_pk", th);
        run<1, 4, 4>("mfma + 4 pk_mul", th);
    }
    return 0;
}
