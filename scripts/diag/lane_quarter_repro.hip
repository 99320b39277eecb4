// Stand-alone victims for the "lanes 48..63 read wrong inputs" fault (scripts/corun_bisect.py): kernels that only LOAD the Gaussian record the
// way k_project_sh_fwd does and ECHO what they loaded.  The expected output is a pure copy of the inputs, so any deviation is a wrong load.
//   MODE 0: five early loads, no LDS                       MODE 1: + the cooperative float4 staging of features_rest through LDS (as the product)
//   MODE 2: staging first, the five loads after it          MODE 3: as 1, plus transcendental work (exp / sqrt / divide) before the echo
//   MODE 4: as 0 but with the 46 KB LDS allocation only (never touched)
//   MODE 5: loads + 64 dependent PACKED fp32 FMAs (v_pk_fma_f32, inline asm) per lane     MODE 6: the same arithmetic as 128 scalar v_fma_f32
//   (5 and 6 give the same bits; only 5 deviates beside another process's MFMA kernels: profiles/r04_packed_fp32_fault.txt)
// hipcc --offload-arch=gfx950 -O3 -shared -fPIC scripts/diag/lane_quarter_repro.hip -o scripts/diag/lane_quarter_repro.so
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int MODE>
__global__ __launch_bounds__(256) void k_echo(int64_t N, const float *__restrict__ means, const float *__restrict__ log_scales,
                                              const float *__restrict__ quats, const float *__restrict__ op_logit,
                                              const float *__restrict__ f_dc, const float *__restrict__ f_rest, float *__restrict__ out)
{
    constexpr int R = 45;
    __shared__ __attribute__((aligned(16))) float srest[(MODE == 0) ? 4 : 256 * R];
    const int tid = threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.x * 256;
    const int64_t i = i0 + tid;
    const int64_t ic = i < N ? i : N - 1;
    float p0, p1, p2, l0, l1, l2, opl, d0, d1, d2;
    float4 q;
    auto loads = [&]() {
        p0 = means[3 * ic]; p1 = means[3 * ic + 1]; p2 = means[3 * ic + 2];
        l0 = log_scales[3 * ic]; l1 = log_scales[3 * ic + 1]; l2 = log_scales[3 * ic + 2];
        q = *reinterpret_cast<const float4 *>(quats + 4 * ic);
        opl = op_logit[ic];
        d0 = f_dc[3 * ic]; d1 = f_dc[3 * ic + 1]; d2 = f_dc[3 * ic + 2];
    };
    auto stage = [&]() {
        const int64_t cnt = ((N - i0 < 256 ? N - i0 : 256)) * R;
        const float *src = f_rest + i0 * R;
        for (int64_t j = tid; j < cnt / 4; j += 256) reinterpret_cast<float4 *>(srest)[j] = reinterpret_cast<const float4 *>(src)[j];
        for (int64_t j = (cnt / 4) * 4 + tid; j < cnt; j += 256) srest[j] = src[j];
    };
    if (MODE == 2) { stage(); loads(); }
    else { loads(); if (MODE == 1 || MODE == 3) stage(); }
    if (MODE == 4 && N < 0) srest[tid] = 1.f;        // (keeps the allocation)
    float extra = 0.f;
    if (MODE == 3 && i < N) {
        float s0 = expf(l0), s1 = expf(l1), s2 = expf(l2);
        float qn = sqrtf(((q.x * q.x + q.y * q.y) + q.z * q.z) + q.w * q.w);
        extra = (s0 + s1 + s2) / qn + 1.f / (1.f + __expf(-opl));
    }
    if (MODE == 1 || MODE == 2 || MODE == 3) __syncthreads();
    if (i >= N) return;
    float rs = 0.f;
    if (MODE == 1 || MODE == 2 || MODE == 3) {
        const float *r = srest + tid * R;
#pragma unroll
        for (int k = 0; k < R; ++k) rs += r[k];
    }
    float *o = out + i * 16;
    o[0] = p0; o[1] = p1; o[2] = p2; o[3] = l0; o[4] = l1; o[5] = l2; o[6] = q.x; o[7] = q.y; o[8] = q.z; o[9] = q.w;
    o[10] = opl; o[11] = d0; o[12] = d1; o[13] = d2; o[14] = rs; o[15] = extra;
}

typedef float float2v __attribute__((ext_vector_type(2)));

template <bool PACKED>
__global__ __launch_bounds__(256) void k_fma(int64_t N, const float *__restrict__ quats, float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float4 q = *reinterpret_cast<const float4 *>(quats + 4 * i);
    float2v acc = {q.x, q.y}, a = {0.75f + 0.01f * q.z, 0.5f - 0.01f * q.w}, b = {q.w * 0.125f, q.z * 0.25f};
#pragma unroll
    for (int k = 0; k < 64; ++k) {
        if (PACKED) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
        else {
            float x = acc.x, y = acc.y;
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a.x), "v"(b.x));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y) : "v"(a.y), "v"(b.y));
            acc.x = x; acc.y = y;
        }
    }
    float *o = out + i * 16;
#pragma unroll
    for (int k = 0; k < 16; ++k) o[k] = 0.f;
    o[6] = q.x; o[7] = q.y; o[8] = q.z; o[9] = q.w; o[14] = acc.x; o[15] = acc.y;
}

extern "C" int diag_echo(int mode, int64_t N, const float *means, const float *ls, const float *quats, const float *opl, const float *dc,
                         const float *rest, float *out, void *stream)
{
    dim3 g((unsigned)((N + 255) / 256)), b(256);
    hipStream_t s = (hipStream_t)stream;
    switch (mode) {
    case 0: hipLaunchKernelGGL(k_echo<0>, g, b, 0, s, N, means, ls, quats, opl, dc, rest, out); break;
    case 1: hipLaunchKernelGGL(k_echo<1>, g, b, 0, s, N, means, ls, quats, opl, dc, rest, out); break;
    case 2: hipLaunchKernelGGL(k_echo<2>, g, b, 0, s, N, means, ls, quats, opl, dc, rest, out); break;
    case 3: hipLaunchKernelGGL(k_echo<3>, g, b, 0, s, N, means, ls, quats, opl, dc, rest, out); break;
    case 4: hipLaunchKernelGGL(k_echo<4>, g, b, 0, s, N, means, ls, quats, opl, dc, rest, out); break;
    case 5: hipLaunchKernelGGL(k_fma<true>, g, b, 0, s, N, quats, out); break;
    case 6: hipLaunchKernelGGL(k_fma<false>, g, b, 0, s, N, quats, out); break;
    default: return 1;
    }
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
