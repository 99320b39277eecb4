// mx_probe.hip -- operand-layout probe of v_mfma_scale_f32_16x16x128_f8f6f4 (MX-scaled fp8 MFMA, gfx950): one wave, raw per-lane
// operand registers in, raw accumulator registers out; scripts/mx_probe.py packs test matrices under candidate layouts and compares
// with a host matmul.  build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o scripts/ubench/libmx_probe.so scripts/ubench/mx_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int OPA, int OPB>
__global__ void k_probe(const int *A, const int *B, const int *SA, const int *SB, float *D)
{
    const int lane = threadIdx.x;
    i32x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = A[lane * 8 + i]; b[i] = B[lane * 8 + i]; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, OPA, SA[lane], OPB, SB[lane]);
    for (int i = 0; i < 4; ++i) D[lane * 4 + i] = c[i];
}

extern "C" int mx_probe(const void *A, const void *B, const void *SA, const void *SB, void *D, int opa, int opb, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
#define L(X, Y) hipLaunchKernelGGL((k_probe<X, Y>), dim3(1), dim3(64), 0, s, (const int *)A, (const int *)B, (const int *)SA, (const int *)SB, (float *)D)
    if (opa == 0 && opb == 0) L(0, 0); else if (opa == 1 && opb == 0) L(1, 0); else if (opa == 2 && opb == 3) L(2, 3); else if (opa == 3 && opb == 1) L(3, 1); else return -1;
#undef L
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
