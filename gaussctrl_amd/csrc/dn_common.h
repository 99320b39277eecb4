// dn_common.h -- element types and small device helpers shared by the denoise kernels (gfx950).
#pragma once
#include "common.h"

namespace dn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// dtype codes of the C ABI
enum { DT_BF16 = 0, DT_F16 = 1 };

struct BF16 {
    using elem = __bf16;
    using vec8 = bf16x8;
    static __device__ __forceinline__ f32x4 mfma(uint4 a, uint4 b, f32x4 c)
    {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16 mfma32(uint4 a, uint4 b, f32x16 c)      // v_mfma_f32_32x32x16_bf16
    {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ float to_f(unsigned short u) { return __uint_as_float((unsigned)u << 16); }
    static __device__ __forceinline__ unsigned short from_f(float f)
    {
        __bf16 h = (__bf16)f;
        return __builtin_bit_cast(unsigned short, h);
    }
};

struct F16 {
    using elem = _Float16;
    using vec8 = f16x8;
    static __device__ __forceinline__ f32x4 mfma(uint4 a, uint4 b, f32x4 c)
    {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16 mfma32(uint4 a, uint4 b, f32x16 c)      // v_mfma_f32_32x32x16_f16
    {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ float to_f(unsigned short u) { return (float)__builtin_bit_cast(_Float16, u); }
    static __device__ __forceinline__ unsigned short from_f(float f)
    {
        _Float16 h = (_Float16)f;
        return __builtin_bit_cast(unsigned short, h);
    }
};

template <class T>
__device__ __forceinline__ unsigned pack2(float a, float b)
{
    return (unsigned)T::from_f(a) | ((unsigned)T::from_f(b) << 16);
}
// one v_cvt_pk_bf16_f32 per pair: a 2-vector conversion lets hipcc emit the packed instruction AND pad the
// VALU-write -> MFMA-read hazard itself (an inline-asm cvt_pk feeding an MFMA operand read stale registers)
template <>
__device__ __forceinline__ unsigned pack2<BF16>(float a, float b)
{
    typedef __attribute__((ext_vector_type(2))) float f2;
    typedef __attribute__((ext_vector_type(2))) __bf16 b2;
    const b2 h = __builtin_convertvector(f2{a, b}, b2);
    return __builtin_bit_cast(unsigned, h);
}
template <>
__device__ __forceinline__ unsigned pack2<F16>(float a, float b)
{
    typedef __attribute__((ext_vector_type(2))) float f2;
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    const h2 h = __builtin_convertvector(f2{a, b}, h2);
    return __builtin_bit_cast(unsigned, h);
}

template <class T>
__device__ __forceinline__ void unpack8(uint4 v, float *f)
{
    f[0] = T::to_f((unsigned short)(v.x & 0xffff)); f[1] = T::to_f((unsigned short)(v.x >> 16));
    f[2] = T::to_f((unsigned short)(v.y & 0xffff)); f[3] = T::to_f((unsigned short)(v.y >> 16));
    f[4] = T::to_f((unsigned short)(v.z & 0xffff)); f[5] = T::to_f((unsigned short)(v.z >> 16));
    f[6] = T::to_f((unsigned short)(v.w & 0xffff)); f[7] = T::to_f((unsigned short)(v.w >> 16));
}

template <class T>
__device__ __forceinline__ uint4 pack8(const float *f)
{
    return make_uint4(pack2<T>(f[0], f[1]), pack2<T>(f[2], f[3]), pack2<T>(f[4], f[5]), pack2<T>(f[6], f[7]));
}

__device__ __forceinline__ float silu(float x) { return x / (1.f + __expf(-x)); }
// exact (erf) GELU of torch.nn.functional.gelu; erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, i.e. fp32 round-off
// level) in ~12 VALU ops instead of the ~35 of the device-library erff -- this sits in the epilogue of the largest GEMMs
__device__ __forceinline__ float erf_as(float x)
{
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * ax);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float r = 1.f - poly * __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
    return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erf_as(x * 0.70710678118654752440f)); }

__device__ __forceinline__ float wave_sum_f(float v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

}  // namespace dn
