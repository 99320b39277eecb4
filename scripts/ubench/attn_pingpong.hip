// Schedule microbenchmark for the D = 40 attention loop (round 6): the per-tile work of one k_attn5 wave -- 10 v_mfma_f32_32x32x16_bf16 + 4 v_mfma_f32_16x16x32_bf16,
// 32 v_exp_f32 + 16 v_cvt_pk_bf16_f32 + 8 v_permlane16_swap, with the REAL data dependencies (S -> exp -> P -> P V) and no memory traffic -- in two schedules:
//   MODE 0 "interleaved" (the product kernel's): every wave runs  C [S0 = 3 MFMA + units of S1'] D [O1 += 4 MFMA + units of S0] F [S1 = 3 MFMA + units of S0]
//          G [O0 += 4 MFMA + units of S1], one s_barrier per tile; the two waves of a SIMD run in lockstep.
//   MODE 1 "ping-pong": segment X = the tile's 14 MFMAs back to back (S0, S1 of tile i; both P V of tile i - 1), segment Y = its 56 VALU (P of tile i), a barrier after
//          each; waves 0..3 run X Y X Y ..., waves 4..7 (the SIMD partners) Y X Y X ...: one wave's matrix segment beside the other's vector segment.
//   MODE 2: segments as in 1 but both halves in phase (X X / Y Y): what the phase shift itself buys.
// Prints cycles per tile (per SIMD = per wave pair).  build: hipcc --offload-arch=gfx950 -O3 -o attn_pingpong attn_pingpong.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

__device__ __forceinline__ f32x16 mm32(uint4 a, uint4 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mm16(uint4 a, uint4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0); }
__device__ __forceinline__ unsigned unit(float x0, float x1)
{
    unsigned r;
    float e0, e1;
    asm volatile("v_exp_f32 %0, %1" : "=v"(e0) : "v"(x0));
    asm volatile("v_exp_f32 %0, %1" : "=v"(e1) : "v"(x1));
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(e0), "v"(e1));
    return r;
}
__device__ __forceinline__ void units(const f32x16 &S, uint4 &p0, uint4 &p1, int lo, int hi)      // units lo .. hi-1 of 8: unit u = registers 2u, 2u + 1
{
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        if (u < lo || u >= hi) continue;
        const unsigned v = unit(S[2 * u], S[2 * u + 1]);
        uint4 &p = u < 4 ? p0 : p1;
        if ((u & 3) == 0) p.x = v; else if ((u & 3) == 1) p.y = v; else if ((u & 3) == 2) p.z = v; else p.w = v;
    }
}
__device__ __forceinline__ void p16(uint4 &p0, uint4 &p1)      // in place
{
    const auto x = __builtin_amdgcn_permlane16_swap(p0.x, p1.x, false, false), y = __builtin_amdgcn_permlane16_swap(p0.y, p1.y, false, false);
    const auto z = __builtin_amdgcn_permlane16_swap(p0.z, p1.z, false, false), w = __builtin_amdgcn_permlane16_swap(p0.w, p1.w, false, false);
    p0 = make_uint4(x[0], y[0], z[0], w[0]); p1 = make_uint4(x[1], y[1], z[1], w[1]);
}
#define SB __builtin_amdgcn_sched_barrier(0)

template <int MODE>
__global__ __launch_bounds__(512, 1) void k(float *out, long long *cyc, int iters, int late_mask)
{
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint4 kf[3], qf[2][3], vf[2], vf16;
    for (int i = 0; i < 3; ++i) { kf[i] = make_uint4(0x3c003c00u + threadIdx.x, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u + i); qf[0][i] = kf[i]; qf[1][i] = make_uint4(kf[i].y, kf[i].x, kf[i].w, kf[i].z); }
    vf[0] = kf[0]; vf[1] = kf[1]; vf16 = kf[2];
    f32x16 S0, S1, os0, os1, z16;
    for (int r = 0; r < 16; ++r) { S0[r] = -1.f; S1[r] = -2.f; os0[r] = 0.f; os1[r] = 0.f; z16[r] = 0.f; }
    f32x4 o1[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) o1[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    uint4 pf[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) pf[a][b] = make_uint4(0, 0, 0, 0);
    auto X = [&]() __attribute__((always_inline)) {        // 14 MFMAs: S0, S1 of this tile (two interleaved 3-chains), both P V of the previous tile
        S0 = mm32(kf[0], qf[0][0], z16); S1 = mm32(kf[0], qf[1][0], z16); SB;
        S0 = mm32(kf[1], qf[0][1], S0); S1 = mm32(kf[1], qf[1][1], S1); SB;
        S0 = mm32(kf[2], qf[0][2], S0); S1 = mm32(kf[2], qf[1][2], S1); SB;
        os0 = mm32(vf[0], pf[0][0], os0); os1 = mm32(vf[0], pf[1][0], os1); SB;
        os0 = mm32(vf[1], pf[0][1], os0); os1 = mm32(vf[1], pf[1][1], os1); SB;
    };
    auto X16 = [&]() __attribute__((always_inline)) {      // the four 16x16x32 MFMAs on the swapped P (after the 32x32x16 ones have read P)
        p16(pf[0][0], pf[0][1]); p16(pf[1][0], pf[1][1]); SB;
        o1[0][0] = mm16(vf16, pf[0][0], o1[0][0]); o1[0][1] = mm16(vf16, pf[0][1], o1[0][1]);
        o1[1][0] = mm16(vf16, pf[1][0], o1[1][0]); o1[1][1] = mm16(vf16, pf[1][1], o1[1][1]); SB;
    };
    auto Y = [&]() __attribute__((always_inline)) {        // 48 VALU: P of this tile from S0, S1
        units(S0, pf[0][0], pf[0][1], 0, 8); units(S1, pf[1][0], pf[1][1], 0, 8); SB;
    };
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    if (MODE == 3 || MODE == 4) {
        // MODE 3: the interleaved instruction stream with the units DETACHED from the MFMAs (they read private constants and write private sinks): the pure issue
        // bound of the mix.  MODE 4: the MFMAs alone in that order (one barrier per tile).
        f32x16 Sc0 = S0, Sc1 = S1;
        uint4 sk[2][2] = {{pf[0][0], pf[0][1]}, {pf[1][0], pf[1][1]}};
        for (int it = 0; it < iters; ++it) {
            __builtin_amdgcn_s_barrier();
#define U(Sx, a, b, lo, hi) if (MODE == 3) units(Sx, a, b, lo, hi)
            S0 = mm32(kf[0], qf[0][0], z16); U(Sc1, sk[1][0], sk[1][1], 4, 5); SB;
            S0 = mm32(kf[1], qf[0][1], S0); U(Sc1, sk[1][0], sk[1][1], 5, 6); SB;
            S0 = mm32(kf[2], qf[0][2], S0); U(Sc1, sk[1][0], sk[1][1], 6, 8); SB;
            os1 = mm32(vf[0], pf[1][0], os1); SB;
            os1 = mm32(vf[1], pf[1][1], os1); if (MODE == 3) p16(sk[1][0], sk[1][1]); U(Sc0, sk[0][0], sk[0][1], 0, 1); SB;
            o1[1][0] = mm16(vf16, pf[1][0], o1[1][0]); U(Sc0, sk[0][0], sk[0][1], 1, 2); SB;
            o1[1][1] = mm16(vf16, pf[1][1], o1[1][1]); U(Sc0, sk[0][0], sk[0][1], 2, 3); SB;
            S1 = mm32(kf[0], qf[1][0], z16); U(Sc0, sk[0][0], sk[0][1], 3, 5); SB;
            S1 = mm32(kf[1], qf[1][1], S1); U(Sc0, sk[0][0], sk[0][1], 5, 7); SB;
            S1 = mm32(kf[2], qf[1][2], S1); U(Sc0, sk[0][0], sk[0][1], 7, 8); SB;
            os0 = mm32(vf[0], pf[0][0], os0); SB;
            os0 = mm32(vf[1], pf[0][1], os0); if (MODE == 3) p16(sk[0][0], sk[0][1]); U(Sc1, sk[1][0], sk[1][1], 0, 2); SB;
            o1[0][0] = mm16(vf16, pf[0][0], o1[0][0]); U(Sc1, sk[1][0], sk[1][1], 2, 3); SB;
            o1[0][1] = mm16(vf16, pf[0][1], o1[0][1]); U(Sc1, sk[1][0], sk[1][1], 3, 4); SB;
#undef U
        }
        S0[0] += __uint_as_float(sk[0][0].x ^ sk[0][1].y ^ sk[1][0].z ^ sk[1][1].w) + Sc0[3] + Sc1[5];
    } else if (MODE == 5) {
        // the VALU of a tile alone (32 v_exp + 16 v_cvt_pk + 8 lane swaps per wave), one barrier per tile
        for (int it = 0; it < iters; ++it) {
            __builtin_amdgcn_s_barrier();
            units(S0, pf[0][0], pf[0][1], 0, 8); units(S1, pf[1][0], pf[1][1], 0, 8); SB;
            p16(pf[0][0], pf[0][1]); p16(pf[1][0], pf[1][1]); SB;
        }
    } else if (MODE == 0) {
        for (int it = 0; it < iters; ++it) {
            __builtin_amdgcn_s_barrier();
            // C
            S0 = mm32(kf[0], qf[0][0], z16); units(S1, pf[1][0], pf[1][1], 4, 5); SB;
            S0 = mm32(kf[1], qf[0][1], S0); units(S1, pf[1][0], pf[1][1], 5, 6); SB;
            S0 = mm32(kf[2], qf[0][2], S0); units(S1, pf[1][0], pf[1][1], 6, 8); SB;
            // D
            os1 = mm32(vf[0], pf[1][0], os1); SB;
            os1 = mm32(vf[1], pf[1][1], os1); p16(pf[1][0], pf[1][1]); units(S0, pf[0][0], pf[0][1], 0, 1); SB;
            o1[1][0] = mm16(vf16, pf[1][0], o1[1][0]); units(S0, pf[0][0], pf[0][1], 1, 2); SB;
            o1[1][1] = mm16(vf16, pf[1][1], o1[1][1]); units(S0, pf[0][0], pf[0][1], 2, 3); SB;
            // F
            S1 = mm32(kf[0], qf[1][0], z16); units(S0, pf[0][0], pf[0][1], 3, 5); SB;
            S1 = mm32(kf[1], qf[1][1], S1); units(S0, pf[0][0], pf[0][1], 5, 7); SB;
            S1 = mm32(kf[2], qf[1][2], S1); units(S0, pf[0][0], pf[0][1], 7, 8); SB;
            // G
            os0 = mm32(vf[0], pf[0][0], os0); SB;
            os0 = mm32(vf[1], pf[0][1], os0); p16(pf[0][0], pf[0][1]); units(S1, pf[1][0], pf[1][1], 0, 2); SB;
            o1[0][0] = mm16(vf16, pf[0][0], o1[0][0]); units(S1, pf[1][0], pf[1][1], 2, 3); SB;
            o1[0][1] = mm16(vf16, pf[0][1], o1[0][1]); units(S1, pf[1][0], pf[1][1], 3, 4); SB;
        }
    } else {
        const bool late = MODE == 1 && ((late_mask >> wid) & 1);
        for (int it = 0; it < iters; ++it) {
            __builtin_amdgcn_s_barrier();
            if (late) { Y(); } else { X(); X16(); }
            __builtin_amdgcn_s_barrier();
            if (late) { X(); X16(); } else { Y(); }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += S0[r] + S1[r] + os0[r] + os1[r];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) s += o1[a][b][0] + o1[a][b][1] + o1[a][b][2] + o1[a][b][3] + __uint_as_float(pf[a][b].x);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE> void run(const char *name, int blocks, int late_mask = 0xf0)
{
    float *out; long long *cyc;
    hipMalloc(&out, sizeof(float) * 512 * blocks); hipMalloc(&cyc, 8);
    const int iters = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, out, cyc, 200, late_mask);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, out, cyc, iters, late_mask);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    // per tile and SIMD: 2 waves x (10 x 32 + 4 x 16) = 768 cycles of matrix pipe
    printf("%-44s blocks %3d: %8.1f ns per tile, %7.1f shader-clock ticks per tile (matrix pipe needs 768 cycles per tile and SIMD)\n", name, blocks, ms * 1e6 / iters, (double)c / iters);
    hipFree(out); hipFree(cyc);
}

int main()
{
    for (int blocks : {1, 256}) {
        run<0>("interleaved, lockstep (product schedule)", blocks);
        run<2>("segments X | Y, both halves in phase", blocks);
        run<3>("interleaved order, units detached from the MFMAs", blocks);
        run<4>("the 14 MFMAs alone (+ barrier)", blocks);
        run<5>("the 56 VALU alone (+ barrier)", blocks);
        run<1>("ping-pong, late waves 4..7 (mask f0)", blocks, 0xf0);
        run<1>("ping-pong, late waves 1,3,5,7 (mask aa)", blocks, 0xaa);
        run<1>("ping-pong, late waves 2,3,6,7 (mask cc)", blocks, 0xcc);
        run<1>("ping-pong, late waves 1,2,4,7 (mask 96)", blocks, 0x96);
        run<1>("ping-pong mask 00 (= in phase, branchy code)", blocks, 0x00);
    }
    return 0;
}
