// wstream.hip -- how fast can EVERY workgroup of a row-resident fused transformer-tail kernel stream the block's weights?
// (the bound the persistent-kernel analysis rests on.)  G workgroups of 512 threads each pull the same W bytes
// (1.64 M bf16 weights = 3.28 MB at C = 320) from global memory / L2 into a 3-slot LDS ring with global_load_lds_dwordx4, 16 KB per
// slot (8 waves x 2 x 1 KB), touching each slot once with ds_read so that the data really lands.  No MFMA: pure operand delivery.
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/wstream scripts/ubench/wstream.hip ; run: scripts/ubench/wstream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ void glds16(const void *sbase, unsigned voff, unsigned lds_dst)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

template <int SLOT_KB>
__global__ __launch_bounds__(512, 1) void k_stream(const unsigned char *w, size_t bytes, float *sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int SLOT = SLOT_KB * 1024, NS = 3, PER_WAVE = SLOT / 8, NI = PER_WAVE / 1024;      // DMA instructions per wave per slot
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t nslots = bytes / SLOT;
    // rotate the start per workgroup so that the 384 workgroups do not all hit the same L2 lines at once
    const size_t rot = (blockIdx.x * 7) % nslots;
    auto issue = [&](size_t s) {
        const unsigned char *src = w + ((s + rot) % nslots) * SLOT + wid * PER_WAVE;
#pragma unroll
        for (int i = 0; i < NI; ++i) glds16(src + i * 1024, lane * 16, lds0 + (unsigned)((s % NS) * SLOT + wid * PER_WAVE + i * 1024));
    };
    float acc = 0.f;
    issue(0); issue(1);
    for (size_t s = 0; s < nslots; ++s) {
        if (s + 2 < nslots) { issue(s + 2); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI) : "memory"); }
        else if (s + 1 < nslots) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const float4 v = *reinterpret_cast<const float4 *>(smem + (s % NS) * SLOT + tid * 16);
        acc += v.x + v.w;
        __builtin_amdgcn_s_barrier();
    }
    if (acc == 12345.678f) sink[0] = acc;
}

int main()
{
    const size_t bytes = (size_t)1640000 * 2 / 16384 * 16384;      // weights of one SD1.5 level-0 transformer tail (C = 320), bf16
    unsigned char *w; float *sink;
    hipMalloc(&w, bytes); hipMalloc(&sink, 4);
    hipMemset(w, 1, bytes);
    hipFuncSetAttribute((const void *)k_stream<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 16 * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int G : {96, 192, 256, 384, 768}) {
        for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k_stream<16>, dim3(G), dim3(512), 3 * 16 * 1024, 0, w, bytes, sink);
        hipEventRecord(a);
        const int n = 10;
        for (int it = 0; it < n; ++it) hipLaunchKernelGGL(k_stream<16>, dim3(G), dim3(512), 3 * 16 * 1024, 0, w, bytes, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double us = ms * 1e3 / n, tb = (double)G * bytes / (us * 1e-6) / 1e12;
        printf("G = %4d workgroups x %.2f MB of weights: %8.1f us per launch   %6.2f TB/s into LDS   (rows per workgroup for 24576 rows: %d)\n",
               G, bytes / 1e6, us, tb, 24576 / G);
    }
    return 0;
}
