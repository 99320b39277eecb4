// dn_gemm_fp8.hip -- instantiates k_gemm8q: OCP fp8 (e4m3) operands on the block-scaled MFMA (BASELINE configs[3]).
#include "dn_gemm_kernels.h"

namespace {
template <class T, int MODE, int NTW, int MT>
void launch8q(const GemmArgs &g, dim3 grid, hipStream_t s)
{
    constexpr size_t lds = 3 * (64 * MT * 128 + 32 * NTW * 128) + 64 * MT * 8;
    static_assert(lds <= 160 * 1024, "LDS ring");
    static gc::AttrOnce once[2];
    if (fuse_of(g)) {
        gc::ensure_dynamic_lds(once[1], (const void *)k_gemm8q<T, MODE, NTW, MT, true>, (int)lds);
        hipLaunchKernelGGL((k_gemm8q<T, MODE, NTW, MT, true>), grid, dim3(512), lds, s, g);
    } else {
        gc::ensure_dynamic_lds(once[0], (const void *)k_gemm8q<T, MODE, NTW, MT, false>, (int)lds);
        hipLaunchKernelGGL((k_gemm8q<T, MODE, NTW, MT, false>), grid, dim3(512), lds, s, g);
    }
}
// channel-partial epilogue (GemmArgs::chan_parts): fast convs whose output feeds a GroupNorm
template <class T, int NTW, int MT>
void launch8q_cs(const GemmArgs &g, dim3 grid, hipStream_t s)
{
    constexpr size_t lds = 3 * (64 * MT * 128 + 32 * NTW * 128) + 32 * NTW * 16;
    static_assert(lds <= 160 * 1024, "LDS ring");
    static gc::AttrOnce once;
    gc::ensure_dynamic_lds(once, (const void *)k_gemm8q<T, 2, NTW, MT, false, true>, (int)lds);
    hipLaunchKernelGGL((k_gemm8q<T, 2, NTW, MT, false, true>), grid, dim3(512), lds, s, g);
}
template <class T>
void dispatch8q(const GemmArgs &g, int mode, int ntw, int mt, dim3 grid, hipStream_t s)
{
    if (g.chan_parts && g.splits == 1) {      // (the launcher admits this for mode 2 without fused statistics only)
        if (ntw == 5) launch8q_cs<T, 5, 2>(g, grid, s);
        else if (mt == 3) launch8q_cs<T, 4, 3>(g, grid, s);
        else launch8q_cs<T, 4, 2>(g, grid, s);
        return;
    }
#define GC_Q(NTW_, MT_) do { if (mode == 2) launch8q<T, 2, NTW_, MT_>(g, grid, s); else launch8q<T, 3, NTW_, MT_>(g, grid, s); } while (0)
    if (ntw == 5) GC_Q(5, 2);
    else { if (mt == 3) GC_Q(4, 3); else GC_Q(4, 2); }
#undef GC_Q
}
}  // namespace

void dn_gemm_launch_fp8(const GemmArgs &g, int dtype, int mode, int ntw, int mt, dim3 grid, hipStream_t s)
{
    if (dtype == DT_BF16) dispatch8q<BF16>(g, mode, ntw, mt, grid, s); else dispatch8q<F16>(g, mode, ntw, mt, grid, s);
}
