// dn_gemm_plain.hip -- instantiates the GEMM / implicit-conv kernels with the lean epilogue (FUSE = false) and the split-K reduce kernels.
#include "dn_gemm_kernels.h"

void dn_gemm_launch_plain(const GemmArgs &g, int dtype, int mode, int ntw, int mt8, dim3 grid, hipStream_t s)
{
    if (mt8) { if (dtype == DT_BF16) dispatch8<BF16, false>(g, mode, ntw, mt8, grid, s); else dispatch8<F16, false>(g, mode, ntw, mt8, grid, s); }
    else { if (dtype == DT_BF16) dispatch4<BF16, false>(g, mode, ntw, grid, s); else dispatch4<F16, false>(g, mode, ntw, grid, s); }
}

void dn_gemm_launch_splitk_epilogue(const GemmArgs &g, int dtype, hipStream_t s)
{
    if (fuse_of(g)) {
        const dim3 eg((unsigned)((g.N / 4 + 63) / 64), (unsigned)((g.M + 15) / 16));
        if (dtype == DT_BF16) hipLaunchKernelGGL((k_splitk_epilogue<BF16>), eg, dim3(256), 0, s, g);
        else hipLaunchKernelGGL((k_splitk_epilogue<F16>), eg, dim3(256), 0, s, g);
    } else {
        const unsigned eg = (unsigned)std::min<int64_t>((g.M * (g.N / 4) + 255) / 256, 2048);
        if (dtype == DT_BF16) hipLaunchKernelGGL((k_splitk_epilogue_plain<BF16>), dim3(eg), dim3(256), 0, s, g);
        else hipLaunchKernelGGL((k_splitk_epilogue_plain<F16>), dim3(eg), dim3(256), 0, s, g);
    }
}
