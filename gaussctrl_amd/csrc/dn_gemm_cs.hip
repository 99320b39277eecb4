// dn_gemm_cs.hip -- instantiates the 8-wave GEMM / implicit-conv kernel with the lean epilogue + per-channel partial sums of the
// stored output (CS = true: the statistics pass of the following GroupNorm, GemmArgs::chan_parts) and the matching split-K reduce kernel.
#include "dn_gemm_kernels.h"

void dn_gemm_launch_cs(const GemmArgs &g, int dtype, int mode, int ntw, int mt8, dim3 grid, hipStream_t s)
{
    if (dtype == DT_BF16) dispatch8cs<BF16>(g, mode, ntw, mt8, grid, s); else dispatch8cs<F16>(g, mode, ntw, mt8, grid, s);
}

void dn_gemm_launch_splitk_epilogue_cs(const GemmArgs &g, int dtype, hipStream_t s)
{
    const dim3 eg((unsigned)((g.N / 4 + 15) / 16), (unsigned)((g.M + CS_RB - 1) / CS_RB));
    if (dtype == DT_BF16) hipLaunchKernelGGL((k_splitk_epilogue_cs<BF16>), eg, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((k_splitk_epilogue_cs<F16>), eg, dim3(256), 0, s, g);
}
