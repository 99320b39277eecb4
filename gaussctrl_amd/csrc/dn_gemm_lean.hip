// dn_gemm_lean.hip -- instantiates the 8-wave GEMM / implicit-conv kernel with the LEAN epilogue (bias / row-vector / scale / residual /
// 2-byte store only): the launches that need nothing else -- most convolutions and linears of the UNet -- do not fetch their way through
// the GEGLU / activation / fp32 / transposed-store code of the full epilogue.
#include "dn_gemm_kernels.h"

void dn_gemm_launch_lean(const GemmArgs &g, int dtype, int mode, int ntw, int mt8, dim3 grid, hipStream_t s)
{
    if (dtype == DT_BF16) dispatch8lean<BF16>(g, mode, ntw, mt8, grid, s); else dispatch8lean<F16>(g, mode, ntw, mt8, grid, s);
}
