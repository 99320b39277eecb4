// dn_gemm_fuse.hip -- instantiates the GEMM / implicit-conv kernels with the fused-normalisation epilogue (FUSE = true: producer-side
// row / GroupNorm-group statistics, LayerNorm-folded consumer).
#include "dn_gemm_kernels.h"

void dn_gemm_launch_fuse(const GemmArgs &g, int dtype, int mode, int ntw, int mt8, dim3 grid, hipStream_t s)
{
    if (mt8) { if (dtype == DT_BF16) dispatch8<BF16, true>(g, mode, ntw, mt8, grid, s); else dispatch8<F16, true>(g, mode, ntw, mt8, grid, s); }
    else { if (dtype == DT_BF16) dispatch4<BF16, true>(g, mode, ntw, grid, s); else dispatch4<F16, true>(g, mode, ntw, grid, s); }
}
