// dn_gemm_ln.hip -- instantiates the LEAN LayerNorm fold of the 8-wave GEMM (round 5; k_gemm8<.., LNV> / k_gemm8p<.., 2> in dn_gemm_kernels.h):
// the producer of a LayerNorm's input (proj_in, attn1.to_out + residual, attn2.to_out + residual of the C = 640 / 1280 transformer blocks;
// diffusers BasicTransformerBlock behind /root/reference/gaussctrl/gc_pipeline.py:209-219) leaves per-row partial (sum, sum^2) from its lean
// epilogue, the consumer (Q | K | V^T, attn2.to_q, GEGLU) normalises in its own epilogue -- no LayerNorm launch, and none of the
// everything-epilogue (FUSE = true) that made round 2's form of the same fold slower than the stand-alone kernel.
#include "dn_gemm_kernels.h"

void dn_gemm_launch_ln(const GemmArgs &g, int dtype, int lnv, bool lean, int ntw, int mt8, dim3 grid, hipStream_t s)
{
    if (dtype == DT_BF16) dispatch8ln<BF16>(g, lnv, lean, ntw, mt8, grid, s); else dispatch8ln<F16>(g, lnv, lean, ntw, mt8, grid, s);
}
