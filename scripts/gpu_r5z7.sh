# SQ counters (MFMA busy, instruction mix, LDS conflicts) of the final build's dominant kernels: k_attn5, the 8-wave GEMM on conv and linear shapes, k_attn_wide
set -x
cd $GRAFT_REPO_ROOT
R=$PWD
O=gpurun_out/${1:-r5z7}
mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=$R
PMC_SETS=0,1 PMC_TIMEOUT=300 timeout 700 python scripts/pmc.py 'k_attn5' -- python $R/scripts/attn5_check.py time 0 > $O/pmc_attn5.txt 2>&1; tail -25 $O/pmc_attn5.txt
PMC_SETS=0,1 PMC_TIMEOUT=300 timeout 700 python scripts/pmc.py 'k_gemm8|k_gemm<|k_splitk' -- python $R/scripts/bench_kernels.py conv > $O/pmc_gemm_conv.txt 2>&1; tail -40 $O/pmc_gemm_conv.txt | cut -c1-200
PMC_SETS=0,1 PMC_TIMEOUT=300 timeout 700 python scripts/pmc.py 'k_attn_wide|k_attn3|k_attn<' -- python $R/scripts/bench_kernels.py attn > $O/pmc_attn_other.txt 2>&1; tail -30 $O/pmc_attn_other.txt | cut -c1-200
