set -x
cd $GRAFT_REPO_ROOT
R=$PWD
O=gpurun_out/${1:-r5c}
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_dist_gpu.py -m gpu -q -x 2>&1 | tail -80 > $O/tests_dist.log; tail -60 $O/tests_dist.log
PYTHONPATH=$R PMC_SETS=0,1 PMC_TIMEOUT=300 timeout 700 python scripts/pmc.py 'k_gemm8q' -- python $R/scripts/bench_fp8.py > $O/pmc_gemm8q_fp8.txt 2>&1
head -60 $O/pmc_gemm8q_fp8.txt
