# First GPU call of the next round (~14 GPU-minutes): what the last session of round 4 could not measure any more.
#  1. the whole `-m gpu` suite on the final build (353 tests; the round-4 HEAD suite of 297 ran before the fp8 work, the 88 fp8 cases and a
#     148-test default-path subset after it)
#  2. default bench line (bf16 + f16 / fp8 secondaries) and the rocprofv3 kernel summaries of the bf16 AND the fp8 run (the fp8 path has no
#     kernel trace under profiles/ yet: k_gemm8q variants, k_layernorm_fp8, k_gn_apply_parts<.., true>, split-K reduce share)
#  3. the forced-variant children over the new fp8 tests (GC_GEMM_MT=3 / 4, GC_GEMM8=0 were reasoned about, only MT=2 was run)
set -x
cd $GRAFT_REPO_ROOT
R=$PWD
O=gpurun_out/${1:-r5a}
mkdir -p $O
export TMPDIR=/tmp
bash scripts/gpu_suite_repeat.sh ${1:-r5a} 1
timeout 600 python bench.py > $O/bench_bf16.json 2> $O/bench_bf16.err
tail -1 $O/bench_bf16.json | cut -c1-300
for DT in bf16 fp8; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$DT -o bench -- python $R/bench.py --dtype $DT --no-cpu-baseline --no-secondary > $R/$O/bench_${DT}_profiled.json 2> $R/$O/bench_${DT}_profiled.err)
  DB=$(find $O/prof_$DT -name "*.db" | head -1)
  python scripts/rocpd_stats.py $DB 70 > $O/bench_kernel_stats_$DT.txt
  rm -rf $O/prof_$DT
  head -14 $O/bench_kernel_stats_$DT.txt
done
for V in "GC_GEMM_MT=3" "GC_GEMM_MT=4" "GC_GEMM8=0"; do
  env $V timeout 300 python -m pytest tests/test_denoise_kernels_gpu.py -x -q -k fp8 2>&1 | tail -2 > $O/tests_fp8_$(echo $V | tr '=' '_').log
  tail -1 $O/tests_fp8_$(echo $V | tr '=' '_').log
done
