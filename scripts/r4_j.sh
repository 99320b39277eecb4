cd $GRAFT_REPO_ROOT
R=$PWD
O=gpurun_out/r4j; mkdir -p $O
export TMPDIR=/tmp
: > $O/gn_dbg.txt
for dbg in 0 2 6; do
  (cd /tmp && GC_GEMM_DBG=$dbg timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o t -- python $R/scripts/gn_kernels_trace.py 6 64 320 > /dev/null 2> $R/$O/err.txt)
  DB=$(find $O/prof -name "*.db" | head -1)
  echo "## gemm dbg $dbg" >> $O/gn_dbg.txt
  python scripts/rocpd_stats.py $DB 30 | grep -E "k_gn_apply|k_gemm8" | awk '{printf "%8s calls %9s avg_us %9s min %s\n", $2, $4, $5, $1}' >> $O/gn_dbg.txt
  rm -rf $O/prof
done
cat $O/gn_dbg.txt
timeout 600 python -m pytest tests/test_denoise_kernels_gpu.py -q -x -k "producer_partials" 2>&1 | tail -3
