#!/bin/bash
# round 6, call K: per-kernel times of the raster-only chain (rocprofv3 --kernel-trace --stats) at 1 M and 4 M; the new GEMM guard test
cd $GRAFT_REPO_ROOT
R=$PWD
O=gpurun_out/r6k
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_denoise_kernels_gpu.py -m gpu -q -x -k "refuses or vae" > $O/tests.txt 2>&1; tail -2 $O/tests.txt
for NG in 1000000 4000000; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$NG -o r -- python $R/bench.py --workload raster --gaussians $NG --steps 32 --warmup 2 --no-cpu-baseline > $R/$O/raster_${NG}_profiled.json 2> $R/$O/raster_${NG}_profiled.err)
  DB=$(find $O/prof_$NG -name "*.db" | head -1)
  python scripts/rocpd_stats.py $DB 40 > $O/raster_kernel_stats_$NG.txt
  rm -rf $O/prof_$NG
  head -28 $O/raster_kernel_stats_$NG.txt | cut -c1-200
done
