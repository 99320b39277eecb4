# Round 5, fifth GPU call: packed (tile, id) binning -- list bit-exactness + views tests, raster bench A/B vs the previous build, kernel traces of
# the batched raster workload; RefShard world 4 / 8 end-to-end; tight-box test with the tightened bars.
set -x
cd $GRAFT_REPO_ROOT
R=$PWD
O=gpurun_out/${1:-r5e}
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_raster_gpu.py tests/test_raster_views_gpu.py -m gpu -q -x 2>&1 | tail -12 > $O/tests_raster.log; tail -3 $O/tests_raster.log
for NG in 1000000 4000000; do
  for L in prev new prev new; do
    if [ $L = prev ]; then export GC_HIP_LIB=$R/gaussctrl_amd/libgaussctrl_hip_prev.so; else unset GC_HIP_LIB; fi
    timeout 600 python bench.py --workload raster --gaussians $NG --steps 24 --warmup 2 --no-cpu-baseline > $O/raster_${NG}_$L.json 2> $O/raster_${NG}_$L.err
    python -c "
import json; d=json.loads(open('$O/raster_${NG}_$L.json').read().strip().splitlines()[-1]); c=d['roofline']['chain']; print('$NG $L', d['value'], c['kernel_us_per_view'], c['frac'], c['frac_processed_pairs'], {k:v['avg_us'] for k,v in d['roofline']['stages'].items()})"
  done
done
unset GC_HIP_LIB
for NG in 1000000 4000000; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$NG -o bench -- python $R/bench.py --workload raster --gaussians $NG --steps 16 --warmup 2 --no-cpu-baseline > $R/$O/raster_${NG}_profiled.json 2> $R/$O/raster_${NG}_profiled.err)
  DB=$(find $O/prof_$NG -name "*.db" | head -1)
  python scripts/rocpd_stats.py $DB 40 > $O/raster_kernel_stats_$NG.txt
  rm -rf $O/prof_$NG
  head -30 $O/raster_kernel_stats_$NG.txt
done
timeout 1500 python -m pytest tests/test_dist_gpu.py -m gpu -q -x -k "world4" 2>&1 | tail -30 > $O/tests_shard.log; tail -5 $O/tests_shard.log
