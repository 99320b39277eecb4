#!/bin/bash
# round 6, final evidence (one box, one call): the -m gpu suite + smoke, the default bench line in the driver's form, the same command under rocprofv3 --kernel-trace --stats
# (default = 2 launch sets in flight; --inflight 1 = durations comparable with the live HIP-event numbers; --dtype fp8), raster-only lines
cd $GRAFT_REPO_ROOT
R=$PWD
O=gpurun_out/${1:-r6z}
mkdir -p $O
export TMPDIR=/tmp
S=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q --durations=10 > $O/tests.txt 2>&1
echo "suite seconds: $(( $(date +%s) - S ))" | tee -a $O/tests.txt
tail -3 $O/tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_bf16.json 2> $O/bench_bf16.err
for V in "bf16 2" "bf16 1" "fp8 2"; do
  set -- $V; DT=$1; IF=$2
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o bench -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --dtype $DT --inflight $IF --no-cpu-baseline --no-secondary > $R/$O/bench_${DT}_if${IF}_profiled.json 2> $R/$O/bench_${DT}_if${IF}_profiled.err)
  DB=$(find $O/prof -name "*.db" | head -1)
  python scripts/rocpd_stats.py $DB 70 k_attn5 > $O/bench_kernel_stats_${DT}_if$IF.txt
  rm -rf $O/prof
  head -4 $O/bench_kernel_stats_${DT}_if$IF.txt | cut -c1-170; grep -A9 "by workgroup count" $O/bench_kernel_stats_${DT}_if$IF.txt | cut -c60-170
done
for NG in 1000000 4000000; do
  timeout 600 python bench.py --workload raster --gaussians $NG --steps 32 --warmup 2 --no-cpu-baseline > $O/raster_${NG}.json 2> $O/raster_${NG}.err
  python -c "
import json; d=json.loads(open('$O/raster_${NG}.json').read().strip().splitlines()[-1]); c=d['roofline']['chain']; print('$NG', d['value'], c['kernel_us_per_view'], 'frac', c['frac'], 'counters', c['frac_counters'], 'ratio', c['traffic_ratio'])"
done
python -c "
import json
d=json.loads(open('$O/bench_bf16.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], d['mfma_util_step'], r['frac'], r['avg_launch_us'], 'f16', d['secondary']['value'], 'fp8', d['secondary_fp8']['value'], 'cpu', d['cpu_baseline']['value'])
print(r.get('launch_kinds'))
for k in ('bf16_if2','bf16_if1','fp8_if2'):
    p=json.loads(open('$O/bench_'+k+'_profiled.json').read().strip().splitlines()[-1]); print(k, 'profiled', p['value'], p['roofline']['avg_launch_us'])
"
