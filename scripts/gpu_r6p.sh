#!/bin/bash
# round 6, call P (evidence of the build): default bench line in the driver's form, the same command under rocprofv3 --kernel-trace --stats (bf16 and fp8),
# per-shape GEMM table, configs[3] lines (chunk 8 + mask, fp8 / bf16)
cd $GRAFT_REPO_ROOT
R=$PWD
O=gpurun_out/r6p
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_bf16.json 2> $O/bench_bf16.err
for DT in bf16 fp8; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$DT -o bench -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --dtype $DT --no-cpu-baseline --no-secondary > $R/$O/bench_${DT}_profiled.json 2> $R/$O/bench_${DT}_profiled.err)
  DB=$(find $O/prof_$DT -name "*.db" | head -1)
  python scripts/rocpd_stats.py $DB 70 > $O/bench_kernel_stats_$DT.txt
  rm -rf $O/prof_$DT
  head -6 $O/bench_kernel_stats_$DT.txt | cut -c1-170
done
GC_BENCH_SHAPES=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $O/bench_shapes.json 2> $O/shapes.txt
for DT in fp8 bf16; do
  timeout 600 python bench.py --dtype $DT --chunk-size 8 --mask --steps 10 --warmup 3 --no-secondary --no-cpu-baseline > $O/bench_config4_$DT.json 2> $O/bench_config4_$DT.err
  python -c "
import json; d=json.loads([l for l in open('$O/bench_config4_$DT.json') if l.startswith('{')][-1]); print('configs[3] $DT', d['value'], d['ms_per_step'], d['config'].get('chunks_per_launch_set'))"
done
python -c "
import json
d=json.loads(open('$O/bench_bf16.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], d['mfma_util_step'], r['frac'], r['avg_launch_us'], 'f16', d['secondary']['value'], 'fp8', d['secondary_fp8']['value'], 'cpu', d['cpu_baseline']['value'])
for k,v in r['other'].items(): print('   ', k, v)
print(d.get('roofline_raster'))
for k in ('bf16','fp8'):
    p=json.loads(open('$O/bench_'+k+'_profiled.json').read().strip().splitlines()[-1]); print(k, 'profiled', p['value'], p['roofline']['avg_launch_us'])
"
