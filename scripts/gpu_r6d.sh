#!/bin/bash
# round 6, call D (re-entry baseline): the whole -m gpu suite on HEAD (time vs the 1200 s limit), smoke, default bench line,
# the same command under rocprofv3 --kernel-trace --stats, per-shape GEMM table
cd $GRAFT_REPO_ROOT
R=$PWD
O=gpurun_out/r6d
mkdir -p $O
export TMPDIR=/tmp
S=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q --durations=30 > $O/tests.txt 2>&1
echo "suite seconds: $(( $(date +%s) - S ))" | tee -a $O/tests.txt
tail -4 $O/tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
timeout 1200 python bench.py > $O/bench_bf16.json 2> $O/bench_bf16.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o bench -- python $R/bench.py --no-cpu-baseline --no-secondary > $R/$O/bench_profiled.json 2> $R/$O/bench_profiled.err)
DB=$(find $O/prof -name "*.db" | head -1)
python scripts/rocpd_stats.py $DB 70 > $O/bench_kernel_stats.txt
rm -rf $O/prof
head -30 $O/bench_kernel_stats.txt | cut -c1-170
GC_BENCH_SHAPES=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $O/bench_shapes.json 2> $O/shapes.txt
python -c "
import json
d=json.loads(open('$O/bench_bf16.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], d['mfma_util_step'], r['frac'], r['avg_launch_us'], 'f16', d['secondary']['value'], 'fp8', d['secondary_fp8']['value'])
for k,v in r['other'].items(): print('   ', k, v)
print(d.get('roofline_raster'))
"
