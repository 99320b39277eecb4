#!/bin/bash
cd $GRAFT_REPO_ROOT; R=$PWD; O=gpurun_out/prof3; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python bench.py --no-cpu-baseline --no-secondary > $O/bench.json 2> $O/bench.err
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o bench -- python $R/bench.py --no-cpu-baseline --no-secondary --steps 2 --warmup 1 > $R/$O/bench_profiled.json 2> $R/$O/bench_profiled.err)
DB=$(find $O/prof -name "*.db" | head -1)
python scripts/rocpd_stats.py $DB 45 > $O/kernel_stats.txt
rm -rf $O/prof
python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('mfma_util_step'), d['roofline'].get('frac'))"
head -50 $O/kernel_stats.txt
