#!/bin/bash
# round 6, call R: kernel timeline of the default bench (every dispatch: start, end, queue, stream) for an offline idle-gap analysis
cd $GRAFT_REPO_ROOT
R=$PWD
O=gpurun_out/r6r
mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o bench -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $R/$O/bench_profiled.json 2> $R/$O/bench_profiled.err)
DB=$(find $O/prof -name "*.db" | head -1)
python scripts/rocpd_dump.py $DB $O/timeline.txt
python scripts/rocpd_stats.py $DB 10 k_attn5 | tail -14
rm -rf $O/prof
gzip -9 $O/timeline.txt; ls -la $O
grep -o '"value": [0-9.]*' $O/bench_profiled.json | head -1
