#!/bin/bash
# round 6: the default bench under rocprofv3 with ONE launch set in flight and the denoise step on ONE stream (GC_DN_STREAMS=0 --inflight 1): no kernel shares the GPU
# with another, so the trace's per-size k_attn5 durations are comparable with bench.py's live single-stream HIP-event numbers (roofline.launch_kinds)
cd $GRAFT_REPO_ROOT
R=$PWD
O=gpurun_out/r6za
mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && GC_DN_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o bench -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --inflight 1 --no-cpu-baseline --no-secondary > $R/$O/bench_profiled.json 2> $R/$O/bench_profiled.err)
DB=$(find $O/prof -name "*.db" | head -1)
python scripts/rocpd_stats.py $DB 70 k_attn5 > $O/bench_kernel_stats_single_stream.txt
rm -rf $O/prof
grep -A9 "by workgroup count" $O/bench_kernel_stats_single_stream.txt | cut -c60-170; tail -12 $O/bench_kernel_stats_single_stream.txt | head -2
python -c "
import json
d=json.loads(open('$O/bench_profiled.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], r['avg_launch_us']); print(r['launch_kinds'])"
