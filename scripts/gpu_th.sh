#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/th; mkdir -p $O; export PYTHONPATH=.
timeout 900 python -m pytest tests/test_ttail_gpu.py -x -q > $O/tests.log 2>&1; tail -12 $O/tests.log
for v in 0 1; do
  GC_FUSED_HEAD=$v timeout 900 python bench.py --steps 4 --warmup 1 --no-secondary --no-cpu-baseline > $O/bench_head$v.json 2> $O/bench_head$v.err
  python -c "
import json; d=json.loads(open('$O/bench_head$v.json').read().strip().splitlines()[-1]); print('fused_head=$v', d['value'], d['ms_per_step'])"
done
