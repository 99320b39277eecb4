#!/bin/bash
O=gpurun_out/tt2; mkdir -p $O
export PYTHONPATH=.
timeout 900 python -m pytest tests/test_ttail_gpu.py -x -q > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout 300 python scripts/ttail_check.py bf16 time > $O/check.log 2>&1; grep -E "per-op|fused:" $O/check.log
for v in 0 1; do
  GC_FUSED_TAIL=$v timeout 900 python bench.py --steps 4 --warmup 1 --no-secondary > $O/bench_fused$v.json 2> $O/bench_fused$v.err
  python -c "
import json,sys
d=json.loads(open('$O/bench_fused$v.json').read().strip().splitlines()[-1]); print('fused_tail=$v', d['value'], d['ms_per_step'])"
done
