#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/sweep; mkdir -p $O; export PYTHONPATH=.
for v in 2 3 2 3; do
  timeout 900 python bench.py --no-cpu-baseline --no-secondary --inflight $v > $O/b_$v.json 2>/dev/null
  python -c "
import json; d=json.loads(open('$O/b_$v.json').read().strip().splitlines()[-1]); print('inflight=$v', d['value'], d['ms_per_step'])"
done
