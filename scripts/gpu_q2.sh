#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/q2; mkdir -p $O; export PYTHONPATH=.
timeout 300 python scripts/attn_small_check.py 2>&1 | grep -v amdgpu.ids | tee $O/attn_small.txt
timeout 1500 python -m pytest tests -m gpu -x -q -k "attention or attn or xview or denoise or fullgeom or random_shapes" > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 900 python bench.py --no-cpu-baseline --no-secondary > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step']); print({k:v for k,v in d['roofline']['other'].items() if 'k_attn' in k})"
