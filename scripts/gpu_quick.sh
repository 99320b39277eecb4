#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/quick; mkdir -p $O; export PYTHONPATH=.
timeout 1800 python -m pytest tests -m gpu -x -q -k "${GC_K:-denoise or gemm or random_shapes or fullgeom or ttail}" > $O/tests.log 2>&1; tail -4 $O/tests.log
timeout 900 python bench.py --no-cpu-baseline --no-secondary --steps ${GC_STEPS:-7} > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d.get('mfma_util_step'))"
