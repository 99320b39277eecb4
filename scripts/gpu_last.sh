#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/last; mkdir -p $O; export PYTHONPATH=.
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/tests.log; tail -3 $O/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
timeout 900 python bench.py > $O/bench_bf16.json 2> $O/bench.err
python -c "
import json; d=json.loads(open('$O/bench_bf16.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['mfma_util_step'], d['roofline']['frac'], d['secondary']['value'])"
