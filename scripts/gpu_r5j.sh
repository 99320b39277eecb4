# (1) statistics-buffer poison tests on the select-instead-of-multiply build; (2) is the 14-step bench limited by the host's lead over the GPU?
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5j}
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_denoise_kernels_gpu.py -m gpu -q -x -k "poisoned or producer_partials" 2>&1 | tail -5 > $O/tests_poison.log; tail -2 $O/tests_poison.log
for ST in 14 42 14 42; do
  timeout 900 python bench.py --no-cpu-baseline --no-secondary --steps $ST > $O/bench_steps$ST.json 2> $O/bench_steps$ST.err
  python -c "
import json; d=json.loads(open('$O/bench_steps$ST.json').read().strip().splitlines()[-1]); print('steps $ST', d['value'], d['ms_per_step'])"
done
for FL in 1 3; do
  timeout 900 python bench.py --no-cpu-baseline --no-secondary --inflight $FL > $O/bench_inflight$FL.json 2> $O/bench_inflight$FL.err
  python -c "
import json; d=json.loads(open('$O/bench_inflight$FL.json').read().strip().splitlines()[-1]); print('inflight $FL', d['value'], d['ms_per_step'])"
done
