# FF-down + proj_out merged GEMM and Q-only ControlNet projection: tests, A/B
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5n}
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_fullgeom_gpu.py tests/test_denoise_model_gpu.py tests/test_plugin_gpu.py -m gpu -q -x -s -k "layernorm_folded or cfg_shared or config4 or batch_invariant or plugin or edit" 2>&1 | tail -40 > $O/tests_m.log; tail -14 $O/tests_m.log
for V in "GC_FFOUT_MERGE=0 GC_Q_ONLY=0" "GC_FFOUT_MERGE=1 GC_Q_ONLY=0" "GC_FFOUT_MERGE=1 GC_Q_ONLY=1" "GC_FFOUT_MERGE=0 GC_Q_ONLY=0" "GC_FFOUT_MERGE=1 GC_Q_ONLY=0" "GC_FFOUT_MERGE=1 GC_Q_ONLY=1"; do
  T=$(echo $V | tr '= ' '__')
  env $V timeout 600 python bench.py --no-cpu-baseline --no-secondary > $O/bench_$T.json 2> $O/bench_$T.err
  python -c "
import json; d=json.loads(open('$O/bench_$T.json').read().strip().splitlines()[-1]); print('$V', d['value'], d['ms_per_step'], {k:(v['launches'], v['ms']) for k,v in d['roofline']['other'].items() if 'linear' in k or 'k_attn<' in k})"
  tail -2 $O/bench_$T.err
done
