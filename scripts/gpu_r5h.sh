# lean LayerNorm fold, second build (statistics through the LDS prologue, under the first k-tiles / before the next tile's DMA)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5h}
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_denoise_kernels_gpu.py -m gpu -q -x -k "layernorm_folded" 2>&1 | tail -5 > $O/tests_ln.log; tail -2 $O/tests_ln.log
for V in "GC_X=0" "GC_DN_FOLD_LN=2" "GC_X=0" "GC_DN_FOLD_LN=2"; do
  T=$(echo $V | tr '= ' '__')
  env $V timeout 600 python bench.py --no-cpu-baseline --no-secondary > $O/bench_$T.json 2> $O/bench_$T.err
  python -c "
import json; d=json.loads(open('$O/bench_$T.json').read().strip().splitlines()[-1]); print('$V', d['value'], d['ms_per_step'], {k:(v['launches'], v['ms']) for k,v in d['roofline']['other'].items() if 'linear' in k})"
done
timeout 900 python -m pytest tests/test_fullgeom_gpu.py -m gpu -q -x -s -k "layernorm_folded" 2>&1 | tail -14 > $O/tests_fold_fullgeom.log; tail -12 $O/tests_fold_fullgeom.log
timeout 300 python scripts/cpu_bound_check.py 2>&1 | head -12 > $O/cpu_bound.txt; head -4 $O/cpu_bound.txt
