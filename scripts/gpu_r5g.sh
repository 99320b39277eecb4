# lean LayerNorm fold: kernel tests, model-level tests with the fold on, A/B default vs fold (lean) vs fold (FUSE, round 2)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5g}
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_denoise_kernels_gpu.py -m gpu -q -x -k "layernorm_folded or linear or gemm" 2>&1 | tail -12 > $O/tests_ln.log; tail -3 $O/tests_ln.log
for V in "GC_X=0" "GC_DN_FOLD_LN=2" "GC_X=0" "GC_DN_FOLD_LN=2" "GC_DN_FOLD_LN=2 GC_GEMM_DBG=8"; do
  T=$(echo $V | tr '= ' '__')
  env $V timeout 600 python bench.py --no-cpu-baseline --no-secondary > $O/bench_$T.json 2> $O/bench_$T.err
  python -c "
import json; d=json.loads(open('$O/bench_$T.json').read().strip().splitlines()[-1]); print('$V', d['value'], d['ms_per_step'])"
done
