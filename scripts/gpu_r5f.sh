# LayerNorm fold at levels 1-3 (existing FUSE-epilogue path, fold_ln = 2) vs default vs the "ln" ablation (upper bound), same box
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5f}
mkdir -p $O
export TMPDIR=/tmp
for V in "GC_X=0" "GC_DN_FOLD_LN=2" "GC_ABLATE=ln" "GC_X=0" "GC_DN_FOLD_LN=2"; do
  T=$(echo $V | tr '=' '_')
  env $V timeout 600 python bench.py --no-cpu-baseline --no-secondary > $O/bench_$T.json 2> $O/bench_$T.err
  python -c "
import json; d=json.loads(open('$O/bench_$T.json').read().strip().splitlines()[-1]); print('$V', d['value'], d['ms_per_step'])"
done
