set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5za}
mkdir -p $O
export TMPDIR=/tmp
for V in 256 64 256 64; do
  timeout 600 python bench.py --dtype fp8 --fp8-min-hw $V --no-cpu-baseline --no-secondary > $O/bench_fp8_$V.json 2> $O/bench_fp8_$V.err
  python -c "
import json; d=json.loads(open('$O/bench_fp8_$V.json').read().strip().splitlines()[-1]); print('fp8-min-hw=$V', d['value'], d['ms_per_step'])" || tail -5 $O/bench_fp8_$V.err
done
