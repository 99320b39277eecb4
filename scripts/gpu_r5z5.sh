# --dtype fp8: e4m3 convolutions + e4m3 transformer linears (default, no LayerNorm fold / graph merges) vs e4m3 convolutions + the bf16 folded / merged linears
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5z5}
mkdir -p $O
export TMPDIR=/tmp
for V in 7 0 7 0; do
  timeout 600 python bench.py --dtype fp8 --fp8-linears $V --no-cpu-baseline --no-secondary > $O/bench_fp8_$V.json 2> $O/bench_fp8_$V.err
  python -c "
import json; d=json.loads(open('$O/bench_fp8_$V.json').read().strip().splitlines()[-1]); print('fp8-linears=$V', d['value'], d['ms_per_step'], d.get('mfma_util_step'))"
done
timeout 600 python bench.py --no-cpu-baseline --no-secondary > $O/bench_bf16.json 2> $O/bench_bf16.err; python -c "
import json; d=json.loads(open('$O/bench_bf16.json').read().strip().splitlines()[-1]); print('bf16', d['value'], d['ms_per_step'])"
