set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5zd}
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); f=d['secondary_fp8']; print('bf16', d['value'], 'f16', d['secondary']['value'], 'fp8', f['value'], f.get('config'), 'cpu', d['cpu_baseline']['value'])"; tail -2 $O/bench.err
