set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5z6}
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_fullgeom_gpu.py -m gpu -q -x -s -k "fp8_convs_with_folded" 2>&1 | grep -v "^  x0" | tail -14 > $O/tests_h.log; tail -14 $O/tests_h.log
timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); f=d['secondary_fp8']; print('bf16', d['value'], 'f16', d['secondary']['value'], 'fp8', f['value'], f.get('config'), f['roofline']['frac'], f.get('mfma_util_step_mixed_peak'))"
