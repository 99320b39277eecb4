# text cross-attention folded into two GEMMs: kernel tests, model-level parity, A/B
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5m}
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_denoise_kernels_gpu.py -m gpu -q -x -k "text_cross or layernorm_folded" 2>&1 | tail -25 > $O/tests_k.log; tail -6 $O/tests_k.log
timeout 1200 python -m pytest tests/test_fullgeom_gpu.py tests/test_denoise_model_gpu.py -m gpu -q -x -s -k "layernorm_folded or cfg_shared" 2>&1 | tail -25 > $O/tests_m.log; tail -14 $O/tests_m.log
for V in "GC_TEXT_FOLD=0" "GC_TEXT_FOLD=1" "GC_TEXT_FOLD=0" "GC_TEXT_FOLD=1"; do
  T=$(echo $V | tr '= ' '__')
  env $V timeout 600 python bench.py --no-cpu-baseline --no-secondary > $O/bench_$T.json 2> $O/bench_$T.err
  python -c "
import json; d=json.loads(open('$O/bench_$T.json').read().strip().splitlines()[-1]); print('$V', d['value'], d['ms_per_step'], {k:(v['launches'], v['ms']) for k,v in d['roofline']['other'].items() if 'linear' in k or 'k_attn<' in k})"
  tail -2 $O/bench_$T.err
done
