# CFG-shared prefix: the level-0 tail kernel reads the shared rows for both CFG halves (gc_ttail_desc.in_rows) instead of three duplicate copies: tests + A/B
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5ze}
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_denoise_model_gpu.py tests/test_fullgeom_gpu.py tests/test_denoise_kernels_gpu.py -m gpu -q -x -k "cfg_shared or edit_f7_h64_all or edit_chunk or tail or batch_invariant or layernorm_folded" 2>&1 | grep -v "^  x0" | tail -4
for V in 0 1 0 1; do
  GC_TAIL_INROWS=$V timeout 600 python bench.py --no-cpu-baseline --no-secondary > $O/bench_$V.json 2> $O/bench_$V.err
  python -c "
import json; d=json.loads(open('$O/bench_$V.json').read().strip().splitlines()[-1]); print('TAIL_INROWS=$V', d['value'], d['ms_per_step'])"
done
