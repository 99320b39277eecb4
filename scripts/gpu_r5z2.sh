# k_attn_wide (set-split head-size-160 attention, all 256 queries of a (frame, head) in one 8-wave workgroup): tests, microbench, bench A/B (GC_ATTN_V=32 = 64-query form)
set -x
cd $GRAFT_REPO_ROOT
R=$PWD
O=gpurun_out/${1:-r5z2}
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_denoise_kernels_gpu.py tests/test_plugin_gpu.py -m gpu -q -x -k "attention" 2>&1 | grep -v "^  x0" | tail -30 > $O/tests_a.log; tail -4 $O/tests_a.log
for V in 32 0; do
  GC_ATTN_V=$V timeout 600 python scripts/bench_kernels.py attn 2>&1 | grep -E "L=  256|L=   64" | sed "s/^/ATTN_V=$V /"
done
for V in 32 0 32 0; do
  GC_ATTN_V=$V timeout 600 python bench.py --no-cpu-baseline --no-secondary > $O/bench_$V.json 2> $O/bench_$V.err
  python -c "
import json; d=json.loads(open('$O/bench_$V.json').read().strip().splitlines()[-1]); print('ATTN_V=$V', d['value'], d['ms_per_step'], {k:(v['launches'], v['ms'], v['tflops']) for k,v in d['roofline']['other'].items() if 'k_attn' in k})"
done
timeout 1500 python -m pytest tests/test_fullgeom_gpu.py tests/test_denoise_model_gpu.py tests/test_dist_gpu.py -m gpu -q -x -k "batch_invariant or edit_f7_h64_all or edit_chunk or config4_geometry or two_ranks" 2>&1 | grep -v "^  x0" | tail -4 > $O/tests_m.log; tail -3 $O/tests_m.log
