# in-kernel reduce of the k-sliced problems (gc_gemm_desc.tile_counters): parity / race test, microbench and bench A/B (GC_FUSED_REDUCE=0|1)
set -x
cd $GRAFT_REPO_ROOT
R=$PWD
O=gpurun_out/${1:-r5u}
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_denoise_kernels_gpu.py -m gpu -q -x -k "ksliced or tile_order" 2>&1 | grep -v "^  x0" | tail -40 > $O/tests_fix.log; tail -25 $O/tests_fix.log
timeout 1500 python -m pytest tests/test_denoise_kernels_gpu.py tests/test_gemm_variants_gpu.py -m gpu -q -x 2>&1 | grep -v "^  x0" | tail -5 > $O/tests_k.log; tail -3 $O/tests_k.log
for V in 0 1; do
  GC_FUSED_REDUCE=$V timeout 600 python scripts/bench_kernels.py conv > $O/ubench_conv_$V.txt 2>&1
  GC_FUSED_REDUCE=$V timeout 600 python scripts/bench_kernels.py linear > $O/ubench_linear_$V.txt 2>&1
done
paste -d'|' $O/ubench_conv_0.txt $O/ubench_conv_1.txt | awk -F'|' '{split($1,a,":"); split($2,b,":"); print a[1] ":" substr(a[2],1,22) " |" substr(b[2],1,22)}' | grep -E "16x16|8x8|s2"
paste -d'|' $O/ubench_linear_0.txt $O/ubench_linear_1.txt | awk -F'|' '{split($1,a,":"); split($2,b,":"); print a[1] ":" substr(a[2],1,22) " |" substr(b[2],1,22)}' | grep -E "K= 5120|K= 2560"
for V in 0 1 0 1; do
  GC_FUSED_REDUCE=$V timeout 600 python bench.py --no-cpu-baseline --no-secondary > $O/bench_$V.json 2> $O/bench_$V.err
  python -c "
import json; d=json.loads(open('$O/bench_$V.json').read().strip().splitlines()[-1]); print('FUSED_REDUCE=$V', d['value'], d['ms_per_step'], {k:(v['launches'], v['ms']) for k,v in d['roofline']['other'].items() if 'gemm' in k and 'generic' not in k})"
done
timeout 1500 python -m pytest tests/test_fullgeom_gpu.py tests/test_denoise_model_gpu.py -m gpu -q -x -k "batch_invariant or layernorm_folded or edit_f7_h64_all or edit_chunk or config4_geometry" 2>&1 | grep -v "^  x0" | tail -5 > $O/tests_m.log; tail -3 $O/tests_m.log
