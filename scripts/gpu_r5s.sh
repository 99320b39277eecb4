# small-grid convs on the k-sliced 8-wave kernel (default) + reduce kernels with 8 slab loads in flight: tests, microbench, bench A/B vs previous build
set -x
cd $GRAFT_REPO_ROOT
R=$PWD
O=gpurun_out/${1:-r5s}
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_denoise_kernels_gpu.py tests/test_gemm_variants_gpu.py -m gpu -q -x 2>&1 | grep -v "^  x0" | tail -5 > $O/tests_k.log; tail -3 $O/tests_k.log
timeout 1500 python -m pytest tests/test_fullgeom_gpu.py tests/test_denoise_model_gpu.py -m gpu -q -x -k "batch_invariant or layernorm_folded or edit_f7_h64_all or edit_chunk or config4_geometry" 2>&1 | grep -v "^  x0" | tail -5 > $O/tests_m.log; tail -3 $O/tests_m.log
timeout 600 python scripts/bench_kernels.py conv > $O/ubench_conv.txt 2>&1; grep -E "16x16|8x8|s2" $O/ubench_conv.txt
timeout 600 python scripts/bench_kernels.py linear > $O/ubench_linear.txt 2>&1; grep -E "M=" $O/ubench_linear.txt
for L in prev new prev new; do
  if [ $L = prev ]; then export GC_HIP_LIB=$R/gaussctrl_amd/libgaussctrl_hip_prev.so; else unset GC_HIP_LIB; fi
  GC_BENCH_SHAPES=1 timeout 600 python bench.py --no-cpu-baseline --no-secondary > $O/bench_$L.json 2> $O/bench_$L.err
  python -c "
import json; d=json.loads(open('$O/bench_$L.json').read().strip().splitlines()[-1]); print('$L', d['value'], d['ms_per_step'], {k:(v['launches'], v['ms']) for k,v in d['roofline']['other'].items() if 'gemm' in k})"
done
unset GC_HIP_LIB
grep "# shape" $O/bench_new.err | head -24
