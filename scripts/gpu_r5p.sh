# GEMM tile order (column panels per XCD, XCD-owned k-slices): bit-identity + parity tests, per-shape microbench and bench A/B vs the previous build
set -x
cd $GRAFT_REPO_ROOT
R=$PWD
O=gpurun_out/${1:-r5p}
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_denoise_kernels_gpu.py -m gpu -q -x -k "tile_order or linear or conv or geglu or splitk or fp8" 2>&1 | tail -8 > $O/tests_k.log; tail -3 $O/tests_k.log
for V in prev new pw0; do
  unset GC_HIP_LIB GC_GEMM_PW
  if [ $V = prev ]; then export GC_HIP_LIB=$R/gaussctrl_amd/libgaussctrl_hip_prev.so; fi
  if [ $V = pw0 ]; then export GC_GEMM_PW=0; fi
  timeout 600 python scripts/bench_kernels.py conv > $O/ubench_conv_$V.txt 2>&1
  timeout 600 python scripts/bench_kernels.py linear > $O/ubench_linear_$V.txt 2>&1
done
unset GC_HIP_LIB GC_GEMM_PW
paste -d'|' $O/ubench_conv_prev.txt $O/ubench_conv_new.txt $O/ubench_conv_pw0.txt | awk -F'|' '{split($1,a,":"); split($2,b,":"); split($3,c,":"); print a[1] ":" substr(a[2],1,22) " |" substr(b[2],1,22) " |" substr(c[2],1,22)}'
paste -d'|' $O/ubench_linear_prev.txt $O/ubench_linear_new.txt $O/ubench_linear_pw0.txt | awk -F'|' '{split($1,a,":"); split($2,b,":"); split($3,c,":"); print a[1] ":" substr(a[2],1,22) " |" substr(b[2],1,22) " |" substr(c[2],1,22)}'
for L in prev new prev new; do
  if [ $L = prev ]; then export GC_HIP_LIB=$R/gaussctrl_amd/libgaussctrl_hip_prev.so; else unset GC_HIP_LIB; fi
  GC_BENCH_SHAPES=1 timeout 600 python bench.py --no-cpu-baseline --no-secondary > $O/bench_$L.json 2> $O/bench_$L.err
  python -c "
import json; d=json.loads(open('$O/bench_$L.json').read().strip().splitlines()[-1]); print('$L', d['value'], d['ms_per_step'], {k:(v['launches'], v['ms']) for k,v in d['roofline']['other'].items() if 'gemm' in k})"
done
unset GC_HIP_LIB
grep "# shape" $O/bench_new.err | head -60
