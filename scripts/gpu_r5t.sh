# m-tiles per wave of the k-sliced problems (GC_GEMM_SPLIT_MT = 2 default | 3 | 4): microbench + bench
set -x
cd $GRAFT_REPO_ROOT
R=$PWD
O=gpurun_out/${1:-r5t}
mkdir -p $O
export TMPDIR=/tmp
for V in 0 3 4; do
  GC_GEMM_SPLIT_MT=$V timeout 600 python scripts/bench_kernels.py conv > $O/ubench_conv_$V.txt 2>&1
  GC_GEMM_SPLIT_MT=$V timeout 600 python scripts/bench_kernels.py linear > $O/ubench_linear_$V.txt 2>&1
done
paste -d'|' $O/ubench_conv_0.txt $O/ubench_conv_3.txt $O/ubench_conv_4.txt | awk -F'|' '{split($1,a,":"); split($2,b,":"); split($3,c,":"); print a[1] ":" substr(a[2],1,22) " |" substr(b[2],1,22) " |" substr(c[2],1,22)}' | grep -E "16x16|8x8|s2"
paste -d'|' $O/ubench_linear_0.txt $O/ubench_linear_3.txt $O/ubench_linear_4.txt | awk -F'|' '{split($1,a,":"); split($2,b,":"); split($3,c,":"); print a[1] ":" substr(a[2],1,22) " |" substr(b[2],1,22) " |" substr(c[2],1,22)}' | grep -E "K= 5120|K= 2560"
GC_GEMM_SPLIT_MT=3 timeout 600 python -m pytest tests/test_denoise_kernels_gpu.py -m gpu -q -x -k "conv or splitk or linear" 2>&1 | grep -v "^  x0" | tail -3
for V in 0 3 4 0 3 4; do
  GC_GEMM_SPLIT_MT=$V timeout 600 python bench.py --no-cpu-baseline --no-secondary > $O/bench_$V.json 2> $O/bench_$V.err
  python -c "
import json; d=json.loads(open('$O/bench_$V.json').read().strip().splitlines()[-1]); print('SPLIT_MT=$V', d['value'], d['ms_per_step'], {k:(v['launches'], v['ms']) for k,v in d['roofline']['other'].items() if 'gemm' in k and 'generic' not in k})"
done
