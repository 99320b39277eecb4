# 8 x 8-map convolutions on the 8-wave kernel with XCD-owned k-slices (GC_GEMM_CONVSPLIT=2) vs the 4-wave split-K kernel: microbench + bench A/B
set -x
cd $GRAFT_REPO_ROOT
R=$PWD
O=gpurun_out/${1:-r5r}
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_denoise_kernels_gpu.py -m gpu -q -x -k "tile_order" 2>&1 | grep -v "^  x0" | tail -30 > $O/tests_tile.log; tail -3 $O/tests_tile.log
GC_GEMM_CONVSPLIT=2 timeout 900 python -m pytest tests/test_denoise_kernels_gpu.py -x -q -k "conv" 2>&1 | grep -v "^  x0" | tail -5
timeout 600 python scripts/bench_kernels.py conv > $O/ubench_conv_default.txt 2>&1
GC_GEMM_CONVSPLIT=2 timeout 600 python scripts/bench_kernels.py conv > $O/ubench_conv_split2.txt 2>&1
paste -d'|' $O/ubench_conv_default.txt $O/ubench_conv_split2.txt | awk -F'|' '{split($1,a,":"); split($2,b,":"); print a[1] ":" substr(a[2],1,22) " |" substr(b[2],1,22)}'
for V in "GC_X=0" "GC_GEMM_CONVSPLIT=2" "GC_X=0" "GC_GEMM_CONVSPLIT=2"; do
  T=$(echo $V | tr '= ' '__')
  env $V timeout 600 python bench.py --no-cpu-baseline --no-secondary > $O/bench_$T.json 2> $O/bench_$T.err
  python -c "
import json; d=json.loads(open('$O/bench_$T.json').read().strip().splitlines()[-1]); print('$V', d['value'], d['ms_per_step'], {k:(v['launches'], v['ms']) for k,v in d['roofline']['other'].items() if 'conv' in k})"
done
