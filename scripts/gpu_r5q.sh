# tile-order bit-identity test with its traceback, the GC_GEMM8=0 forced-variant child, bench A/B prev / new build
set -x
cd $GRAFT_REPO_ROOT
R=$PWD
O=gpurun_out/${1:-r5q}
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_denoise_kernels_gpu.py -m gpu -q -x -k "tile_order" 2>&1 | tail -60 > $O/tests_tile.log; grep -v "^  x0" $O/tests_tile.log | tail -40
GC_GEMM8=0 timeout 900 python -m pytest tests/test_denoise_kernels_gpu.py -x -q -k "linear or geglu or conv" 2>&1 | grep -v "^  x0" | tail -30 > $O/tests_gemm8_0.log; tail -30 $O/tests_gemm8_0.log
for L in prev new prev new; do
  if [ $L = prev ]; then export GC_HIP_LIB=$R/gaussctrl_amd/libgaussctrl_hip_prev.so; else unset GC_HIP_LIB; fi
  GC_BENCH_SHAPES=1 timeout 600 python bench.py --no-cpu-baseline --no-secondary > $O/bench_$L.json 2> $O/bench_$L.err
  python -c "
import json; d=json.loads(open('$O/bench_$L.json').read().strip().splitlines()[-1]); print('$L', d['value'], d['ms_per_step'], {k:(v['launches'], v['ms']) for k,v in d['roofline']['other'].items() if 'gemm' in k})"
done
unset GC_HIP_LIB
grep "# shape" $O/bench_new.err | head -70
