# fp8 path: 16 x 16-map e4m3 convolutions k-sliced on 192-row tiles (3 slices) vs 128-row tiles (2 slices, previous build)
set -x
cd $GRAFT_REPO_ROOT
R=$PWD
O=gpurun_out/${1:-r5zb}
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_denoise_kernels_gpu.py -m gpu -q -k "fp8" 2>&1 | grep -v "^  x0" | tail -3
for L in prev new prev new; do
  if [ $L = prev ]; then export GC_HIP_LIB=$R/gaussctrl_amd/libgaussctrl_hip_prev.so; else unset GC_HIP_LIB; fi
  timeout 600 python bench.py --dtype fp8 --no-cpu-baseline --no-secondary > $O/bench_$L.json 2> $O/bench_$L.err
  python -c "
import json; d=json.loads(open('$O/bench_$L.json').read().strip().splitlines()[-1]); print('$L', d['value'], d['ms_per_step'], {k:(v['launches'], v['ms']) for k,v in d['roofline']['other'].items() if 'gemm8q' in k}, d['roofline']['kernel'][:40], d['roofline']['avg_launch_us'])"
done
unset GC_HIP_LIB
timeout 1500 python -m pytest tests/test_fullgeom_gpu.py -m gpu -q -k "fp8_convs_with_folded or edit_f7_h64_fp8_convs" 2>&1 | grep -v "^  x0" | tail -3
