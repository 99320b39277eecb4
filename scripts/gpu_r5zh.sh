# split-K reduce kernels: the residual rows fetched together with the slabs (one round trip less): tests, microbench, bench A/B vs previous build
set -x
cd $GRAFT_REPO_ROOT
R=$PWD
O=gpurun_out/${1:-r5zh}
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_denoise_kernels_gpu.py -m gpu -q -x -k "conv or linear or splitk or nan_poisoned or tile_order" 2>&1 | grep -v "^  x0" | tail -3
for L in prev new; do
  if [ $L = prev ]; then export GC_HIP_LIB=$R/gaussctrl_amd/libgaussctrl_hip_prev.so; else unset GC_HIP_LIB; fi
  timeout 600 python scripts/bench_kernels.py conv 2>&1 | grep -E "16x16   1280-> 1280 s1 ups0|8x8    1280-> 1280 s1 ups0|32x32    640->  640 s2" | sed "s/^/$L /"
done
for L in prev new prev new; do
  if [ $L = prev ]; then export GC_HIP_LIB=$R/gaussctrl_amd/libgaussctrl_hip_prev.so; else unset GC_HIP_LIB; fi
  timeout 600 python bench.py --no-cpu-baseline --no-secondary > $O/bench_$L.json 2> $O/bench_$L.err
  python -c "
import json; d=json.loads(open('$O/bench_$L.json').read().strip().splitlines()[-1]); print('$L', d['value'], d['ms_per_step'])"
done
unset GC_HIP_LIB
timeout 1500 python -m pytest tests/test_fullgeom_gpu.py tests/test_denoise_model_gpu.py -m gpu -q -x -k "batch_invariant or edit_f7_h64_all or edit_chunk or vae" 2>&1 | grep -v "^  x0" | tail -3
