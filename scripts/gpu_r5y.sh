# head-size-160 attention with the tile prefetch at one workgroup per CU (no spills): split and unsplit microbench vs previous build, bench A/B
set -x
cd $GRAFT_REPO_ROOT
R=$PWD
O=gpurun_out/${1:-r5y}
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_denoise_kernels_gpu.py -m gpu -q -x -k "attention" 2>&1 | grep -v "^  x0" | tail -3
for BI in 0 1; do for L in prev new; do
  if [ $L = prev ]; then export GC_HIP_LIB=$R/gaussctrl_amd/libgaussctrl_hip_prev.so; else unset GC_HIP_LIB; fi
  GC_BATCH_INVARIANT=$BI timeout 600 python scripts/bench_kernels.py attn 2>&1 | grep -E "L=  256|L=   64" | sed "s/^/BI=$BI $L /"
done; done
unset GC_HIP_LIB
for L in prev new prev new; do
  if [ $L = prev ]; then export GC_HIP_LIB=$R/gaussctrl_amd/libgaussctrl_hip_prev.so; else unset GC_HIP_LIB; fi
  timeout 600 python bench.py --no-cpu-baseline --no-secondary > $O/bench_$L.json 2> $O/bench_$L.err
  python -c "
import json; d=json.loads(open('$O/bench_$L.json').read().strip().splitlines()[-1]); print('$L', d['value'], d['ms_per_step'], {k:(v['launches'], v['ms']) for k,v in d['roofline']['other'].items() if 'k_attn' in k})"
done
