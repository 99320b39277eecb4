# head-size-80 attention (k_attn3) with 32 queries per wave (128 per workgroup, one workgroup per CU) vs 16 (prev build): microbench + tests + bench
set -x
cd $GRAFT_REPO_ROOT
R=$PWD
O=gpurun_out/${1:-r5z4}
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_denoise_kernels_gpu.py tests/test_plugin_gpu.py -m gpu -q -k "attention" 2>&1 | grep -v "^  x0" | tail -5
for L in prev new; do
  if [ $L = prev ]; then export GC_HIP_LIB=$R/gaussctrl_amd/libgaussctrl_hip_prev.so; else unset GC_HIP_LIB; fi
  timeout 600 python scripts/bench_kernels.py attn 2>&1 | grep -E "L= 1024" | sed "s/^/$L /"
done
for L in prev new prev new; do
  if [ $L = prev ]; then export GC_HIP_LIB=$R/gaussctrl_amd/libgaussctrl_hip_prev.so; else unset GC_HIP_LIB; fi
  timeout 600 python bench.py --no-cpu-baseline --no-secondary > $O/bench_$L.json 2> $O/bench_$L.err
  python -c "
import json; d=json.loads(open('$O/bench_$L.json').read().strip().splitlines()[-1]); print('$L', d['value'], d['ms_per_step'], {k:(v['launches'], v['ms'], v['tflops']) for k,v in d['roofline']['other'].items() if 'k_attn' in k})"
done
