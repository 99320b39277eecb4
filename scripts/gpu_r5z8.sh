# k_attn5: own chunk swizzle for the 16-row V^T block of the 16x16x32 P V MFMAs (2-way LDS bank conflict of round 4's PV16): tests, counters, A/B
set -x
cd $GRAFT_REPO_ROOT
R=$PWD
O=gpurun_out/${1:-r5z8}
mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=$R
timeout 900 python -m pytest tests/test_denoise_kernels_gpu.py tests/test_plugin_gpu.py -m gpu -q -k "attention" 2>&1 | grep -v "^  x0" | tail -4
for L in prev new prev new; do
  if [ $L = prev ]; then export GC_HIP_LIB=$R/gaussctrl_amd/libgaussctrl_hip_prev.so; else unset GC_HIP_LIB; fi
  timeout 300 python scripts/attn5_check.py time 0 2>&1 | tail -2 | sed "s/^/$L /"
done
unset GC_HIP_LIB
PMC_SETS=1 PMC_TIMEOUT=300 timeout 400 python scripts/pmc.py 'k_attn5' -- python $R/scripts/attn5_check.py time 0 2>&1 | grep -E "BANK_CONFLICT|IDX_ACTIVE|WAIT_INST_LDS"
for L in prev new prev new; do
  if [ $L = prev ]; then export GC_HIP_LIB=$R/gaussctrl_amd/libgaussctrl_hip_prev.so; else unset GC_HIP_LIB; fi
  timeout 600 python bench.py --no-cpu-baseline --no-secondary > $O/bench_$L.json 2> $O/bench_$L.err
  python -c "
import json; d=json.loads(open('$O/bench_$L.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$L', d['value'], d['ms_per_step'], r['frac'], r['avg_launch_us'])"
done
