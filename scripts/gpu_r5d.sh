# Round 5, fourth GPU call: dist tests (duplicate-view fix), fp8 kernel tests + A/B of the conflict-free fp8 fragment chunks (previous build =
# gaussctrl_amd/libgaussctrl_hip_prev.so through GC_HIP_LIB), SQ counters of k_gemm8q after the fix, PMC HBM traffic of the batched-views rasterizer.
set -x
cd $GRAFT_REPO_ROOT
R=$PWD
O=gpurun_out/${1:-r5d}
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_dist_gpu.py -m gpu -q -x 2>&1 | tail -30 > $O/tests_dist.log; tail -3 $O/tests_dist.log
timeout 900 python -m pytest tests/test_denoise_kernels_gpu.py -m gpu -q -x -k "fp8" 2>&1 | tail -8 > $O/tests_fp8_kernels.log; tail -2 $O/tests_fp8_kernels.log
timeout 900 python -m pytest tests/test_fullgeom_gpu.py -m gpu -q -x -k "fp8" 2>&1 | tail -8 > $O/tests_fp8_fullgeom.log; tail -2 $O/tests_fp8_fullgeom.log
for L in prev new prev new; do
  if [ $L = prev ]; then export GC_HIP_LIB=$R/gaussctrl_amd/libgaussctrl_hip_prev.so; else unset GC_HIP_LIB; fi
  timeout 600 python bench.py --dtype fp8 --no-cpu-baseline --no-secondary > $O/bench_fp8_$L.json 2> $O/bench_fp8_$L.err
  python -c "
import json; d=json.loads(open('$O/bench_fp8_$L.json').read().strip().splitlines()[-1]); print('$L', d['value'], d.get('mfma_util_step_mixed_peak'), (d.get('roofline_fp8') or {}).get('frac'), (d.get('roofline_fp8') or {}).get('kernel'))"
done
unset GC_HIP_LIB
PYTHONPATH=$R PMC_SETS=0,1 PMC_TIMEOUT=300 timeout 700 python scripts/pmc.py 'k_gemm8q' -- python $R/scripts/bench_fp8.py > $O/pmc_gemm8q_fp8.txt 2>&1
grep -E "^##|BANK_CONFLICT|IDX_ACTIVE|WAVE_CYCLES" $O/pmc_gemm8q_fp8.txt
timeout 900 python scripts/pmc_traffic.py 1000000 16 $O/raster_traffic_views8_1m.json 8 > $O/raster_traffic_1m.log 2>&1; tail -30 $O/raster_traffic_1m.log | head -40
timeout 900 python scripts/pmc_traffic.py 4000000 16 $O/raster_traffic_views8_4m.json 8 > $O/raster_traffic_4m.log 2>&1; tail -5 $O/raster_traffic_4m.log
timeout 900 python bench.py --no-cpu-baseline > $O/bench_bf16_secondaries.json 2> $O/bench_bf16_secondaries.err; tail -1 $O/bench_bf16_secondaries.json | cut -c1-300
