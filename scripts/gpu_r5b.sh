# Round 5, second GPU call: the batched-views rasterizer (tests, raster-only bench at 1 M / 4 M batched vs one camera per launch set),
# regression of the touched single-view kernels, batch-invariant chunk-8 test, plugin / dist tests, SQ counters of k_gemm8q.
set -x
cd $GRAFT_REPO_ROOT
R=$PWD
O=gpurun_out/${1:-r5b}
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_raster_views_gpu.py -m gpu -q -x 2>&1 | tail -30 > $O/tests_views.log; tail -3 $O/tests_views.log
timeout 1200 python -m pytest tests/test_raster_gpu.py tests/test_plugin_gpu.py -m gpu -q -x 2>&1 | tail -15 > $O/tests_raster.log; tail -2 $O/tests_raster.log
for NG in 1000000 4000000; do
  for MODE in "" "--no-view-batch --chunk-size 3"; do
    T=$(echo "$NG$MODE" | tr -d ' -')
    timeout 600 python bench.py --workload raster --gaussians $NG --steps 24 --warmup 2 --no-cpu-baseline $MODE > $O/raster_$T.json 2> $O/raster_$T.err
    tail -1 $O/raster_$T.json | cut -c1-200
    python - <<PY
import json
d=json.loads(open("$O/raster_$T.json").read().strip().splitlines()[-1])
c=d["roofline"]["chain"]; print("$T", d["value"], c["kernel_us_per_view"], c["frac"], c["frac_processed_pairs"])
for k,v in d["roofline"]["stages"].items(): print("   ", k, v["avg_us"])
PY
  done
done
timeout 900 python -m pytest tests/test_fullgeom_gpu.py -m gpu -q -x -k "invariant" 2>&1 | tail -5 > $O/tests_invariant.log; tail -2 $O/tests_invariant.log
timeout 1500 python -m pytest tests/test_dist_gpu.py -m gpu -q -x 2>&1 | tail -8 > $O/tests_dist.log; tail -2 $O/tests_dist.log
timeout 600 python bench.py --no-cpu-baseline --no-secondary > $O/bench_bf16_quick.json 2> $O/bench_bf16_quick.err; tail -1 $O/bench_bf16_quick.json | cut -c1-300
timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-view-batch > $O/bench_bf16_quick_noviewbatch.json 2> $O/bench_bf16_quick_noviewbatch.err; tail -1 $O/bench_bf16_quick_noviewbatch.json | cut -c1-300
PMC_SETS=0,1 PMC_TIMEOUT=300 timeout 700 python scripts/pmc.py 'k_gemm8q' -- python $R/scripts/bench_fp8.py > $O/pmc_gemm8q_fp8.txt 2>&1
head -50 $O/pmc_gemm8q_fp8.txt
