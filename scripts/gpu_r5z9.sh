set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5z9}
mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_fullgeom_gpu.py -m gpu -q -s -k "config4_f12_fp8" 2>&1 | grep -v "^  x0" | tail -20 > $O/tests_c4.log; tail -20 $O/tests_c4.log
timeout 900 python bench.py --dtype fp8 --chunk-size 8 --mask --no-cpu-baseline --no-secondary > $O/bench_c4_fp8.json 2> $O/bench_c4_fp8.err; tail -1 $O/bench_c4_fp8.json | cut -c1-200
timeout 900 python bench.py --chunk-size 8 --mask --no-cpu-baseline --no-secondary > $O/bench_c4_bf16.json 2> $O/bench_c4_bf16.err; tail -1 $O/bench_c4_bf16.json | cut -c1-200
