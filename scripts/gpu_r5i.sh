# full -m gpu suite on the current build (batched views, packed binning, fp8 chunk fix, lean LayerNorm fold default) + default bench + host check
set -x
cd $GRAFT_REPO_ROOT
R=$PWD
O=gpurun_out/${1:-r5i}
mkdir -p $O
export TMPDIR=/tmp
GC_TEST_MARGINS=$R/$O/margins_1.jsonl timeout 2700 python -m pytest tests -m gpu -q -x 2>&1 | tail -45 > $O/tests_1.log; tail -3 $O/tests_1.log
timeout 900 python bench.py > $O/bench_bf16.json 2> $O/bench_bf16.err; tail -1 $O/bench_bf16.json | cut -c1-300
timeout 300 python scripts/cpu_bound_check.py 2>&1 | head -40 > $O/cpu_bound.txt; head -4 $O/cpu_bound.txt
