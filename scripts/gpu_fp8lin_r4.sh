# fp8 transformer linears (weights.add_fp8_linears): unit + full-geometry parity, then the same-box A/B of `--dtype fp8` with and without them
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r4fp8lin}
mkdir -p $O
export TMPDIR=/tmp
GC_TEST_MARGINS=$PWD/$O/margins.jsonl timeout 900 python -m pytest tests -m gpu -q -s -k "layernorm_fp8 or linear_fp8 or fp8_convs_and_linears or test_abi" 2>&1 | grep -v "amdgpu.ids" | tail -40 > $O/tests.log
tail -15 $O/tests.log
timeout 300 python bench.py --dtype fp8 --no-cpu-baseline --no-secondary > $O/bench_fp8_convs.json 2> $O/bench_fp8_convs.err
timeout 300 python bench.py --dtype fp8 --fp8-linears 7 --no-cpu-baseline --no-secondary > $O/bench_fp8_lin7.json 2> $O/bench_fp8_lin7.err
timeout 300 python bench.py --dtype fp8 --fp8-linears 1 --no-cpu-baseline --no-secondary > $O/bench_fp8_lin1.json 2> $O/bench_fp8_lin1.err
timeout 300 python bench.py --no-cpu-baseline --no-secondary > $O/bench_bf16.json 2> $O/bench_bf16.err
timeout 300 python bench.py --dtype fp8 --no-cpu-baseline --no-secondary > $O/bench_fp8_convs_b.json 2> $O/bench_fp8_convs_b.err
timeout 300 python bench.py --dtype fp8 --fp8-linears 7 --no-cpu-baseline --no-secondary > $O/bench_fp8_lin7_b.json 2> $O/bench_fp8_lin7_b.err
timeout 300 python bench.py --chunk-size 8 --mask --no-cpu-baseline --no-secondary > $O/bench_config4_bf16.json 2> $O/bench_config4_bf16.err
timeout 300 python bench.py --chunk-size 8 --mask --dtype fp8 --no-cpu-baseline --no-secondary > $O/bench_config4_fp8_convs.json 2> $O/bench_config4_fp8_convs.err
timeout 300 python bench.py --chunk-size 8 --mask --dtype fp8 --fp8-linears 7 --no-cpu-baseline --no-secondary > $O/bench_config4_fp8_lin7.json 2> $O/bench_config4_fp8_lin7.err
GC_BATCH_INVARIANT=1 timeout 300 python bench.py --no-cpu-baseline --no-secondary > $O/bench_bf16_batch_invariant.json 2> $O/bench_bf16_batch_invariant.err
for f in $O/bench_*.json; do echo $f; tail -1 $f | cut -c1-160; done
for f in $O/*.err; do tail -n 3 $f | cut -c1-300; done
