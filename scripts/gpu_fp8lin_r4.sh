# fp8 path (weights.add_fp8_convs / add_fp8_linears): unit + full-geometry parity, then same-box A/Bs of `--dtype fp8`
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r4fp8lin}
mkdir -p $O
export TMPDIR=/tmp
GC_TEST_MARGINS=$PWD/$O/margins.jsonl timeout 900 python -m pytest tests -m gpu -q -s -k "fp8" 2>&1 | grep -v "amdgpu.ids" | tail -40 > $O/tests.log
tail -15 $O/tests.log
B="timeout 300 python bench.py --no-cpu-baseline --no-secondary"
$B > $O/bench_bf16.json 2> $O/bench_bf16.err

$B --dtype fp8 --fp8-min-hw 1024 > $O/bench_fp8_lin7_hw1024.json 2> $O/bench_fp8_lin7_hw1024.err
$B --dtype fp8 > $O/bench_fp8_lin7_hw256.json 2> $O/bench_fp8_lin7_hw256.err

$B --dtype fp8 > $O/bench_fp8_lin7_hw256_b.json 2> $O/bench_fp8_lin7_hw256_b.err
$B --chunk-size 8 --mask --dtype fp8 > $O/bench_config4_fp8.json 2> $O/bench_config4_fp8.err
$B --chunk-size 8 --mask > $O/bench_config4_bf16.json 2> $O/bench_config4_bf16.err
$B --dtype fp8 --fp8-min-hw 64 > $O/bench_fp8_lin7_hw64.json 2> $O/bench_fp8_lin7_hw64.err
for f in $O/bench_*.json; do echo $f; tail -1 $f | cut -c1-160; done
for f in $O/*.err; do tail -n 3 $f | grep -v amdgpu.ids | cut -c1-300; done
