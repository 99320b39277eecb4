# Round 5, first GPU call: (1) the new configs[3] fp8 parity tests + falsifiable fp8 bar (no -x: see every verdict); (2) whole -m gpu suite;
# (3) default bench line (bf16 + f16 / fp8 secondaries, fp8 roofline vs 5 PF); (4) rocprofv3 kernel summaries of the bf16 AND the fp8 run;
# (5) one SQ-counter pass over k_gemm8q in the fp8 run.
set -x
cd $GRAFT_REPO_ROOT
R=$PWD
O=gpurun_out/${1:-r5a}
mkdir -p $O
export TMPDIR=/tmp
rocm-smi --showuniqueid 2>/dev/null | grep -i unique | head -1 > $O/box.txt; hostname >> $O/box.txt
GC_TEST_MARGINS=$R/$O/margins_fp8.jsonl timeout 1200 python -m pytest tests/test_fullgeom_gpu.py -m gpu -q -s -k "fp8" 2>&1 | tail -60 > $O/tests_fp8_new.log
tail -3 $O/tests_fp8_new.log
GC_TEST_MARGINS=$R/$O/margins_1.jsonl timeout 2400 python -m pytest tests -m gpu -q -x --deselect tests/test_fullgeom_gpu.py::test_config4_f12_fp8_convs_and_linears 2>&1 | tail -40 > $O/tests_1.log
tail -2 $O/tests_1.log
timeout 900 python bench.py > $O/bench_bf16.json 2> $O/bench_bf16.err
tail -1 $O/bench_bf16.json | cut -c1-400
for DT in bf16 fp8; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$DT -o bench -- python $R/bench.py --dtype $DT --no-cpu-baseline --no-secondary > $R/$O/bench_${DT}_profiled.json 2> $R/$O/bench_${DT}_profiled.err)
  DB=$(find $O/prof_$DT -name "*.db" | head -1)
  python scripts/rocpd_stats.py $DB 70 > $O/bench_kernel_stats_$DT.txt
  rm -rf $O/prof_$DT
  head -14 $O/bench_kernel_stats_$DT.txt
done
PMC_SETS=0,1 PMC_TIMEOUT=400 timeout 900 python scripts/pmc.py 'k_gemm8q' -- python $R/bench.py --dtype fp8 --no-cpu-baseline --no-secondary --steps 3 --warmup 0 > $O/pmc_gemm8q_fp8.txt 2>&1
head -40 $O/pmc_gemm8q_fp8.txt
