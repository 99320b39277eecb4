# evidence run of the round's HEAD build (k_attn5 with the 16x16x32 O^T rows): suite x1 with margins, smoke, default bench, kernel trace,
# PMC traffic of the dominant kernel.  Sized for ~15 GPU-minutes.
set -x
cd $GRAFT_REPO_ROOT
R=$PWD
O=gpurun_out/${1:-r4head}
mkdir -p $O
export TMPDIR=/tmp
bash scripts/gpu_suite_repeat.sh ${1:-r4head} 1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
tail -3 $O/smoke.log
timeout 600 python bench.py > $O/bench_bf16.json 2> $O/bench_bf16.err
tail -1 $O/bench_bf16.json | cut -c1-400
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o bench -- python $R/bench.py --no-cpu-baseline --no-secondary > $R/$O/bench_bf16_profiled.json 2> $R/$O/bench_profiled.err)
DB=$(find $O/prof -name "*.db" | head -1)
python scripts/rocpd_stats.py $DB 70 > $O/bench_kernel_stats.txt
rm -rf $O/prof
head -12 $O/bench_kernel_stats.txt
timeout 400 python scripts/pmc_kernel_traffic.py 'k_attn5' $O/attn_traffic.json -- python $R/scripts/attn5_check.py time 0 > $O/attn_traffic.log 2>&1
cat $O/attn_traffic.json | head -20
