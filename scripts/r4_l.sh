cd $GRAFT_REPO_ROOT
R=$PWD
O=gpurun_out/r4l; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o bench -- python $R/bench.py --no-cpu-baseline --no-secondary > $R/$O/bench_profiled.json 2> $R/$O/bench_profiled.err)
DB=$(find $O/prof -name "*.db" | head -1)
python scripts/rocpd_stats.py $DB 70 > $O/kernel_stats.txt
rm -rf $O/prof
head -60 $O/kernel_stats.txt | cut -c1-150; tail -2 $O/kernel_stats.txt
