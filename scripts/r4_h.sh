cd $GRAFT_REPO_ROOT
R=$PWD
O=gpurun_out/r4h; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_denoise_kernels_gpu.py -q -x -k "producer_partials" 2>&1 | tail -3
run() { name=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-secondary > $O/$name.json 2> $O/$name.err; python - $name $O/$name.json <<'P'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]).read().splitlines() if l.startswith("{")][-1])
    print(f"{sys.argv[1]:28s} views/s {d['value']:.3f}  ms_per_step {d['ms_per_step']:.1f}")
except Exception as e:
    print(sys.argv[1], "FAILED", e)
P
}
run parts_on A=1
run parts_off GC_GN_PARTS=0
run parts_on2 A=1
run parts_off2 GC_GN_PARTS=0
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o bench -- python $R/bench.py --no-cpu-baseline --no-secondary > $R/$O/bench_profiled.json 2> $R/$O/bench_profiled.err)
DB=$(find $O/prof -name "*.db" | head -1)
python scripts/rocpd_stats.py $DB 60 > $O/kernel_stats_parts_on.txt
rm -rf $O/prof
head -45 $O/kernel_stats_parts_on.txt | cut -c1-150
