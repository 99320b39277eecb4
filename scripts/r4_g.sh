cd $GRAFT_REPO_ROOT
O=gpurun_out/r4g; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python scripts/gn_parts_bench.py > $O/gn_parts_bench.txt 2>&1; cat $O/gn_parts_bench.txt
timeout 600 python -m pytest tests/test_denoise_kernels_gpu.py -q -x -k "producer_partials" 2>&1 | tail -5
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
