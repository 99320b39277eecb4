cd $GRAFT_REPO_ROOT
O=gpurun_out/r4e; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_denoise_kernels_gpu.py -q -x -k "producer_partials or conv3x3 or linear" 2>&1 | tail -25 > $O/t1.log; tail -25 $O/t1.log
timeout 900 python -m pytest tests/test_fullgeom_gpu.py tests/test_denoise_model_gpu.py tests/test_ttail_gpu.py -q -x 2>&1 | tail -25 > $O/t2.log; tail -12 $O/t2.log
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
