cd $GRAFT_REPO_ROOT
O=gpurun_out/r4d; mkdir -p $O
export TMPDIR=/tmp
run() { name=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-secondary > $O/$name.json 2> $O/$name.err; python - $name $O/$name.json <<'P'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]).read().splitlines() if l.startswith("{")][-1])
    print(f"{sys.argv[1]:28s} views/s {d['value']:.3f}  ms_per_step {d['ms_per_step']:.1f}")
except Exception as e:
    print(sys.argv[1], "FAILED", e)
P
}
run base A=1
run fuse_gn GC_DN_FUSE_GN=1
run fuse_gn_noglobalatomics GC_DN_FUSE_GN=1 GC_GEMM_DBG=1
run fuse_gn_nolds GC_DN_FUSE_GN=1 GC_GEMM_DBG=3
run fuse_gn_nodpp GC_DN_FUSE_GN=1 GC_GEMM_DBG=7
run base2 A=1
run gn2 GC_DN_GN2=1
