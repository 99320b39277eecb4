cd $GRAFT_REPO_ROOT
R=$PWD
O=gpurun_out/r4k2; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_denoise_kernels_gpu.py tests/test_gemm_variants_gpu.py -q -x 2>&1 | tail -3
: > $O/gn_kernels.txt
for shp in "6 64 320" "6 32 640"; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o t -- python $R/scripts/gn_kernels_trace.py $shp > /dev/null 2> $R/$O/err.txt)
  DB=$(find $O/prof -name "*.db" | head -1)
  echo "## B H C = $shp" >> $O/gn_kernels.txt
  python scripts/rocpd_stats.py $DB 30 | grep -E "k_gemm|k_splitk" | awk '{printf "%8s calls %9s avg_us  %s\n", $2, $4, $1}' >> $O/gn_kernels.txt
  rm -rf $O/prof
done
cat $O/gn_kernels.txt
run() { name=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-secondary > $O/$name.json 2> $O/$name.err; python - $name $O/$name.json <<'P'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]).read().splitlines() if l.startswith("{")][-1])
    print(f"{sys.argv[1]:28s} views/s {d['value']:.3f}  ms_per_step {d['ms_per_step']:.1f}")
except Exception as e:
    print(sys.argv[1], "FAILED", e)
P
}
run lean_parts A=1
run nolean_noparts GC_GN_PARTS=0 GC_GEMM_DBG=4
run lean_parts2 A=1
run nolean_noparts2 GC_GN_PARTS=0 GC_GEMM_DBG=4
