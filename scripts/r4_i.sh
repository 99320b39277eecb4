cd $GRAFT_REPO_ROOT
R=$PWD
O=gpurun_out/r4i; mkdir -p $O
export TMPDIR=/tmp
: > $O/gn_kernels.txt
for shp in "6 64 320" "6 32 640" "6 16 1280" "6 32 320"; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o t -- python $R/scripts/gn_kernels_trace.py $shp > /dev/null 2> $R/$O/err.txt)
  DB=$(find $O/prof -name "*.db" | head -1)
  echo "## B H C = $shp" >> $O/gn_kernels.txt
  python scripts/rocpd_stats.py $DB 30 | grep -E "k_gn|k_concat|k_gemm|k_splitk" | awk '{printf "%8s calls %9s avg_us  %s\n", $2, $4, $1}' >> $O/gn_kernels.txt
  rm -rf $O/prof
done
cat $O/gn_kernels.txt
