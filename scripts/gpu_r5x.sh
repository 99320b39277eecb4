# unsplit head-size-160 attention (GC_BATCH_INVARIANT=1 never offers the set-split workspace) with / without the tile prefetch
set -x
cd $GRAFT_REPO_ROOT
R=$PWD
O=gpurun_out/${1:-r5x}
mkdir -p $O
export TMPDIR=/tmp
for L in prev new; do
  if [ $L = prev ]; then export GC_HIP_LIB=$R/gaussctrl_amd/libgaussctrl_hip_prev.so; else unset GC_HIP_LIB; fi
  GC_BATCH_INVARIANT=1 timeout 600 python scripts/bench_kernels.py attn > $O/ubench_attn_unsplit_$L.txt 2>&1
done
unset GC_HIP_LIB
paste -d'|' $O/ubench_attn_unsplit_prev.txt $O/ubench_attn_unsplit_new.txt | awk -F'|' '{split($1,a,":"); split($2,b,":"); print a[1] ":" substr(a[2],1,24) " |" substr(b[2],1,24)}'
