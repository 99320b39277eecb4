#!/bin/bash
# round 6, call Y: the files of the -m gpu suite up to the forced-variant test, repeated (the in-suite context in which the group-statistics test failed twice)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6y
mkdir -p $O
for i in 1 2 3 4 5 6; do
  S=$(date +%s)
  timeout 900 python -m pytest tests/test_denoise_kernels_gpu.py tests/test_denoise_model_gpu.py tests/test_dist_gpu.py tests/test_fullgeom_gpu.py tests/test_gemm_variants_gpu.py -m gpu -q -x > $O/part_$i.txt 2>&1
  echo "run $i ($(( $(date +%s) - S )) s): $(tail -1 $O/part_$i.txt)"
  grep -h "STATS-MISMATCH" $O/part_$i.txt | head -3 | cut -c1-400
done
