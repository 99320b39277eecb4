#!/bin/bash
# round 6, call W: reproduce the intermittent failure of test_conv_output_group_statistics in the forced-MT children (full logs kept)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6w
mkdir -p $O
for i in 1 2 3 4 5 6 7 8; do
  for mt in 4 2; do
    GC_GEMM_MT=$mt OMP_NUM_THREADS=16 MKL_NUM_THREADS=16 timeout 600 python -m pytest tests/test_denoise_kernels_gpu.py -x -q -p no:cacheprovider -k "group_statistics or output_statistics" --tb=long > $O/mt${mt}_$i.txt 2>&1
    echo "mt $mt run $i: $(tail -1 $O/mt${mt}_$i.txt)"
  done
done
grep -l "failed" $O/*.txt | head
