#!/bin/bash
# round 6, call S: achieved HBM rate of the GroupNorm apply kernel at the launch-set shapes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6s
timeout 600 python scripts/gn_apply_bw.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6s/gn_apply_bw.txt
