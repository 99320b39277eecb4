#!/bin/bash
# round 6, call X2: margin of the group-statistics test over random bias / row-vector draws (forced MT 4 and 2)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6x
for mt in 4 2; do timeout 900 python scripts/diag/group_stats_margin.py 2000 $mt 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6x/margin_mt$mt.txt; done
