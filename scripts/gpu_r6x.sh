#!/bin/bash
# round 6, call X: stress of the group-statistics epilogue on the shapes that failed (forced MT 2 / 4 / 3 / auto), one process at a time
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6x
mkdir -p $O
for mt in 4 2 3 0; do
  timeout 900 python scripts/diag/group_stats_stress.py 4000 $mt > $O/stress_mt$mt.txt 2>&1
  echo "mt $mt: $(tail -1 $O/stress_mt$mt.txt)"; grep -c "STATS\|OUTPUT" $O/stress_mt$mt.txt; grep "STATS\|OUTPUT" $O/stress_mt$mt.txt | head -5
done
