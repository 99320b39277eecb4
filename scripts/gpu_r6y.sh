#!/bin/bash
# round 6, call Y2: the whole -m gpu suite twice with seeded tests and the two-form statistics bar (+ the margins summary)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6y
mkdir -p $O
for i in 1 2; do
  S=$(date +%s)
  timeout 1500 python -m pytest tests -m gpu -q -x --durations=5 > $O/full_$i.txt 2>&1
  echo "suite $i seconds: $(( $(date +%s) - S ))" | tee -a $O/full_$i.txt
  grep -E "passed|failed" $O/full_$i.txt | tail -1
  grep -E "STATS-MISMATCH|^FAILED" $O/full_$i.txt | head -5 | cut -c1-300
done
grep -A12 "^margins:" $O/full_2.txt | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
