#!/bin/bash
# round 6, call O: the whole -m gpu suite + smoke on the current build (tap-inner convs, 4 chunks per launch set, clear kernel, k_attn5 in-place swaps)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6o
mkdir -p $O
S=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q --durations=15 > $O/tests.txt 2>&1
echo "suite seconds: $(( $(date +%s) - S ))" | tee -a $O/tests.txt
tail -5 $O/tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
python scripts/margins_summary.py > $O/margins.txt 2>/dev/null; tail -5 $O/margins.txt
