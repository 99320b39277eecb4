#!/bin/bash
O=gpurun_out/tt; mkdir -p $O
export PYTHONPATH=.
timeout 600 python scripts/ttail_check.py ${TT_DT:-bf16} time > $O/check.log 2>&1
tail -16 $O/check.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ttprof -o tt -- python $GRAFT_REPO_ROOT/scripts/ttail_check.py ${TT_DT:-bf16} time > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'P' > $O/kernels.txt
import csv, glob
f = glob.glob('/tmp/ttprof/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:25]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
P
cat $O/kernels.txt
