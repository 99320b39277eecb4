# Review item 1b: the whole `-m gpu` suite N times on this box, margins recorded; tails + margin summary under gpurun_out/$1
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-suite}; N=${2:-3}; mkdir -p $O
export TMPDIR=/tmp
rocm-smi --showuniqueid 2>/dev/null | grep -i unique | head -1 > $O/box.txt; hostname >> $O/box.txt
for i in $(seq 1 $N); do
  GC_TEST_MARGINS=$PWD/$O/margins_$i.jsonl timeout 2700 python -m pytest tests -m gpu -q -x 2>&1 | tail -60 > $O/tests_$i.log
  tail -2 $O/tests_$i.log
done
python scripts/margins_summary.py $O/margins_*.jsonl > $O/margins_summary.txt
head -40 $O/margins_summary.txt
