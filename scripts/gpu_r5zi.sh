set -x
cd $GRAFT_REPO_ROOT
R=$PWD
O=gpurun_out/${1:-r5zi}
mkdir -p $O
export TMPDIR=/tmp
GC_TEST_MARGINS=$R/$O/margins.jsonl timeout 2700 python -m pytest tests -m gpu -q -x 2>&1 | tail -40 > $O/tests.log
tail -2 $O/tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
