set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_dist_gpu.py -m gpu -q -x 2>&1 | grep -v "^  x0" | tail -6
