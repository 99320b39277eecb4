set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4a; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python scripts/tight_box_noise.py > $O/tight_box_noise.txt 2> $O/tight_box_noise.err
tail -40 $O/tight_box_noise.txt; tail -5 $O/tight_box_noise.err
