cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4b
timeout 600 python scripts/tight_box_noise.py > gpurun_out/r4b/tight_box_noise.txt 2>&1
bash scripts/gpu_suite_repeat.sh r4b 2
