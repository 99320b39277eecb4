# CFG-shared prefix: parity (full-geometry trajectories, both dtypes, invariant mode, dist tests) + A/B; then the N > 1 functional check on one GPU
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5l}
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_fullgeom_gpu.py tests/test_denoise_model_gpu.py -m gpu -q -x -k "not vae and not fp8" 2>&1 | tail -8 > $O/tests_fullgeom.log; tail -3 $O/tests_fullgeom.log
for V in "GC_CFG_SHARE=0" "GC_CFG_SHARE=1" "GC_CFG_SHARE=0" "GC_CFG_SHARE=1"; do
  T=$(echo $V | tr '= ' '__')
  env $V timeout 600 python bench.py --no-cpu-baseline --no-secondary > $O/bench_$T.json 2> $O/bench_$T.err
  python -c "
import json; d=json.loads(open('$O/bench_$T.json').read().strip().splitlines()[-1]); print('$V', d['value'], d['ms_per_step'], d['roofline']['launches'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"
done
timeout 1500 python -m pytest tests/test_dist_gpu.py tests/test_plugin_gpu.py -m gpu -q -x 2>&1 | tail -5 > $O/tests_dist.log; tail -2 $O/tests_dist.log
export GC_BENCH_ONE_GPU=1 GC_BENCH_BACKEND=gloo
for CFG in "2 rotate" "4 rotate" "2 allgather" "8 replicate"; do
  set -- $CFG; N=$1; MODE=$2
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600 + N)) bench.py --gpus $N --steps 4 --warmup 1 --ref-mode $MODE --denoise-steps 4 --gaussians 200000 > $O/bench_${N}_$MODE.json 2> $O/bench_${N}_$MODE.err
  echo "rc=$?"; tail -1 $O/bench_${N}_$MODE.json | cut -c1-200; tail -2 $O/bench_${N}_$MODE.err | cut -c1-300
done
