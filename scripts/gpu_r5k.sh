# N > 1 code path on one GPU over gloo (functional only): 2 and 4 ranks, rotate and allgather reference modes, after the batched-views change
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5k}
mkdir -p $O
export TMPDIR=/tmp GC_BENCH_ONE_GPU=1 GC_BENCH_BACKEND=gloo
for CFG in "2 rotate" "4 rotate" "2 allgather" "8 replicate"; do
  set -- $CFG; N=$1; MODE=$2
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600 + N)) bench.py --gpus $N --steps 4 --warmup 1 --ref-mode $MODE --denoise-steps 4 --gaussians 200000 > $O/bench_${N}_$MODE.json 2> $O/bench_${N}_$MODE.err
  echo "rc=$?"; tail -1 $O/bench_${N}_$MODE.json | cut -c1-260; tail -3 $O/bench_${N}_$MODE.err | cut -c1-300
done
