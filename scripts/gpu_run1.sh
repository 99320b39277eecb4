set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
timeout 1500 python -m pytest tests -m gpu -x -q -k "production_geometry or vae_encode_h512 or config2_bear or image2latent or dist" 2>&1 | tail -25 > gpurun_out/r3a/tests_new.log
timeout 600 python bench.py --steps 14 --warmup 1 > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err
tail -5 gpurun_out/r3a/bench.err
GC_BENCH_ONE_GPU=1 GC_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 8 --warmup 1 > gpurun_out/r3a/bench2_gloo.json 2> gpurun_out/r3a/bench2_gloo.err
tail -5 gpurun_out/r3a/bench2_gloo.err
GC_BENCH_ONE_GPU=1 GC_BENCH_BACKEND=nccl timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 4 --warmup 1 > gpurun_out/r3a/bench2_nccl.json 2> gpurun_out/r3a/bench2_nccl.err
tail -5 gpurun_out/r3a/bench2_nccl.err
cat gpurun_out/r3a/tests_new.log
