set -x
cd $GRAFT_REPO_ROOT
R=$PWD
mkdir -p gpurun_out/r3c
export TMPDIR=/tmp
python scripts/attn5_check.py abl 2>&1 | grep "timing\|ablation" > gpurun_out/r3c/attn5_ablation.txt
timeout 1200 python -m pytest tests -m gpu -x -q -k "dist_gpu" 2>&1 | tail -6 > gpurun_out/r3c/tests_raster.log
cat gpurun_out/r3c/tests_raster.log
GC_BENCH_ONE_GPU=1 GC_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 8 --warmup 1 > gpurun_out/r3c/bench2_gloo.json 2> gpurun_out/r3c/bench2_gloo.err
tail -3 gpurun_out/r3c/bench2_gloo.err
GC_BENCH_ONE_GPU=1 GC_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 4 --steps 4 --warmup 0 > gpurun_out/r3c/bench4_gloo.json 2> gpurun_out/r3c/bench4_gloo.err
tail -3 gpurun_out/r3c/bench4_gloo.err
# HBM traffic of the dominant kernel (separate FETCH_SIZE / WRITE_SIZE passes)
timeout 900 python scripts/pmc_kernel_traffic.py 'k_attn5' gpurun_out/r3c/attn_traffic.json -- python $R/scripts/attn5_check.py time 0 > gpurun_out/r3c/attn_traffic.log 2>&1
tail -12 gpurun_out/r3c/attn_traffic.log
# kernel trace of the default bench
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3c/prof -o bench -- python $R/bench.py --steps 14 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/r3c/bench_profiled.json 2> $R/gpurun_out/r3c/bench_profiled.err)
ls -R gpurun_out/r3c/prof | head -20
DB=$(find gpurun_out/r3c/prof -name "*.db" | head -1)
if [ -n "$DB" ]; then python scripts/rocpd_stats.py $DB 70 > gpurun_out/r3c/bench_kernel_stats.txt; else find gpurun_out/r3c/prof -name "*kernel_stats*" | head; fi
head -30 gpurun_out/r3c/bench_kernel_stats.txt
rm -rf gpurun_out/r3c/prof
timeout 600 python bench.py --workload raster --gaussians 1000000 --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/r3c/raster_1m.json 2> gpurun_out/r3c/raster_1m.err
timeout 600 python bench.py --workload raster --gaussians 4000000 --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/r3c/raster_4m.json 2> gpurun_out/r3c/raster_4m.err
python - <<'P'
import json
for n in ("bench2_gloo","bench4_gloo","raster_1m","raster_4m","bench_profiled"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/r3c/{n}.json").read().strip().splitlines() if l.startswith("{")][-1])
        print(n, d["value"], d["ms_per_step"], d["config"].get("parallelism","")[:100], (d.get("roofline") or {}).get("chain"))
    except Exception as e: print(n, "ERR", e)
P
