# end-of-round evidence run (one box): suite x2 with margins, default bench, kernel trace, PMC traffic of the dominant kernel, dtype / config
# variants, raster workload, N > 1 on one GPU (owner broadcast and all-gather modes), SQ counters of the row-resident kernels
set -x
cd $GRAFT_REPO_ROOT
R=$PWD
O=gpurun_out/${1:-r4f}
mkdir -p $O
export TMPDIR=/tmp
bash scripts/gpu_suite_repeat.sh ${1:-r4f} 2
timeout 900 python bench.py > $O/bench_bf16.json 2> $O/bench_bf16.err
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o bench -- python $R/bench.py --no-cpu-baseline --no-secondary > $R/$O/bench_bf16_profiled.json 2> $R/$O/bench_profiled.err)
DB=$(find $O/prof -name "*.db" | head -1)
python scripts/rocpd_stats.py $DB 70 > $O/bench_kernel_stats.txt
rm -rf $O/prof
timeout 900 python scripts/pmc_kernel_traffic.py 'k_attn5' $O/attn_traffic.json -- python $R/scripts/attn5_check.py time 0 > $O/attn_traffic.log 2>&1
timeout 600 python bench.py --dtype f16 --no-cpu-baseline --no-secondary > $O/bench_f16.json 2> $O/bench_f16.err
timeout 600 python bench.py --dtype fp8 --no-cpu-baseline --no-secondary > $O/bench_fp8.json 2> $O/bench_fp8.err
timeout 600 python bench.py --chunk-size 8 --mask --no-cpu-baseline --no-secondary > $O/bench_config4_bf16.json 2> $O/bench_config4_bf16.err
timeout 600 python bench.py --chunk-size 8 --mask --dtype fp8 --no-cpu-baseline --no-secondary > $O/bench_config4_fp8.json 2> $O/bench_config4_fp8.err
timeout 600 python bench.py --gaussians 2000000 --views 10 --no-cpu-baseline --no-secondary > $O/bench_config3_one_shard.json 2> $O/bench_config3.err
timeout 600 python bench.py --inflight 3 --no-cpu-baseline --no-secondary > $O/bench_bf16_inflight3.json 2> $O/bench_inflight3.err
GC_GN_PARTS=0 GC_GEMM_DBG=4 timeout 600 python bench.py --no-cpu-baseline --no-secondary > $O/bench_bf16_round3_kernels.json 2> $O/bench_round3_kernels.err
timeout 600 python bench.py --no-cpu-baseline --no-secondary > $O/bench_bf16_again.json 2> $O/bench_again.err
PYTHONPATH=$R timeout 900 python scripts/pmc.py 'k_thead' -- python $R/scripts/thead_check.py > $O/thead_pmc.txt 2>&1
timeout 600 python bench.py --workload raster --gaussians 1000000 --steps 20 --warmup 2 > $O/raster_1m.json 2> $O/raster_1m.err
timeout 600 python bench.py --workload raster --gaussians 4000000 --steps 20 --warmup 2 --no-cpu-baseline > $O/raster_4m.json 2> $O/raster_4m.err
GC_BENCH_ONE_GPU=1 GC_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 14 --warmup 1 --no-cpu-baseline > $O/bench2_gloo.json 2> $O/bench2_gloo.err
GC_BENCH_ONE_GPU=1 GC_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29514 bench.py --gpus 2 --steps 14 --warmup 1 --ref-mode allgather --no-cpu-baseline > $O/bench2_gloo_allgather.json 2> $O/bench2_gloo_allgather.err
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
tail -3 $O/smoke.log
python - $O <<'P'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith("{")][-1])
        r=d.get("roofline") or {}
        print(f.split("/")[-1], d.get("value"), d.get("ms_per_step"), d.get("mfma_util_step"), r.get("frac"), r.get("avg_launch_us"), (d.get("secondary") or {}).get("value"), (d.get("roofline_raster") or r.get("chain") or {}).get("frac"))
    except Exception as e: print(f, "ERR", str(e)[:80])
P
