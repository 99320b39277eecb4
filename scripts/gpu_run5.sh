set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3e
timeout 1800 python -m pytest tests -m gpu -x -q -k "raster or fused or model_get_outputs or accumulates or loss or pipeline_flow or tight" 2>&1 | tail -12 > gpurun_out/r3e/tests.log
cat gpurun_out/r3e/tests.log
timeout 600 python bench.py --workload raster --gaussians 1000000 --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/r3e/raster_1m.json 2> gpurun_out/r3e/raster_1m.err
timeout 600 python bench.py --workload raster --gaussians 4000000 --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/r3e/raster_4m.json 2> gpurun_out/r3e/raster_4m.err
python - <<'P'
import json
for n in ("raster_1m","raster_4m"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/r3e/{n}.json").read().strip().splitlines() if l.startswith("{")][-1])
        r=d["roofline"]; print(n, d["value"], r["chain"]["kernel_us_per_view"], r["chain"]["frac"], r["chain"]["M_mean"])
        for k,v in r["stages"].items(): print("   ",k, v["avg_us"])
    except Exception as e: print(n, "ERR", e); print(open(f"gpurun_out/r3e/{n}.err").read()[-800:])
P
