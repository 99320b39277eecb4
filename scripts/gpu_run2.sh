set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3b/tests.log
tail -5 gpurun_out/r3b/tests.log
timeout 600 python bench.py --steps 14 --warmup 1 --no-cpu-baseline > gpurun_out/r3b/bench_attn5.json 2> gpurun_out/r3b/bench_attn5.err
GC_ATTN_V=4 timeout 600 python bench.py --steps 14 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/r3b/bench_attn4.json 2> gpurun_out/r3b/bench_attn4.err
timeout 600 python bench.py --steps 14 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/r3b/bench_attn5_b.json 2> gpurun_out/r3b/bench_attn5_b.err
python - <<'P'
import json
for n in ("bench_attn5","bench_attn4","bench_attn5_b"):
    try:
        d=json.loads(open(f"gpurun_out/r3b/{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d["roofline"]["kernel"][:40], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d.get("secondary"))
    except Exception as e: print(n, "ERR", e)
P
