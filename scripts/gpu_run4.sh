set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3d
for n in 1 2 3 2 1; do
timeout 600 python bench.py --steps 14 --warmup 1 --no-cpu-baseline --no-secondary --inflight $n > gpurun_out/r3d/bench_inflight$n.json 2> gpurun_out/r3d/bench_inflight$n.err
python - <<P
import json
try:
    d=json.loads([l for l in open("gpurun_out/r3d/bench_inflight$n.json").read().strip().splitlines() if l.startswith("{")][-1])
    print("inflight $n", d["value"], d["ms_per_step"], d["mfma_util_step"])
except Exception as e:
    print("inflight $n ERR", e); print(open("gpurun_out/r3d/bench_inflight$n.err").read()[-1500:])
P
done
