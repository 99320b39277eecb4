#!/bin/bash
# round 6, call F: chunks per launch set (co-batching) sweep on one box: 2 (default) / 3 / 4 / 7 / 14, in-flight 2 and 1
mkdir -p gpurun_out/r6f
for cfg in "2 2" "3 2" "4 2" "7 2" "7 1" "14 1" "14 2"; do
  set -- $cfg
  timeout 400 python bench.py --steps 28 --warmup 14 --cobatch $1 --inflight $2 --no-secondary --no-cpu-baseline > gpurun_out/r6f/bench_cb$1_if$2.json 2> gpurun_out/r6f/bench_cb$1_if$2.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/r6f/bench_cb$1_if$2.json") if l.startswith("{")][-1])
    print("cobatch $1 inflight $2:", d["value"], "views/s", d["ms_per_step"], "ms/step", "mfma_util", d["mfma_util_step"], "roof", d["roofline"]["frac"])
except Exception as e:
    print("cobatch $1 inflight $2: FAILED", e); print(open("gpurun_out/r6f/bench_cb$1_if$2.err").read()[-1500:])
PY
done
