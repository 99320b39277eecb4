#!/bin/bash
# round 6, call A: the -m gpu suite with the new full-size batched-views tests and the suite-time cuts, then the co-batching A/B on the same box
mkdir -p gpurun_out/r6a
python -m pytest tests -m gpu -q --durations=40 > gpurun_out/r6a/tests.txt 2>&1
tail -4 gpurun_out/r6a/tests.txt
for cb in 1 2; do
  python bench.py --steps 20 --warmup 5 --cobatch $cb --no-secondary --no-cpu-baseline > gpurun_out/r6a/bench_cobatch$cb.json 2> gpurun_out/r6a/bench_cobatch$cb.err
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r6a/bench_cobatch$cb.json") if l.startswith("{")][-1])
print("cobatch $cb:", d["value"], "views/s", d["ms_per_step"], "ms/step", "mfma_util", d["mfma_util_step"], "roof", d["roofline"]["frac"], d["roofline"]["kernel"][:30])
for k,v in d["roofline"]["other"].items(): print("   ", k, v)
PY
done
python bench.py --steps 20 --warmup 5 --cobatch 2 --inflight 1 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cobatch 2 inflight 1:', d['value'])"
python bench.py --steps 20 --warmup 5 --cobatch 4 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cobatch 4:', d['value'])"
