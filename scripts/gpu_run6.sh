set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3g
run() { # name, env..., args
  name=$1; shift
  env "$@" > /dev/null 2>&1
}
for cfg in "A 1 2" "B 0 2" "C 0 3" "D 0 4" "E 1 3" "F 1 4" "A2 1 2"; do
  set -- $cfg
  GC_DN_STREAMS=$2 timeout 600 python bench.py --steps 14 --warmup 1 --no-cpu-baseline --no-secondary --inflight $3 > gpurun_out/r3g/bench_$1.json 2> gpurun_out/r3g/bench_$1.err
  python - <<P
import json
try:
    d=json.loads([l for l in open("gpurun_out/r3g/bench_$1.json").read().strip().splitlines() if l.startswith("{")][-1])
    print("GC_DN_STREAMS=$2 inflight=$3:", d["value"], d["ms_per_step"])
except Exception as e:
    print("$1 ERR", e); print(open("gpurun_out/r3g/bench_$1.err").read()[-600:])
P
done
