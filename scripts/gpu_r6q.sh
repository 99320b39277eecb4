#!/bin/bash
# round 6, call Q: N > 1 bench paths on ONE GPU over gloo at 4 chunks per launch set (self-launch form + torchrun form); f16 vs bf16 k_attn5 at widening logit spreads;
# default bench + rocprofv3 with the per-launch-size breakdown (roofline.launch_kinds vs rocpd_stats by workgroup count)
cd $GRAFT_REPO_ROOT
R=$PWD
O=gpurun_out/r6q
mkdir -p $O
export TMPDIR=/tmp
export GC_BENCH_ONE_GPU=1 GC_BENCH_BACKEND=gloo
timeout 900 python bench.py --gpus 2 --steps 8 --warmup 2 --denoise-steps 4 --gaussians 200000 --no-secondary --no-cpu-baseline > $O/bench_2_selflaunch.json 2> $O/bench_2_selflaunch.err
echo "self-launch --gpus 2 rc=$?"; tail -1 $O/bench_2_selflaunch.json | cut -c1-260; tail -2 $O/bench_2_selflaunch.err | cut -c1-300
for CFG in "4 rotate" "2 allgather" "8 replicate"; do
  set -- $CFG; N=$1; MODE=$2
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600 + N)) bench.py --gpus $N --steps 8 --warmup 2 --ref-mode $MODE --denoise-steps 4 --gaussians 200000 --no-secondary --no-cpu-baseline > $O/bench_${N}_$MODE.json 2> $O/bench_${N}_$MODE.err
  echo "$N $MODE rc=$?"; tail -1 $O/bench_${N}_$MODE.json | cut -c1-260; tail -2 $O/bench_${N}_$MODE.err | cut -c1-300
done
unset GC_BENCH_ONE_GPU GC_BENCH_BACKEND
python - <<PY 2>&1 | grep -v amdgpu.ids | tee $O/attn5_f16_spread.txt
import sys, torch
sys.argv=["x"]; sys.path.insert(0,"scripts")
import attn5_check as a
for qs in (0.5, 0.7, 0.84, 1.0):
    for dt in (torch.bfloat16, torch.float16):
        a.timing(dt, 0, iters=10, qscale=qs)
PY
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_bf16.json 2> $O/bench_bf16.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o bench -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $R/$O/bench_profiled.json 2> $R/$O/bench_profiled.err)
DB=$(find $O/prof -name "*.db" | head -1)
python scripts/rocpd_stats.py $DB 70 k_attn5 > $O/bench_kernel_stats_bf16.txt
rm -rf $O/prof
tail -12 $O/bench_kernel_stats_bf16.txt | cut -c1-170
python -c "
import json
for f in ('bench_bf16','bench_profiled'):
    d=json.loads(open('$O/'+f+'.json').read().strip().splitlines()[-1]); r=d['roofline']
    print(f, d['value'], d['ms_per_step'], d['mfma_util_step'], r['frac'], r['avg_launch_us']); print(r.get('launch_kinds'))
"
