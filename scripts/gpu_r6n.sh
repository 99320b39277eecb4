#!/bin/bash
# round 6, call N (re-used for every k_attn5 A/B): attention tests, then same-box A/B of two builds
# two builds (the previous libgaussctrl_hip.so kept as attn5_old_build.so.bak), k_attn5 timing at the production launch + default bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6n
mkdir -p $O
timeout 900 python -m pytest tests/test_denoise_kernels_gpu.py -m gpu -q -x -k "attention or attn" > $O/tests.txt 2>&1
tail -2 $O/tests.txt
cp gaussctrl_amd/libgaussctrl_hip.so /tmp/new.so
for rep in 1 2; do
  for b in new old; do
    if [ $b = old ]; then cp gaussctrl_amd/attn5_old_build.so.bak gaussctrl_amd/libgaussctrl_hip.so; else cp /tmp/new.so gaussctrl_amd/libgaussctrl_hip.so; fi
    python - <<PY
import sys, torch
sys.argv=["x"]
sys.path.insert(0,"scripts")
import attn5_check as a
for dt in (torch.bfloat16, torch.float16):
    us=[a.timing(dt, 0, iters=20) for _ in range(3)]
    print("$b", dt, "k_attn5 us", [round(x,1) for x in us])
PY
  done
done 2>&1 | grep -v "^timing" | tee $O/attn5_ab.txt
for b in old new old new; do
  if [ $b = old ]; then cp gaussctrl_amd/attn5_old_build.so.bak gaussctrl_amd/libgaussctrl_hip.so; else cp /tmp/new.so gaussctrl_amd/libgaussctrl_hip.so; fi
  timeout 400 python bench.py --steps 28 --warmup 14 --no-secondary --no-cpu-baseline > $O/bench_$b.json 2> $O/bench_$b.err
  python -c "
import json
d=json.loads([l for l in open('$O/bench_$b.json') if l.startswith('{')][-1])
print('$b:', d['value'], 'views/s', d['ms_per_step'], d['mfma_util_step'], 'attn5 frac', d['roofline']['frac'], d['roofline']['avg_launch_us'])
"
done | tee -a $O/attn5_ab.txt
cp /tmp/new.so gaussctrl_amd/libgaussctrl_hip.so
