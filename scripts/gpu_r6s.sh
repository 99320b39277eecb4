#!/bin/bash
# round 6, call S3: stream configuration at 4 chunks per launch set, same box: default (2 sets in flight x ControlNet || UNet), single-stream denoise, 3 sets in flight
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6s
mkdir -p $O
run() { # name, env, args
  env $2 timeout 400 python bench.py --steps 28 --warmup 14 --no-secondary --no-cpu-baseline $3 > $O/bench_$1.json 2> $O/bench_$1.err
  python -c "
import json
d=json.loads([l for l in open('$O/bench_$1.json') if l.startswith('{')][-1]); print('$1:', d['value'], 'views/s', d['ms_per_step'])"
}
run default "X=1" ""
run dn_single_stream "GC_DN_STREAMS=0" ""
run inflight3 "X=1" "--inflight 3"
run inflight3_single "GC_DN_STREAMS=0" "--inflight 3"
run default2 "X=1" ""
