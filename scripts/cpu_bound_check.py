"""Is the denoise chunk CPU-launch-bound?  Host enqueue time (Python returns, nothing synchronised) vs GPU time of one 20-step chunk of
3 views on bench.py's networks, with the number of C-ABI launches.  python scripts/cpu_bound_check.py"""
import sys, time
import torch
sys.argv = [sys.argv[0], "--no-cpu-baseline", "--no-secondary"]
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gaussctrl_amd import _lib

args = bench.parse()
args.views = 40; args.chunk_size = args.chunk_size or 3
dev = torch.device("cuda", 0)
B = bench.Bench(args, "bf16", 0, 1, dev, None, None)
B.run(1, 1)                                    # setup + warm
pipe, st = B.pipe, B.state
c = B.c
z0 = B.z0[:c]; disp = torch.rand(c, 3, 512, 512, device=dev)
lib = _lib.lib()
n_calls = [0]
for _ in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lat = pipe.edit_chunk_cached(z0, disp, B.ctx_neg, B.ctx_pos, st["bank"])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"edit_chunk_cached ({c} views, 20 steps): host enqueue {1e3 * (t1 - t0):.1f} ms, until the GPU is done {1e3 * (t2 - t0):.1f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
lat = pipe.edit_chunk_cached(z0, disp, B.ctx_neg, B.ctx_pos, st["bank"])
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
