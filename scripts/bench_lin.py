import sys, torch
sys.path.insert(0, '.')
from gaussctrl_amd.sd import ops
from scripts.bench_kernels import timeit  # noqa
