#!/bin/bash
O=gpurun_out/all; mkdir -p $O
export PYTHONPATH=.
timeout 2400 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log
