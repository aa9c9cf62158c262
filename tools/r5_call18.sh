#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5t
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_harness.py -q -m gpu -k "bench" > $O/t.log 2>&1; tail -4 $O/t.log
