#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5o
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_harness.py -q -m gpu -k "graph or bench" > $O/t.log 2>&1; tail -5 $O/t.log
