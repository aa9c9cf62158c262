#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5r
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_e2e.py tests/test_gpu_trainer.py -q -m gpu -k "two_taps or train_matches or trainer or arena" > $O/t.log 2>&1; tail -6 $O/t.log
