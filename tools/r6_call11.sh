#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6c11
mkdir -p $O
python tools/r6_probe_small.py 2>&1 | grep -v amdgpu.ids
python tools/r6_hbm_stages.py 2>&1 | grep adamw
timeout 1500 python -m pytest tests/test_gpu_graph.py tests/test_gpu_convergence.py tests/test_gpu_rccl.py tests/test_gpu_harness.py tests/test_gpu_trainer.py -q -m gpu -p no:cacheprovider > $O/t.log 2>&1; tail -8 $O/t.log
