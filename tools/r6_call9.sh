#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6c9
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_conv.py -q -m gpu -p no:cacheprovider -x -k "maxpool" > $O/t.log 2>&1; tail -3 $O/t.log
echo "--- general"; python tools/r6_hbm_stages.py 0 2>&1 | grep -i "pool"
echo "--- fast"; python tools/r6_hbm_stages.py 1 2>&1 | tail -20
