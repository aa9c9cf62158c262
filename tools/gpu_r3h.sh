#!/bin/bash
set -x
O=gpurun_out/r3h; mkdir -p $O
export TMPDIR=/tmp
SKIP_REST=1 timeout 300 python tools/bench_tile4.py > $O/tile4.log 2>&1; grep -v amdgpu $O/tile4.log | cut -c1-500
timeout 300 python -m pytest tests/test_gpu_conv.py -q -p no:cacheprovider -k "halo or ragged" > $O/t.log 2>&1; tail -3 $O/t.log
