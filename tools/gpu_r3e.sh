#!/bin/bash
set -x
O=gpurun_out/r3e; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_conv.py -q --maxfail=20 -p no:cacheprovider -k "halo" > $O/t_halo.log 2>&1; tail -25 $O/t_halo.log | cut -c1-300
timeout 300 python tools/bench_tile4.py > $O/tile4.log 2>&1; cat $O/tile4.log | cut -c1-500
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullsize.py tests/test_gpu_harness.py -q --maxfail=20 -s -p no:cacheprovider > $O/tests.log 2>&1; grep "explained\|passed\|failed\|^FAILED\|^E  " $O/tests.log | cut -c1-300
