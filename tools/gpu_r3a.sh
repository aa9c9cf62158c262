#!/bin/bash
# round-3 GPU session A: new-kernel parity, stage / full-size fixtures, tile A/B, bench
set -x
O=gpurun_out/r3a; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_stages.py -q --maxfail=20 -p no:cacheprovider > $O/t_conv_stages.log 2>&1; tail -15 $O/t_conv_stages.log
timeout 300 python tools/bench_tile4.py > $O/tile4.log 2>&1; cat $O/tile4.log
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullsize.py -q --maxfail=20 -s -p no:cacheprovider > $O/t_e2e_full.log 2>&1; tail -25 $O/t_e2e_full.log
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/bench.log 2>&1; tail -c 3000 $O/bench.log
timeout 200 python tools/find_copies.py > $O/copies.log 2>&1; head -40 $O/copies.log
