#!/bin/bash
set -x
O=gpurun_out/r3b; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_stages.py -q --maxfail=20 -p no:cacheprovider -k "fold or two_threads or stem or whole_postprocess or four_wave" > $O/t_conv_stages.log 2>&1; tail -8 $O/t_conv_stages.log
SKIP_TILES=1 timeout 300 python tools/bench_tile4.py > $O/tile4.log 2>&1; tail -12 $O/tile4.log
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullsize.py -q --maxfail=20 -s -p no:cacheprovider > $O/t_e2e_full.log 2>&1; grep "bf16 bound\|passed\|failed" $O/t_e2e_full.log
timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $O/bench.log 2>&1; tail -c 6000 $O/bench.log
