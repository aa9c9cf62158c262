#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5l
mkdir -p $O
root=$PWD
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_fullsize.py tests/test_gpu_trainer.py -q -m gpu -k "batchnorm or deviates_like or bit_identical or arena" > $O/t.log 2>&1; tail -4 $O/t.log
(cd /tmp && NRPN_WGRAD_STREAM=0 timeout 90 rocprofv3 --kernel-trace --stats -d /tmp/prof -o p --output-format csv -- python $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-probe --no-extras > $root/$O/prof_bench.log 2>&1)
cp $(find /tmp/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats_single_stream.csv
grep -E "finalize|chan_partial|bn_" $O/kernel_stats_single_stream.csv | cut -c1-140
