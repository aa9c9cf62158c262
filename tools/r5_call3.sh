#!/bin/bash
# GPU call 3 of round 5: ROIPool kernels, the two tests adjusted after call 2, a kernel profile of the bf16x3 step
cd "$(dirname "$0")/.."
O=gpurun_out/r5c
mkdir -p $O
root=$PWD
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_detector.py -q -m gpu > $O/t_det.log 2>&1; tail -15 $O/t_det.log
timeout 600 python -m pytest tests/test_gpu_trainer.py tests/test_gpu_e2e.py -q -m gpu -k "trains_like or (bf16x3 and resnet)" > $O/t_b.log 2>&1; tail -8 $O/t_b.log
(cd /tmp && NRPN_BF16X3=1 timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/prof3 -o p --output-format csv -- python $root/bench.py --dtype f32 --steps 4 --warmup 2 --no-cpu-baseline --no-probe --no-extras > $root/$O/prof_x3.log 2>&1)
cp $(find /tmp/prof3 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_bf16x3.csv
tail -2 $O/prof_x3.log | cut -c1-300
head -25 $O/kernel_stats_bf16x3.csv | cut -c1-160
