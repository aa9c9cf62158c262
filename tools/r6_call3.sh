#!/bin/bash
# round 6, call 3: lean prologue of the implicit-GEMM loaders -- conv parity tests + forward-only kernel sequence
cd "$(dirname "$0")/.."
O=gpurun_out/${OUT:-r6c3}
mkdir -p $O
root=$PWD
export TMPDIR=/tmp
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/fwd -o p --output-format csv -- python $root/tools/forward_trace.py 20 > $root/$O/forward_prof.log 2>&1)
cp $(find /tmp/fwd -name "*kernel_stats.csv" | head -1) $O/forward_kernel_stats.csv
python tools/forward_trace.py --seq $(find /tmp/fwd -name "*kernel_trace.csv" | head -1) 20 > $O/forward_seq.txt; tail -1 $O/forward_seq.txt
python tools/forward_trace.py 50 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_conv.py -q -m gpu -p no:cacheprovider -x > $O/t_conv.log 2>&1; tail -3 $O/t_conv.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench.log 2> $O/bench.err; grep '^{' $O/bench.log > $O/bench_n1.json
python tools/bench_line.py $O < $O/bench_n1.json | cut -c1-300
