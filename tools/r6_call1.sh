#!/bin/bash
# round 6, call 1: kernel table of the forward-only pass (VERDICT r5 #1 evidence) + a short bench line of the tree as it starts the round
cd "$(dirname "$0")/.."
O=gpurun_out/r6c1
mkdir -p $O
root=$PWD
export TMPDIR=/tmp
python tools/forward_trace.py 50 > $O/forward_plain.log 2>&1; tail -1 $O/forward_plain.log
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/fwd -o p --output-format csv -- python $root/tools/forward_trace.py 20 > $root/$O/forward_prof.log 2>&1)
cp $(find /tmp/fwd -name "*kernel_stats.csv" | head -1) $O/forward_kernel_stats.csv
python tools/forward_trace.py --seq $(find /tmp/fwd -name "*kernel_trace.csv" | head -1) 20 > $O/forward_seq.txt; tail -1 $O/forward_seq.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench.log 2> $O/bench.err; grep '^{' $O/bench.log > $O/bench_n1.json
python tools/bench_line.py r6c1 < $O/bench_n1.json | cut -c1-300
