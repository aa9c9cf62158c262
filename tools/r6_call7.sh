#!/bin/bash
# round 6, call 7: two-stream timeline of a training step
cd "$(dirname "$0")/.."
O=gpurun_out/r6c7
mkdir -p $O
root=$PWD
export TMPDIR=/tmp
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/st -o p --output-format csv -- python $root/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-probe --no-extras > $root/$O/prof.log 2>&1)
python tools/step_timeline.py $(find /tmp/st -name "*kernel_trace.csv" | head -1) > $O/timeline.txt; head -1 $O/timeline.txt
python tools/trace_gaps.py $(find /tmp/st -name "*kernel_trace.csv" | head -1) 4 $O/gaps.json | head -3
grep '^{' $O/prof.log | python tools/bench_line.py profiled | cut -c1-100
