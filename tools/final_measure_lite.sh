#!/bin/bash
# Short form of tools/final_measure.sh for the end of a GPU budget: the default bench line, the PMC passes the bench's `traffic` field is tied to,
# then the two rocprofv3 kernel-stats runs and the smoke call -- most important first, every step under its own timeout.
set -u
out=gpurun_out/final
mkdir -p $out
root=$PWD
export TMPDIR=/tmp
timeout 150 python bench.py > $out/bench.log 2>$out/bench.err; grep '^{' $out/bench.log > $out/bench_n1.json; python tools/bench_line.py final < $out/bench_n1.json | cut -c1-200
timeout 200 bash tools/pmc_conv.sh $out/pmc > $out/pmc.log 2>&1; tail -3 $out/pmc.log | cut -c1-200
(cd /tmp && timeout 60 rocprofv3 --kernel-trace --stats -d /tmp/prof2 -o p --output-format csv -- python $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-probe --no-extras > $root/$out/prof_bench2.log 2>&1)
cp $(find /tmp/prof2 -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
(cd /tmp && NRPN_WGRAD_STREAM=0 timeout 60 rocprofv3 --kernel-trace --stats -d /tmp/prof -o p --output-format csv -- python $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-probe --no-extras > $root/$out/prof_bench.log 2>&1)
cp $(find /tmp/prof -name "*kernel_stats.csv" | head -1) $out/kernel_stats_single_stream.csv
python tools/prof_summary.py $(find /tmp/prof -name "*kernel_trace.csv" | head -1) $out/kernel_summary_single_stream.json 7
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
