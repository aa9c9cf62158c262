#!/bin/bash
# Final measurement of round 5: the bench line, PMC passes (the bench's `traffic` field is tied to them), kernel-stats runs (bf16 step in its measured
# configuration + single stream, bf16x3 step), smoke, then the whole GPU suite.  Most important first, every step under its own timeout.
cd "$(dirname "$0")/.."
O=${OUT:-gpurun_out/r5f}
mkdir -p $O
root=$PWD
export TMPDIR=/tmp
if [ "${SKIP_PMC:-0}" != "1" ]; then timeout 300 bash tools/pmc_conv.sh $O/pmc > $O/pmc.log 2>&1; tail -2 $O/pmc.log | cut -c1-200; cp $O/pmc/pmc_summary.json profiles/r05_pmc_conv_256x256_40c.json; fi

timeout 500 python bench.py > $O/bench.log 2> $O/bench.err; grep '^{' $O/bench.log > $O/bench_n1.json; python tools/bench_line.py r5f < $O/bench_n1.json | cut -c1-300
python -c "import json; d=json.load(open('$O/bench_n1.json')); print({k: d.get(k) for k in ('ms_per_step','value','fp32_ms_per_step','bf16x3_ms_per_step','dense_head_ms_per_step')}, d['roofline']['traffic'], d['roofline']['frac'], {k: v.get('ms_per_step') for k, v in d['secondary'].items()})"
(cd /tmp && timeout 90 rocprofv3 --kernel-trace --stats -d /tmp/prof2 -o p --output-format csv -- python $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-probe --no-extras > $root/$O/prof_bench2.log 2>&1)
cp $(find /tmp/prof2 -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
(cd /tmp && NRPN_WGRAD_STREAM=0 timeout 90 rocprofv3 --kernel-trace --stats -d /tmp/prof -o p --output-format csv -- python $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-probe --no-extras > $root/$O/prof_bench.log 2>&1)
cp $(find /tmp/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats_single_stream.csv
python tools/prof_summary.py $(find /tmp/prof -name "*kernel_trace.csv" | head -1) $O/kernel_summary_single_stream.json 13
if [ "${SKIP_X3:-0}" != "1" ]; then
(cd /tmp && NRPN_BF16X3=1 timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/prof3 -o p --output-format csv -- python $root/bench.py --dtype f32 --steps 4 --warmup 2 --no-cpu-baseline --no-probe --no-extras > $root/$O/prof_x3.log 2>&1)
cp $(find /tmp/prof3 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_bf16x3.csv
fi
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
NRPN_PARITY_LOG=$PWD/$O/parity_measured.json timeout 1500 python -m pytest tests -q -m gpu --durations=10 -p no:cacheprovider > $O/t_all.log 2>&1
tail -25 $O/t_all.log
