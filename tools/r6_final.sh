#!/bin/bash
# Final measurement of round 6: PMC passes (the bench's `traffic` field is tied to them), the MFMA ceiling probe, the bench line, kernel-stats
# runs (measured configuration + single stream), the FORWARD-ONLY kernel table, Swin-S+FCOS launch table, power / clock log,
# smoke, then the whole GPU suite.  Most important first, every step under its own timeout.
cd "$(dirname "$0")/.."
O=${OUT:-gpurun_out/r6f}
mkdir -p $O
root=$PWD
export TMPDIR=/tmp
if [ "${SKIP_PMC:-0}" != "1" ]; then timeout 400 bash tools/pmc_conv.sh $O/pmc > $O/pmc.log 2>&1; tail -2 $O/pmc.log | cut -c1-200; cp $O/pmc/pmc_summary.json profiles/r06_pmc_conv_256x256_40c.json; fi
if [ "${SKIP_CEIL:-0}" != "1" ]; then timeout 200 python tools/mfma_peak_probe.py 3 > $O/ceiling.log 2>&1; tail -3 $O/ceiling.log | cut -c1-200; cp gpurun_out/r04_mfma_ceiling.json $O/mfma_ceiling.json 2>/dev/null; cp $O/mfma_ceiling.json profiles/r06_mfma_ceiling.json 2>/dev/null; fi

timeout 600 python bench.py > $O/bench.log 2> $O/bench.err; grep '^{' $O/bench.log > $O/bench_n1.json; python tools/bench_line.py r6f < $O/bench_n1.json | cut -c1-300
python -c "import json; d=json.load(open('$O/bench_n1.json')); print({k: d.get(k) for k in ('ms_per_step','value','fp32_ms_per_step','bf16x3_ms_per_step','dense_head_ms_per_step')}, d['roofline']['traffic'], d['roofline']['frac'], {k: v.get('ms_per_step') for k, v in d['secondary'].items()}, len(open('$O/bench_n1.json').read()))"
# forward-only pass: kernel table + launch sequence
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/fwd -o p --output-format csv -- python $root/tools/forward_trace.py 20 > $root/$O/forward_prof.log 2>&1)
cp $(find /tmp/fwd -name "*kernel_stats.csv" | head -1) $O/forward_kernel_stats.csv
python tools/forward_trace.py --seq $(find /tmp/fwd -name "*kernel_trace.csv" | head -1) 20 > $O/forward_seq.txt; tail -1 $O/forward_seq.txt
python tools/forward_trace.py 50 2>&1 | tail -1
# training step: measured configuration (two streams) and single stream
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/prof2 -o p --output-format csv -- python $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-probe --no-extras > $root/$O/prof_bench2.log 2>&1)
cp $(find /tmp/prof2 -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
(cd /tmp && NRPN_WGRAD_STREAM=0 timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/prof -o p --output-format csv -- python $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-probe --no-extras > $root/$O/prof_bench.log 2>&1)
cp $(find /tmp/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats_single_stream.csv
python tools/prof_summary.py $(find /tmp/prof -name "*kernel_trace.csv" | head -1) $O/kernel_summary_single_stream.json 13
# Swin-S + FCOS (configs[3] workload): launch table
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/sf -o p --output-format csv -- python $root/bench.py --model swin_fcos --graph auto --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-probe > $root/$O/prof_sf.log 2>&1)
cp $(find /tmp/sf -name "*kernel_stats.csv" | head -1) $O/swin_fcos_kernel_stats.csv
python - <<PY
import csv
rows = list(csv.DictReader(open("$O/swin_fcos_kernel_stats.csv")))
calls = sum(int(r["Calls"]) for r in rows); native = sum(int(r["Calls"]) for r in rows if "at::native" in r["Name"] or "rocclr" in r["Name"] or "rocprim" in r["Name"])
print("swin_fcos launches in trace", calls, "torch-native", native, "steps 19 (3 warm-up + 10 timed + 6 host)", round(calls / 19), "per step")
PY
if [ "${SKIP_POWER:-0}" != "1" ]; then timeout 200 python tools/power_probe.py > $O/power_clock.log 2>&1; tail -6 $O/power_clock.log | cut -c1-160; fi
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
NRPN_PARITY_LOG=$PWD/$O/parity_measured.json timeout 2400 python -m pytest tests -q -m gpu --durations=10 -p no:cacheprovider > $O/t_all.log 2>&1
tail -25 $O/t_all.log
