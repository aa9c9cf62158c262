#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5k
mkdir -p $O
root=$PWD
export TMPDIR=/tmp
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/prof4 -o p --output-format csv -- python $root/bench.py --model swin_fcos --steps 6 --warmup 5 --no-cpu-baseline --no-probe --no-extras > $root/$O/prof.log 2>&1)
cp $(find /tmp/prof4 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_swin_fcos.csv
tail -2 $O/prof.log | cut -c1-200
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r5k/kernel_stats_swin_fcos.csv')))
steps=17
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms per step', tot/1e6/steps)
for r in rows[:28]: print(f"{float(r['TotalDurationNs'])/1e6/steps:7.3f} ms/step  {int(r['Calls'])/steps:6.1f} calls  {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:120]}")
PY
