#!/bin/bash
# round 6, call 6: secondary workloads on the tree with the lean prologue (base .so = same tree: sanity), launch counts
cd "$(dirname "$0")/.."
O=gpurun_out/r6c6
mkdir -p $O
for m in swin_fcos swin_rpn resnet_rpn; do
  timeout 200 python bench.py --model $m --graph auto --steps 20 --no-cpu-baseline --no-extras --no-probe > $O/bench_$m.json 2>$O/bench_$m.err
  python -c "import json; d=json.load(open('$O/bench_$m.json')); print('$m', d['ms_per_step'], d.get('host'))" | cut -c1-250
done
export TMPDIR=/tmp
root=$PWD
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/sf -o p --output-format csv -- python $root/bench.py --model swin_fcos --graph auto --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-probe > $root/$O/prof_sf.log 2>&1)
cp $(find /tmp/sf -name "*kernel_stats.csv" | head -1) $O/swin_fcos_kernel_stats.csv
head -30 $O/swin_fcos_kernel_stats.csv | cut -c1-160
