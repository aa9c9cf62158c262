#!/bin/bash
# Round-end measurement on the GPU box: smoke, the default bench line (with the CPU baseline), and a rocprofv3 kernel trace + stats of the
# same workload (single stream: clean per-kernel durations).  Outputs under gpurun_out/final/.
set -u
out=gpurun_out/final
mkdir -p $out
root=$PWD
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
python bench.py > $out/bench.log 2>$out/bench.err; grep '^{' $out/bench.log > $out/bench_n1.json; python tools/bench_line.py final < $out/bench_n1.json | cut -c1-200
export TMPDIR=/tmp
(cd /tmp && NRPN_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -d /tmp/prof -o p --output-format csv -- python $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-probe --no-extras > $root/$out/prof_bench.log 2>&1)
cp $(find /tmp/prof -name "*kernel_stats.csv" | head -1) $out/kernel_stats_single_stream.csv
python tools/prof_summary.py $(find /tmp/prof -name "*kernel_trace.csv" | head -1) $out/kernel_summary_single_stream.json 7
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof2 -o p --output-format csv -- python $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-probe --no-extras > $root/$out/prof_bench2.log 2>&1)
cp $(find /tmp/prof2 -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
grep '^{' $out/prof_bench.log | python tools/bench_line.py profiled-single | cut -c1-120
grep '^{' $out/prof_bench2.log | python tools/bench_line.py profiled-streams | cut -c1-120
# secondary workloads (not the BASELINE metric): eager and with the trunk as captured HIP graphs
for m in resnet_rpn swin_rpn swin_fcos vgg_fcos; do
  for g in off on; do
    timeout 200 python bench.py --model $m --graph $g --steps 30 --no-cpu-baseline --no-extras --no-probe > $out/bench_${m}_graph_$g.json 2>/dev/null
    python -c "import json; d=json.load(open('$out/bench_${m}_graph_$g.json')); print('$m graph=$g', d['ms_per_step'], d['value'], d.get('host'))" | cut -c1-200
  done
done
timeout 200 python bench.py --graph on --steps 30 --no-cpu-baseline --no-extras --no-probe > $out/bench_vgg_rpn_graph_on.json 2>/dev/null
NRPN_CONE=0 timeout 200 python bench.py --steps 30 --no-cpu-baseline --no-extras > $out/bench_vgg_rpn_dense_head.json 2>/dev/null
python -c "import json; print('vgg graph=on', json.load(open('$out/bench_vgg_rpn_graph_on.json'))['ms_per_step'], 'dense head', json.load(open('$out/bench_vgg_rpn_dense_head.json'))['ms_per_step'])"
# PMC counters of the kernels that can serve the dominant shape (halo form, 256x256 tile on 8 / 4 waves) and the 256x256 wgrad kernel
bash tools/pmc_conv.sh $out/pmc > $out/pmc.log 2>&1; tail -5 $out/pmc.log | cut -c1-200
