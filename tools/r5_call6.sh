#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5h
mkdir -p $O
for r in 1 2; do
  for g in off on; do
    timeout 300 python bench.py --steps 30 --warmup 10 --no-extras --no-cpu-baseline --graph $g > $O/bench_graph_${g}_r$r.json 2> $O/bench_graph_${g}_r$r.err
    python -c "import json; d=json.load(open('$O/bench_graph_${g}_r$r.json')); print('graph $g round $r', d['ms_per_step'], d['host']['enqueue_ms_per_step'], d['roofline']['avg_ms'])"
  done
done
for g in off on; do
  timeout 300 python bench.py --steps 20 --warmup 8 --no-extras --no-cpu-baseline --graph $g --scenes-per-gpu 2 > $O/bench_spg2_graph_${g}.json 2> $O/bench_spg2_graph_${g}.err
  python -c "import json; d=json.load(open('$O/bench_spg2_graph_${g}.json')); print('spg2 graph $g', d['ms_per_step'], d['value'])"
done
