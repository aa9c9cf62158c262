#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5n
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_graph.py -q -m gpu -k "forward_only or second_trainer" > $O/t.log 2>&1; tail -6 $O/t.log
for r in 1 2; do
  for g in off fwd; do
    timeout 300 python bench.py --steps 30 --warmup 10 --no-extras --no-cpu-baseline --graph $g > $O/bench_graph_${g}_r$r.json 2> $O/bench_graph_${g}_r$r.err
    python -c "import json; d=json.load(open('$O/bench_graph_${g}_r$r.json')); print('graph $g round $r', d['ms_per_step'], d['host']['enqueue_ms_per_step'], d['host']['c_abi_calls_per_step'])"
  done
done
