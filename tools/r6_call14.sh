#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6c14
mkdir -p $O
for rep in 1 2; do for g in off fwd; do
  timeout 200 python bench.py --graph $g --steps 40 --no-cpu-baseline --no-extras --no-probe > $O/g_$g.json 2>$O/g_$g.err
  python -c "import json; d=json.load(open('$O/g_$g.json')); print('vgg_rpn graph=$g', d['ms_per_step'], d['host']['enqueue_ms_per_step'], d['host']['c_abi_calls_per_step'])"
done; done
