#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6c12
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fcos.py tests/test_gpu_harness.py -q -m gpu -p no:cacheprovider -x > $O/t.log 2>&1; tail -3 $O/t.log
for m in swin_fcos vgg_fcos; do
  timeout 200 python bench.py --model $m --graph auto --steps 20 --no-cpu-baseline --no-extras --no-probe > $O/bench_$m.json 2>$O/bench_$m.err
  python -c "import json; d=json.load(open('$O/bench_$m.json')); print('$m', d['ms_per_step'], d.get('host'))" | cut -c1-200
done
