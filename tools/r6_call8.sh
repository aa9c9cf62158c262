#!/bin/bash
# round 6, call 8: in-kernel accumulation of LayerNorm / GroupNorm / attention-table gradients -- tests + Swin-S workloads
cd "$(dirname "$0")/.."
O=gpurun_out/r6c8
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_swin.py tests/test_gpu_fcos.py tests/test_gpu_trainer.py -q -m gpu -p no:cacheprovider -x > $O/t1.log 2>&1; tail -3 $O/t1.log
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -m gpu -p no:cacheprovider -x -k "swin or fcos" > $O/t2.log 2>&1; tail -3 $O/t2.log
for m in swin_fcos swin_rpn; do
  timeout 200 python bench.py --model $m --graph auto --steps 20 --no-cpu-baseline --no-extras --no-probe > $O/bench_$m.json 2>$O/bench_$m.err
  python -c "import json; d=json.load(open('$O/bench_$m.json')); print('$m', d['ms_per_step'], d.get('host'))" | cut -c1-250
done
