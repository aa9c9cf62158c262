#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5g
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullsize.py tests/test_gpu_conv.py -q -m gpu -k "bf16x3" > $O/t_x3.log 2>&1; tail -4 $O/t_x3.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.err
python -c "import json; d=json.load(open('$O/bench.json')); print({k: d.get(k) for k in ('ms_per_step','fp32_ms_per_step','bf16x3_ms_per_step','dense_head_ms_per_step')}); print(json.dumps(d['secondary'], indent=0)[:2500])"
