#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5i
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_fcos.py -q -m gpu > $O/t_fcos.log 2>&1; tail -12 $O/t_fcos.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
python -c "import json; d=json.load(open('$O/bench.json')); print({k: d.get(k) for k in ('ms_per_step','bf16x3_ms_per_step')}); s=d['secondary']; print({k: (v.get('ms_per_step'), v.get('host_enqueue_ms_per_step'), v.get('c_abi_calls_per_step'), v.get('error')) for k, v in s.items()})"
