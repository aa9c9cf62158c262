#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5j
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_fcos.py tests/test_gpu_fullsize.py tests/test_gpu_harness.py -q -m gpu -k "fcos" > $O/t_fcos.log 2>&1; tail -12 $O/t_fcos.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
python -c "import json; d=json.load(open('$O/bench.json')); print({k: d.get(k) for k in ('ms_per_step','bf16x3_ms_per_step')}); s=d['secondary']; print({k: (v.get('ms_per_step'), v.get('host_enqueue_ms_per_step'), v.get('c_abi_calls_per_step'), v.get('error')) for k, v in s.items()})"
timeout 300 python bench.py --model vgg_fcos --steps 15 --warmup 6 --no-extras --no-cpu-baseline --graph on > $O/bench_vgg_fcos_graph.json 2> $O/bench_vgg_fcos_graph.err; python -c "import json; d=json.load(open('$O/bench_vgg_fcos_graph.json')); print('vgg_fcos graph', d['ms_per_step'], d['host'])"
timeout 300 python bench.py --model vgg_fcos --steps 15 --warmup 6 --no-extras --no-cpu-baseline --graph off > $O/bench_vgg_fcos_eager.json 2> $O/bench_vgg_fcos_eager.err; python -c "import json; d=json.load(open('$O/bench_vgg_fcos_eager.json')); print('vgg_fcos eager', d['ms_per_step'], d['host'])"
