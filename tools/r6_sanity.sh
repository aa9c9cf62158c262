#!/bin/bash
# last sanity pass on the committed tree: smoke, a short bench line, the kernel / harness parity files that touch this round's changes
cd "$(dirname "$0")/.."
O=gpurun_out/r6sanity
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 300 python bench.py --steps 30 --no-cpu-baseline --no-extras > $O/bench.json 2>$O/bench.err; python tools/bench_line.py sanity < $O/bench.json | cut -c1-200
timeout 1200 python -m pytest tests/test_gpu_conv.py tests/test_gpu_fcos.py tests/test_gpu_swin.py tests/test_gpu_stages.py -q -m gpu -p no:cacheprovider > $O/t.log 2>&1; tail -2 $O/t.log
