#!/bin/bash
set -x
O=gpurun_out/r3c; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullsize.py tests/test_gpu_harness.py tests/test_gpu_trainer.py -q --maxfail=20 -s -p no:cacheprovider > $O/tests.log 2>&1; grep "bf16 bound\|explained\|passed\|failed\|^FAILED\|^E  " $O/tests.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
timeout 600 python bench.py --steps 50 --warmup 10 > $O/bench.log 2>&1; tail -c 2500 $O/bench.log
