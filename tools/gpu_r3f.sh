#!/bin/bash
set -x
O=gpurun_out/r3f; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_conv.py -q -p no:cacheprovider -k "halo" > $O/t_halo.log 2>&1; tail -5 $O/t_halo.log | cut -c1-300
SKIP_REST=1 timeout 300 python tools/bench_tile4.py > $O/tile4.log 2>&1; head -8 $O/tile4.log | cut -c1-500
timeout 600 python tools/diag_unmatched.py eval_resnet_obb_200x200x130 eval_swin_obb_160x120x64 eval_swin_obb_200x200x130 > $O/diag.log 2>&1; cat $O/diag.log | cut -c1-330
