#!/bin/bash
set -x
O=gpurun_out/r3d; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/diag_unmatched.py eval_resnet_obb eval_swin_obb eval_swin_obb_200x200x130 > $O/diag.log 2>&1; cat $O/diag.log | cut -c1-400
timeout 300 python -m pytest tests/test_gpu_trainer.py -q -p no:cacheprovider > $O/trainer.log 2>&1; tail -3 $O/trainer.log
