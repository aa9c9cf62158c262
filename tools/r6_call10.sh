#!/bin/bash
cd "$(dirname "$0")/.."
timeout 600 python -m pytest tests/test_gpu_conv.py -q -m gpu -p no:cacheprovider -x -k "maxpool or upsample" 2>&1 | tail -2
for sl in 512 1024 2048 4096; do echo "--- slabs $sl"; NRPN_BN_SLABS=$sl python tools/r6_hbm_stages.py 2>&1 | grep -E "bn_|upsample"; done
