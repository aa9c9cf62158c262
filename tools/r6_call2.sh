#!/bin/bash
# round 6, call 2: the new parity tests (north_star tolerance on raw head outputs, ragged bf16x3 gradients)
cd "$(dirname "$0")/.."
O=gpurun_out/r6c2
mkdir -p $O
NRPN_PARITY_LOG=$PWD/$O/parity_new.json timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -s -k "north_star or ragged_conv_under" > $O/t_new.log 2>&1
grep -E "parity\]|headout\]|passed|failed|Error|assert" $O/t_new.log | cut -c1-220 | tail -40
