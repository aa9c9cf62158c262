#!/bin/bash
# NEEDS tools/patches/r6_fused_splitk.patch applied (git apply) and the library rebuilt (the fused split-K experiment is not merged).
cd "$(dirname "$0")/.."
for d in 0 1 2 4 6; do NRPN_FUSED_DBG=$d timeout 120 python tools/r6_probe_fused.py 2>&1 | grep dbg; done
