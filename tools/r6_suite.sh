#!/bin/bash
# whole GPU suite (parity log on) + smoke
cd "$(dirname "$0")/.."
O=gpurun_out/${OUT:-r6suite}
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
NRPN_PARITY_LOG=$PWD/$O/parity_measured.json timeout 2400 python -m pytest tests -q -m gpu --durations=8 -p no:cacheprovider > $O/t_all.log 2>&1
tail -16 $O/t_all.log
