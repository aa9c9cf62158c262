#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6c13
mkdir -p $O
for rep in 1 2; do for f in 0 1; do
  NRPN_GN_FAST=$f timeout 200 python bench.py --model swin_fcos --graph auto --steps 20 --no-cpu-baseline --no-extras --no-probe > $O/b_$f.json 2>$O/b_$f.err
  python -c "import json; d=json.load(open('$O/b_$f.json')); print('swin_fcos gn_fast=$f', d['ms_per_step'])"
done; done
for rep in 1 2; do for f in 0 1; do
  NRPN_POOL_FAST=$f timeout 200 python bench.py --steps 30 --no-cpu-baseline --no-extras --no-probe > $O/p_$f.json 2>$O/p_$f.err
  python -c "import json; d=json.load(open('$O/p_$f.json')); print('vgg_rpn pool_fast=$f', d['ms_per_step'])"
done; done
