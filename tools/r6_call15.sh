#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6c15
mkdir -p $O
for rep in 1 2; do for f in 1 0; do
  NRPN_HALO_AUTO=$f timeout 200 python bench.py --steps 30 --no-cpu-baseline --no-extras > $O/h_$f.json 2>$O/h_$f.err
  python -c "import json; d=json.load(open('$O/h_$f.json')); r=d['roofline']; print('vgg_rpn halo_auto=$f', d['ms_per_step'], r['kernel'], r['avg_ms'], r['forward_vgg19_fpn']['ms'])"
done; done
