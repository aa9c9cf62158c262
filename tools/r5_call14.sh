#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5p
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_conv.py -q -m gpu -k "two_taps or conv_forward_backward or wgrad or bf16x3" > $O/t.log 2>&1; tail -8 $O/t.log
for r in 1 2; do
  for on in 0 1; do
    NRPN_WGRAD_PACK2=$on timeout 300 python bench.py --steps 30 --warmup 10 --no-extras --no-cpu-baseline > $O/bench_pack${on}_r$r.json 2> $O/bench_pack${on}_r$r.err
    python -c "import json; d=json.load(open('$O/bench_pack${on}_r$r.json')); print('pack2=$on round $r', d['ms_per_step'], [(c['cin'], c['cout'], c['avg_us']) for c in d['roofline']['conv_breakdown'] if c['call']=='conv3d_wgrad' and c['cin']<=128 and c['voxels']==64000])"
  done
done
