#!/bin/bash
set -x
O=gpurun_out/r3g; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_e2e.py tests/test_gpu_fullsize.py tests/test_gpu_trainer.py -q --maxfail=20 -s -p no:cacheprovider > $O/tests.log 2>&1; grep "explained\|passed\|failed\|^FAILED\|^E  " $O/tests.log | cut -c1-300
timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $O/bench.log 2>&1; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r3g/bench.log') if l.startswith('{')][-1])
r=d['roofline']
print('ms_per_step', d['ms_per_step'], 'value', d['value'])
print({k: r[k] for k in ('kernel','avg_ms','achieved','frac','traffic')})
print('step', r['step'], 'fwd', {k: r['forward_vgg19_fpn'][k] for k in ('ms','tflops','mfma_frac')})
for e in r['conv_breakdown'][:12]: print(e)
print({k: d.get(k) for k in ('scenes_per_gpu_2','fp32_ms_per_step','eval_forward_protocol')})
PY
