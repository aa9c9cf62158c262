#!/bin/bash
# GPU call 1 of round 5: new-kernel tests, kernel A/Bs, bench A/B (halo pairing), full bench line, then the whole GPU suite with the parity log.
cd "$(dirname "$0")/.."
O=gpurun_out/r5a
mkdir -p $O
export NRPN_PARITY_LOG=$PWD/$O/parity_measured.json
timeout 600 python -m pytest tests/test_gpu_conv.py -q -m gpu -k "halo or bf16x3 or split or maxpool or upsample or batchnorm" > $O/t_new.log 2>&1
tail -5 $O/t_new.log
timeout 300 python tools/bench_r5.py $O/bench_r5.json > $O/bench_r5.log 2>&1
tail -12 $O/bench_r5.log
for r in 1 2; do
  for p in 2 1; do
    NRPN_HALO_PAIRING=$p timeout 300 python bench.py --steps 30 --warmup 10 --no-extras --no-cpu-baseline > $O/bench_pair${p}_r$r.json 2> $O/bench_pair${p}_r$r.err
    python -c "import json,sys; d=json.load(open('$O/bench_pair${p}_r$r.json')); print('pairing $p round $r', d['ms_per_step'], d['roofline']['avg_ms'], d['roofline']['forward_vgg19_fpn']['ms'])"
  done
done
timeout 900 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > $O/bench_full.json 2> $O/bench_full.err
tail -c 600 $O/bench_full.err
python -c "import json; d=json.load(open('$O/bench_full.json')); print({k: d.get(k) for k in ('ms_per_step','fp32_ms_per_step','bf16x3_ms_per_step','dense_head_ms_per_step','secondary','host')})"
timeout 1500 python -m pytest tests -q -m gpu --durations=25 -p no:cacheprovider > $O/t_all.log 2>&1
tail -40 $O/t_all.log
