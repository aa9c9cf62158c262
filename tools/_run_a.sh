mkdir -p gpurun_out/r3q
for spg in 2 4; do
  timeout 300 python bench.py --scenes-per-gpu $spg --steps 12 --warmup 3 --no-cpu-baseline --no-probe --no-extras > gpurun_out/r3q/spg$spg.log 2>&1
  grep '^{' gpurun_out/r3q/spg$spg.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('scenes/gpu $spg', d['ms_per_step'], d['value'])"
done
timeout 200 python tools/eval_protocol.py 20 160 160 160 2>&1 | tail -4
