set -u
mkdir -p gpurun_out/r3t
timeout 900 python -m pytest tests/test_gpu_trainer.py -x -q 2>&1 | tail -5
for ov in 0 1 0 1; do
NRPN_ADAMW_OVERLAP=$ov timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-probe --no-extras > gpurun_out/r3t/b$ov.log 2>&1
grep '^{' gpurun_out/r3t/b$ov.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('overlap $ov', d['ms_per_step'], d['value'], d['final_loss'])"
done
