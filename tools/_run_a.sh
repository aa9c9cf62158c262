set -u
out=gpurun_out/r3j
mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_postproc.py tests/test_gpu_stages.py -x -q 2>&1 | tail -15
timeout 300 python tools/eval_protocol.py 20 > $out/eval.log 2>&1; tail -5 $out/eval.log
