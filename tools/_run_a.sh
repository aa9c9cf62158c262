set -u
mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > gpurun_out/final/pytest_gpu.log; cat gpurun_out/final/pytest_gpu.log
bash tools/final_measure.sh 2>&1 | tail -12
