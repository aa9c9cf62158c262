timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "bf16_training" 2>&1 | tail -4
