timeout 900 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_harness.py::test_bench_distributed_path_on_one_rank -x -q 2>&1 | tail -25
