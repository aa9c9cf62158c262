timeout 900 python -m pytest "tests/test_gpu_e2e.py::test_train_matches_reference" -x -q -k "160" 2>&1 | tail -4
