import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Kernel-level oracle / torch parity first, composed paths after: with `-x` one red end-to-end test must not hide the kernel tests.
_ORDER = ["test_oracle_golden", "test_abi_and_host", "test_harness_cpu", "test_dist_gloo", "test_gpu_postproc", "test_gpu_conv",
          "test_gpu_geomloss", "test_gpu_sampler", "test_gpu_swin", "test_gpu_fcos", "test_gpu_aug", "test_gpu_roialign", "test_gpu_detector", "test_gpu_e2e", "test_gpu_trainer",
          "test_gpu_fullsize", "test_gpu_harness", "test_gpu_rccl"]


def pytest_collection_modifyitems(config, items):
    def key(item):
        mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return _ORDER.index(mod) if mod in _ORDER else len(_ORDER)
    items.sort(key=key)        # stable: the order inside a file is kept


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
    return load


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from nerf_rpn_amd import lib
    lib.call("check_device", 0)   # fail loudly on a non-gfx950 part or a missing library
    return torch.device("cuda:0")
