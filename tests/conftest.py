import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


GPU_TESTS_STARTED = [0]      # gpu-marked tests started in this process (tests/test_00_rccl_gpu.py must be the first)


def pytest_runtest_setup(item):
    if item.get_closest_marker("gpu") is not None:
        GPU_TESTS_STARTED[0] += 1


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
        return cache[name]

    return load
