import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "tiler_golden.npz"), allow_pickle=False)


@pytest.fixture(scope="session", autouse=True)
def _cu_budget_from_env():
    """RSU_TEST_CU_BUDGET=n reruns the GPU suite with the persistent kernels planned for n CUs (the backward pass plans for half
    of the chip per stream: every op must be right at any budget)"""
    n = os.environ.get("RSU_TEST_CU_BUDGET")
    if n:
        from road_segmentation_unet_amd._lib import call
        call("rsu_set_cu_budget", int(n))
    yield
