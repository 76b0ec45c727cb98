import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # a wedged GPU box must fail a test, not hang the suite: every GPU test gets a generous time limit where pytest-timeout is installed
    # (the longest one, the 200-pass soak, takes about a minute)
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for it in items:
        if it.get_closest_marker("gpu") is not None and it.get_closest_marker("timeout") is None:
            it.add_marker(pytest.mark.timeout(1200))


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
