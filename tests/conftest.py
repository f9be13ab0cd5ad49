import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_present() -> bool:
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    # On a box without any AMD GPU device node the gpu tests cannot run at all; skip them there.
    # On a GPU box nothing is skipped: a missing libdfx.so or a failing dfx_create is a test failure.
    if _gpu_present():
        return
    skip = pytest.mark.skip(reason="no /dev/kfd: not a GPU box")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    import numpy as np

    from oracle import oracle_py

    oracle_py.build()
    # Create the OpenMP worker pool now, before any test initialises the HIP runtime in this process
    # (observed on the MI355X box: the first parallel region after HIP start-up can stall for minutes).
    z = np.zeros((160, 160), np.uint8)
    oracle_py.tvl1_calc(z, z, threads=os.cpu_count() or 1)
    return oracle_py


@pytest.fixture(scope="session")
def dfx():
    import denseflow_amd

    denseflow_amd.load_library()
    return denseflow_amd
