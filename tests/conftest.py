import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_api
    return oracle_api.load()


@pytest.fixture(scope="session")
def evplp():
    import torch  # noqa: F401  (first, so libevplp_hip.so binds to the HIP runtime torch loaded)
    import evplp_amd
    evplp_amd.lib()
    return evplp_amd
