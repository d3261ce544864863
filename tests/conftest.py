import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def root():
    return ROOT


def scene_path(*parts):
    return os.path.join(ROOT, "scenes", *parts)


@pytest.fixture(scope="session", autouse=True)
def _torch_device_first():
    """Where a GPU is present, bring torch's HIP context up before the first context of libptcore.so:
    the order bench.py uses.  (Without a GPU this is a no-op.)"""
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
    yield
