import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")


@pytest.fixture
def options():
    """Set kernel-selection options of liblx2v_hip.so for one test; the previous values come back
    afterwards.  Usage: options(attn_kernel=16, attn_qb=5)."""
    from lightx2v_amd import hip

    saved = {}

    def set_(**kw):
        for name, value in kw.items():
            saved.setdefault(name, hip.get_option(name))
            hip.set_option(name, value)

    yield set_
    for name, value in saved.items():
        hip.set_option(name, value)
