import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracledrv
    oracledrv.lib()
    return oracledrv


@pytest.fixture(scope="session", autouse=True)
def _torch_initialises_hip_first():
    """PyTorch ships its own HIP runtime; tests that also need torch streams must let torch initialise the
    device before libkalign_amd.so (linked against /opt/rocm) does, the order bench.py uses.  No-op on CPU."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
            torch.cuda.current_stream()
    except Exception:
        pass
    yield
