import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oracle

    oracle.build()
    return oracle.lib()


@pytest.fixture(scope="session", autouse=True)
def _torch_hip_runtime_first():
    """PyTorch bundles its own HIP runtime; it must initialise before libsbr_hip.so brings in the system one
    (the other order leaves torch with "No HIP GPUs are available").  No-op on a machine without a GPU."""
    try:
        import torch

        if torch.cuda.is_available():
            torch.zeros(1, device="cuda")
    except Exception:
        pass
    yield
