import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture(scope="session", autouse=True)
def _torch_hip_runtime_first(request):
    """On a GPU box some tests hand torch device tensors to the C ABI.  torch ships its own HIP
    runtime; it has to initialise before the library's (system) runtime has created and destroyed
    contexts in the same process, otherwise torch later reports "No HIP GPUs are available".
    bench.py has the same order (torch first)."""
    if "not gpu" in (request.config.getoption("-m") or ""):
        return
    try:
        import torch
        if torch.cuda.is_available():
            torch.zeros(1, device="cuda")
    except Exception:
        pass
