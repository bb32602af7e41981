import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def have_gpu():
    from galahad_amd._lib import lib
    return lib.gsls_device_count() > 0


@pytest.fixture(scope="session", autouse=True)
def _torch_sees_the_gpu_first():
    """Some GPU tests hand torch tensors to the *_dev entry points.  torch must create its HIP context before
    libgsls.so has touched the device: initialised afterwards, torch reports "No HIP GPUs are available" on this image
    (seen when such a test ran first in a session).  A no-op on the CPU box."""
    if os.path.exists("/dev/kfd"):
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:      # the tests that need torch will say so themselves
            pass
    yield
