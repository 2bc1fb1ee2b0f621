import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    """The HIP library must load and see a device; no fallback, the test fails otherwise."""
    from fftvis_amd import _lib

    _lib.lib()
    if _lib.device_count() < 1:
        pytest.fail("libfftvis_hip loaded but no HIP device is visible")
    return 0
