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
    from oracle import oracle as o
    o.lib()
    return o


@pytest.fixture(scope="session")
def nl():
    """The product package; GPU tests fail loudly if the HIP library or a device is missing."""
    import nightlight_amd
    from nightlight_amd import capi
    capi.load()
    if capi.device_count() < 1:
        pytest.fail("no HIP device visible: GPU tests need a real MI355X (there is no CPU fallback)")
    return nightlight_amd
