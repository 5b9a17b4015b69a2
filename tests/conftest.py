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


@pytest.fixture(scope="session")
def experiments(nl):
    """Skips tests of the kernels that only the experiments build carries (make -C nightlight_amd/csrc EXPERIMENTS=1:
    the four-pixels-per-wave replay, the split / persistent LDS-column pass, chunked passes, the guarded linear fit)."""
    import ctypes
    from nightlight_amd import capi
    lib = ctypes.CDLL(capi.LIB_PATH)
    lib.nl_version.restype = ctypes.c_char_p
    if b"+experiments" not in lib.nl_version():
        pytest.skip("needs the experiments build of libnlstack.so")
    return True
