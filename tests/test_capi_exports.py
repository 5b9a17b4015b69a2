"""CPU-side checks of the product library: it loads, exports every symbol
include/nlstack.h declares, and its host-only entry points behave like the
reference's getWeights (stack.go:231-270).  No compute call needs a GPU here;
the ones that would must fail loudly, never fall back to a CPU path."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "nlstack.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nl_[a-z0-9_]+)\s*\(", text)) - {"nl_reduce_fn"})


def test_library_exports_every_declared_symbol():
    from nightlight_amd import capi
    assert os.path.exists(capi.LIB_PATH), "libnlstack.so missing: run __graft_entry__.build()"
    lib = ctypes.CDLL(capi.LIB_PATH)
    declared = _declared()
    assert len(declared) >= 25
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, "declared in nlstack.h but not exported: %s" % missing
    assert sorted(capi.EXPORTS) == declared, "capi.EXPORTS out of sync with the header"


def test_version_and_error_string():
    from nightlight_amd import capi
    lib = capi.load()
    assert b"gfx950" in lib.nl_version()
    assert isinstance(capi.last_error(), str)


def test_weights_from_scalars_matches_oracle(oracle):
    import nightlight_amd as nl
    rng = np.random.default_rng(1)
    for n in (2, 5, 128):
        vals = (1.0 + rng.random(n)).astype(np.float32)
        for mode in (nl.WEIGHT_EXPOSURE, nl.WEIGHT_INVERSE_NOISE, nl.WEIGHT_INVERSE_HFR):
            rc, want, _ = oracle.get_weights(mode, vals)
            got = nl.weights_from_scalars(mode, vals)
            assert rc == 0 and np.array_equal(got, want)
    assert nl.weights_from_scalars(nl.WEIGHT_NONE, [1.0, 2.0]) is None


def test_weights_error_messages_follow_the_reference():
    import nightlight_amd as nl
    from nightlight_amd import capi
    with pytest.raises(capi.NlError) as e:
        nl.weights_from_scalars(nl.WEIGHT_EXPOSURE, [10.0, 0.0, 5.0])
    assert e.value.code == capi.ERR_MISSING_EXPOSURE
    assert e.value.message == "1: Missing exposure information for exposure-weighted stacking"
    with pytest.raises(capi.NlError) as e:
        nl.weights_from_scalars(7, [1.0])
    assert e.value.code == capi.ERR_INVALID_WEIGHTING
    assert e.value.message.startswith("Invalid weighting mode 7")


def test_no_cpu_fallback_without_a_device():
    """In the build container there is no GPU: creating a handle must fail
    loudly (NL_ERR_NO_DEVICE), not silently compute on the CPU."""
    import nightlight_amd as nl
    from nightlight_amd import capi
    if capi.device_count() > 0:
        pytest.skip("a HIP device is visible")
    with pytest.raises(capi.NlError) as e:
        nl.StackHandle(4, 8, 8)
    assert "no HIP device" in str(e.value) or "hipGetDeviceCount" in str(e.value)
    with pytest.raises(capi.NlError):
        nl.median_filter_3x3(np.zeros(64, np.float32), 8, 8)


def test_product_package_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under nightlight_amd/ may import it."""
    pkg = os.path.join(ROOT, "nightlight_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "nl_oracle" not in src and "import oracle" not in src and "from oracle" not in src, f
