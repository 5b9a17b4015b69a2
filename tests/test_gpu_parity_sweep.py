"""A bounded edition of tests/sweeps/parity_sweep.py as a driver-run GPU test: the default dispatch of the C ABI
on synthetic row tiles of the bench stack (4096 pixels wide: NaN rows at the top, NaN columns at the right, hot /
cold outliers, the all-NaN patch) against the CPU oracle on EVERY pixel -- one case per kernel family and
frame-count class, sized so that the oracle takes about a second per case.  Clip counters identical; values
bit-exact for the exact kernels, within the north star's 1e-5 for the register-resident ones."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [  # mode, frames, rows, row0, image rows, weighted
    (2, 128, 64, 0, 4096, False), (2, 128, 48, 2040, 4096, False),        # headline kernel: top rows / the all-NaN patch
    (3, 128, 32, 0, 4096, False), (2, 32, 128, 0, 4096, False), (3, 24, 96, 0, 4096, False),
    (2, 100, 48, 0, 4096, False), (2, 64, 64, 0, 4096, False),
    (2, 512, 16, 0, 4096, False), (2, 500, 16, 2040, 4096, False),        # selection front end (497..512 frames)
    (3, 512, 8, 0, 4096, False), (2, 300, 24, 0, 4096, False), (3, 200, 24, 0, 4096, False), (2, 256, 24, 0, 4096, False),
    (5, 128, 16, 0, 4096, False), (5, 256, 8, 0, 4096, False), (0, 64, 64, 0, 4096, False), (0, 512, 16, 0, 4096, False),
    (4, 128, 32, 0, 4096, False), (4, 100, 32, 0, 4096, False), (4, 256, 16, 0, 4096, False),
    (2, 128, 16, 0, 4096, True), (3, 96, 16, 0, 4096, True), (2, 48, 32, 0, 4096, True),
    (2, 600, 4, 0, 4096, False),
]


@pytest.mark.parametrize("mode,n,rows,row0,image_rows,weighted", CASES)
def test_default_dispatch_matches_the_oracle_on_every_pixel(nl, oracle, mode, n, rows, row0, image_rows, weighted):
    w = 4096
    with nl.StackHandle(n, w, image_rows, row0=row0, rows=rows) as st:
        st.fill_synthetic(17 + n)
        frames = np.stack([st.download_tile(i) for i in range(n)])
        weights = None
        if weighted:
            weights = np.random.default_rng(n).uniform(0.2, 1.0, n).astype(np.float32)
            st.set_weights(weights)
        got, cl, ch = st.run(mode, 3.0, 2.5)
        got = got[row0 * w:(row0 + rows) * w]
        kernel = st.last_kernel_name
    ow = None if mode in (0, 5) else weights
    rc, want, wl, wh, _ = oracle.stack_apply(mode, frames, ow, 3.0, 2.5, 0.0, num_cpu=os.cpu_count())
    assert rc == 0
    assert np.array_equal(np.isnan(got), np.isnan(want)), kernel
    if mode >= 2:
        assert (cl, ch) == (wl, wh), kernel
    ok = ~np.isnan(want) & (want != got)
    rel = float(np.max(np.abs(got[ok].astype(np.float64) - want[ok]) / np.abs(want[ok].astype(np.float64)))) if ok.any() else 0.0
    assert rel <= 1e-5, (kernel, rel)                         # the north star's tolerance, fp32
    if mode in (0, 1, 5) or weighted:
        assert rel == 0.0, (kernel, rel)                      # bit-exact kernels
