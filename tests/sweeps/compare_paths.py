#!/usr/bin/env python3
"""(Lives under tests/ because it uses the CPU oracle, which only tests, smoke() and the
cpu_baseline leg of bench.py may touch.)
Developer utility (GPU box): run one synthetic stack through the default
dispatch, the forced bit-exact kernels and (optionally, on a row strip) the
CPU oracle, and report where they differ.  Not part of the product path."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nightlight_amd import StackHandle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=128)
    ap.add_argument("--width", type=int, default=4096)
    ap.add_argument("--height", type=int, default=4096)
    ap.add_argument("--mode", type=int, default=2)
    ap.add_argument("--kappa", type=float, default=3.0)
    ap.add_argument("--oracle-rows", type=int, default=0)
    ap.add_argument("--oracle-row0", type=int, default=0)
    ap.add_argument("--seed", type=int, default=0x4E4C5354)
    a = ap.parse_args()
    n, w, h = a.frames, a.width, a.height
    with StackHandle(n, w, h) as st:
        st.fill_synthetic(a.seed)
        st.set_exact(False)
        fast, fl, fh = st.run(a.mode, a.kappa, a.kappa)
        t_fast, fb = st.last_kernel_ms, st.last_fallback_pixels
        st.set_exact(True)
        exact, el, eh = st.run(a.mode, a.kappa, a.kappa)
        t_exact = st.last_kernel_ms
        print("fast : clip", (fl, fh), "%.3f ms" % t_fast, "exact-list pixels", fb)
        print("exact: clip", (el, eh), "%.3f ms" % t_exact)
        ok = ~np.isnan(exact) & (exact != 0)
        rel = np.abs(fast[ok].astype(np.float64) - exact[ok]) / np.abs(exact[ok])
        print("fast vs exact: max rel %.3e, pixels > 1e-5: %d, nan pattern equal: %s"
              % (rel.max(), int((rel > 1e-5).sum()), np.array_equal(np.isnan(fast), np.isnan(exact))))
        bad = np.flatnonzero(ok)[rel > 1e-5]
        for i in bad[:10]:
            print("   pixel", i, divmod(int(i), w), "fast", fast[i], "exact", exact[i])
        if a.oracle_rows > 0:
            from oracle import oracle
            r0, rows = a.oracle_row0, a.oracle_rows
            frames = np.empty((n, rows * w), np.float32)
            for i in range(n):
                frames[i] = st.download_tile(i)[r0 * w:(r0 + rows) * w]
            t0 = time.perf_counter()
            rc, res, ol, oh, _ = oracle.stack_apply(a.mode, frames, None, a.kappa, a.kappa, 0.0,
                                                    num_cpu=os.cpu_count())
            print("oracle rows [%d,%d): clip %r in %.1f s" % (r0, r0 + rows, (ol, oh), time.perf_counter() - t0))
            ex = exact[r0 * w:(r0 + rows) * w]
            same = np.array_equal(ex, res, equal_nan=True)
            print("exact kernel vs oracle on the strip: bit-exact =", same)
            if not same:
                d = np.flatnonzero(~((ex == res) | (np.isnan(ex) & np.isnan(res))))
                print("   %d differing pixels, first:" % d.size, [(int(i), ex[i], res[i]) for i in d[:8]])
            with StackHandle(n, w, h, row0=r0, rows=rows) as s2:
                s2.fill_synthetic(a.seed)
                for exact_mode in (True, False):
                    s2.set_exact(exact_mode)
                    _, cl, ch = s2.run(a.mode, a.kappa, a.kappa)
                    print("   strip handle exact=%s: clip %r (oracle %r)" % (exact_mode, (cl, ch), (ol, oh)))


if __name__ == "__main__":
    main()
