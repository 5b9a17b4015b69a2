#!/usr/bin/env python3
"""Goal-seek (stackfindsigma.go:48-98) pass by pass: what each step of the bisection costs inside the sequence and what a
pass at the SAME pair of sigmas costs in steady state (third of three back-to-back passes on the same handle).

    python tools/goalseek_probe.py            C3 tile: 512 x (512 x 4096) winsorized, and C2: 128 x 4096^2 sigma
Per step: sigma low / high, clipped %, device pass ms in sequence (HIP events), wall ms of the synchronous step, steady-state
pass ms, pixels replayed.  The last lines compare nl_stack_find_sigmas' wall clock with the sums."""
import os
import sys
import time

import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightlight_amd.stack import StackHandle


def bisection(perc_low, perc_high, total):
    """nl::SigmaBisection (stack_kernels.h) -- a generator fed with counters."""
    f = np.float32
    ll, lr, hl, hr = f(1.0), f(11.0), f(1.0), f(11.0)
    lm, hm = f(0.5) * (ll + lr), f(0.5) * (hl + hr)
    i = 0
    while True:
        cl, ch = yield float(lm), float(hm)
        pl = f(cl) * f(100.0) / f(total)
        ph = f(ch) * f(100.0) / f(total)
        dl = int(f(100) * pl + f(0.5)) - int(f(100) * f(perc_low))
        dh = int(f(100) * ph + f(0.5)) - int(f(100) * f(perc_high))
        if (dl == 0 and dh == 0) or i >= 20:
            return
        i += 1
        if dl > 0:
            ll = lm; lm = f(0.5) * (ll + lr)
        elif dl < 0:
            lr = lm; lm = f(0.5) * (ll + lr)
        if dh > 0:
            hl = hm; hm = f(0.5) * (hl + hr)
        elif dh < 0:
            hr = hm; hm = f(0.5) * (hl + hr)


def step_table(st, mode, perc_low=0.5, perc_high=0.5):
    """The bisection on `st`, step by step: [{sigma_low, sigma_high, clipped_pct_low/high, pass_ms (in sequence), wall_ms,
    steady_pass_ms (third of three passes at the same sigmas), replayed_px}], and the steady pass at kappa = 3."""
    total = st.tile_pixels * st.n_frames
    steps = []
    g = bisection(perc_low, perc_high, total)
    sig = next(g)
    try:
        while True:
            t1 = time.perf_counter()
            st.run_async(mode, sig[0], sig[1], 0.0)
            c = st.finish()
            wall = (time.perf_counter() - t1) * 1e3
            steps.append({"sigma_low": sig[0], "sigma_high": sig[1], "clipped_pct_low": round(100.0 * c[0] / total, 4),
                          "clipped_pct_high": round(100.0 * c[1] / total, 4), "pass_ms": round(st.pass_times(0)[0], 3),
                          "wall_ms": round(wall, 3), "replayed_px": st.last_fallback_pixels})
            sig = g.send(c)
    except StopIteration:
        pass
    for s_ in steps:
        for _ in range(3):
            st.run_async(mode, s_["sigma_low"], s_["sigma_high"], 0.0)
        st.finish()
        s_["steady_pass_ms"] = round(st.pass_times(0)[0], 3)
    for _ in range(3):
        st.run_async(mode, 3.0, 3.0, 0.0)
    st.finish()
    return steps, st.pass_times(0)[0]


def main():
    for n, rows, mode, geo in ((512, 512, 3, dict(row0=1536, rows=512)), (128, 4096, 2, {})):
        with StackHandle(n, 4096, 4096, **geo) as st:
            st.fill_synthetic()
            st.find_sigmas(mode, 0.5, 0.5, fetch=False)                 # warm
            t0 = time.perf_counter()
            _, cl, ch, sl, sh, passes = st.find_sigmas(mode, 0.5, 0.5, fetch=False)
            t_lib = (time.perf_counter() - t0) * 1e3
            steps, k3 = step_table(st, mode)
            print("mode %d, %d x %dx4096: nl_stack_find_sigmas %.2f ms for %d passes -> sigma %.4f / %.4f (clipped %d / %d); steady pass at kappa 3: %.3f ms"
                  % (mode, n, rows, t_lib, passes, sl, sh, cl, ch, k3))
            print("  step  sigma low/high     clipped %% low/high   in-sequence pass ms  step wall ms  steady pass ms  replayed px")
            for i, s_ in enumerate(steps):
                print("  %2d   %7.4f %7.4f    %6.3f %6.3f        %8.3f          %8.3f      %8.3f     %8d"
                      % (i + 1, s_["sigma_low"], s_["sigma_high"], s_["clipped_pct_low"], s_["clipped_pct_high"], s_["pass_ms"],
                         s_["wall_ms"], s_["steady_pass_ms"], s_["replayed_px"]))
            print("  sums: in-sequence device passes %.2f ms, step walls %.2f ms, steady-state passes at the same sigmas %.2f ms, %d x kappa-3 pass %.2f ms"
                  % (sum(s_["pass_ms"] for s_ in steps), sum(s_["wall_ms"] for s_ in steps), sum(s_["steady_pass_ms"] for s_ in steps),
                     len(steps), len(steps) * k3))


if __name__ == "__main__":
    main()
