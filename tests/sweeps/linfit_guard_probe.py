#!/usr/bin/env python3
"""(Lives under tests/: it restates the reference's StackLinearFit, stack.go:834-918 + stats.go:569-586, in numpy fp32.)
Would a guarded, non-bit-exact FAST STAGE pay for the linear fit?  CPU only.

The fast stage would keep a pixel while (a) its rejects are a prefix / suffix of the sorted alive column (moments of a
contiguous window) and (b) every reject decision is the same over a rigorous enclosure of the reference's fp32
slope / intercept / sigma; at the first iteration where either fails the pixel restarts in the bit-exact cascade.
This script replays the reference's iterations (sequential fp32 sums, the reference's operation order) on the bench
stack's distribution (synth.hip: sky + gradient, per-frame noise 30 .. 45 ADU, 0.4 % hot / 0.1 % cold outliers) and
reports, per iteration, how many pixels are still iterating and how many would be handed over, by cause.
usage: linfit_guard_probe.py [pixels] [frames] [kappa]"""
import sys
import numpy as np

P = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 128
kappa = np.float32(sys.argv[3]) if len(sys.argv) > 3 else np.float32(3.0)
f32 = np.float32
u = f32(2.0 ** -24)
rng = np.random.default_rng(7)

k = np.arange(N)
bg = 1000.0 + 5.0 * np.sin(k)
gain = 1.0 + 0.02 * np.cos(1.7 * k)
sig = 30.0 * (1.0 + 0.5 * (k % 7) / 6.0)
sky = 200.0 * rng.uniform(0.0, 1.0, P)
v = bg[None, :] + gain[None, :] * sky[:, None] + sig[None, :] * rng.standard_normal((P, N))
uo = rng.uniform(size=(P, N))
um = rng.uniform(size=(P, N))
v = np.where(uo < 0.004, v + 300.0 + 19700.0 * um, np.where(uo < 0.005, v - (100.0 + 800.0 * um), v))
ys = np.sort(v.astype(f32), axis=1)
n = np.full(P, N, np.int64)
idx = np.arange(N, dtype=f32)

# MeanStdDev of xs = 0 .. m-1 for every m, in the reference's order
xm_t = np.zeros(N + 1, f32)
xs_t = np.zeros(N + 1, f32)
for m in range(1, N + 1):
    xs = np.arange(m, dtype=f32)
    s = np.cumsum(xs, dtype=f32)[-1]
    mean = f32(s / f32(m))
    d = (xs - mean).astype(f32)
    var = f32(np.cumsum((d * d).astype(f32), dtype=f32)[-1] / f32(m))
    xm_t[m], xs_t[m] = mean, f32(np.sqrt(np.float64(var)))

iterating = np.ones(P, bool)            # the reference is still looping on this pixel
handed = np.zeros(P, bool)              # the fast stage has given this pixel to the cascade
print("pixels %d, frames %d, kappa %g" % (P, N, kappa))
print("iter  still iterating  handed over now (holes / doubt / both)   handed over so far   of those still iterating: handed")
for it in range(1, 40):
    if not iterating.any():
        break
    valid = idx[None, :] < n[:, None]
    fn = n.astype(f32)
    y0 = np.where(valid, ys, f32(0))
    ym = (np.cumsum(y0, axis=1, dtype=f32)[:, -1] / fn).astype(f32)
    d = np.where(valid, (ys - ym[:, None]).astype(f32), f32(0))
    yvar = (np.cumsum((d * d).astype(f32), axis=1, dtype=f32)[:, -1] / fn).astype(f32)
    ysd = np.sqrt(yvar.astype(np.float64)).astype(f32)
    xm, xsd = xm_t[n], xs_t[n]
    dx = np.where(valid, (idx[None, :] - xm[:, None]).astype(f32), f32(0))
    prod = (dx * d).astype(f32)
    C = np.cumsum(prod, axis=1, dtype=f32)[:, -1]
    denom = (xsd * ysd).astype(f32)
    denom = (denom * (fn + f32(1))).astype(f32)
    with np.errstate(divide="ignore", invalid="ignore"):
        corr = (C / denom).astype(f32)
        sl = ((corr * ysd).astype(f32) / xsd).astype(f32)
    icpt = (ym - (sl * xm).astype(f32)).astype(f32)
    lin = ((idx[None, :] * sl[:, None]).astype(f32) + icpt[:, None]).astype(f32)
    diff = np.where(valid, (ys - lin).astype(f32), f32(0))
    sigma = (np.cumsum(np.abs(diff), axis=1, dtype=f32)[:, -1] / fn).astype(f32)
    lb, hb = (kappa * sigma).astype(f32), (kappa * sigma).astype(f32)
    r_lo = (lin - ys).astype(f32)
    r_hi = (ys - lin).astype(f32)
    rej_lo = valid & (r_lo > lb[:, None])
    rej_hi = valid & ~rej_lo & (r_hi > hb[:, None])
    rej = rej_lo | rej_hi
    # (a) rejects must be a prefix and a suffix of the alive column
    keep = valid & ~rej
    first = np.argmax(keep, axis=1)
    last = N - 1 - np.argmax(keep[:, ::-1], axis=1)
    inside = (idx[None, :] > first[:, None]) & (idx[None, :] < last[:, None])
    holes = (rej & inside).any(axis=1) & keep.any(axis=1)
    # (b) rigorous enclosure of the reference's values (any summation order for the fast stage's own sums):
    # ymean off by <= n u |ymean|, slope by <= (n + 8) u relative (correlation sum + 5 operations), sigma by
    # <= (n + 3) u relative plus the mean shift of lin
    eps = (fn + f32(8)) * u
    dlin = (idx[None, :] * (np.abs(sl) * eps)[:, None] + (1.01 * fn * u * np.abs(ym) + xm * np.abs(sl) * eps)[:, None]
            + 4 * u * np.abs(lin))
    dsig = (fn + 3) * u * sigma + np.where(valid, dlin, 0).sum(axis=1) / fn
    band = dlin + (kappa * dsig)[:, None] + 2 * u * (np.abs(r_lo) + lb[:, None])
    doubt = (valid & ((np.abs(r_lo - lb[:, None]) <= band) | (np.abs(r_hi - hb[:, None]) <= band))).any(axis=1)
    live = iterating & ~handed
    now_h = live & holes & ~doubt
    now_d = live & doubt & ~holes
    now_b = live & doubt & holes
    handed |= now_h | now_d | now_b
    print("%4d  %8d (%5.1f %%)  %7d / %7d / %7d (%.2f %% of all)      %5.1f %%        %5.1f %%"
          % (it, iterating.sum(), 100.0 * iterating.mean(), now_h.sum(), now_d.sum(), now_b.sum(),
             100.0 * (now_h.sum() + now_d.sum() + now_b.sum()) / P, 100.0 * handed.mean(),
             100.0 * (handed & iterating).sum() / max(iterating.sum(), 1)))
    # the reference's next iteration: drop the rejects (the sorted order of the rest is unchanged), stop where nothing
    # was rejected or fewer than 3 samples were in play
    nrej = rej.sum(axis=1)
    stop = iterating & ((nrej == 0) | (n < 3))
    order = np.argsort(~keep, axis=1, kind="stable")
    ys = np.take_along_axis(np.where(keep, ys, f32(np.inf)), order, axis=1)
    upd = iterating & ~stop
    n = np.where(upd, keep.sum(axis=1), n)
    iterating &= ~stop
print("pixels the fast stage would finish: %.1f %%; handed to the bit-exact cascade: %.1f %%" % (100.0 * (~handed).mean(), 100.0 * handed.mean()))
