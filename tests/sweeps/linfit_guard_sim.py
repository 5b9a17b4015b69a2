#!/usr/bin/env python3
"""(Lives under tests/: it restates the reference's StackLinearFit, stack.go:834-918 + stats.go:569-586, in numpy fp32.)
Round 5: simulation of a STATE-CARRYING guarded stage for the linear fit.  CPU only.

Per iteration the guarded stage
  * forms the reference's ymean EXACTLY (the sequential fp32 sum in sorted order: one add per sample),
  * keeps D = sum(y-c), Q = sum (y-c)^2, P = sum rank*(y-c) of the alive samples and updates them by the rejected
    samples (O(rejects); a removal strictly inside the alive range needs a suffix sum: counted as `mid`),
  * derives slope0 = (P - xm D) / (xsd^2 (m+1)) and a RIGOROUS enclosure of the reference's fp32 slope (its
    correlation sum is a sequential sum of products: gamma_(m+1) * sum |terms|, Cauchy-Schwarz for the latter; five
    more roundings; the reference's ystddev cancels up to those roundings),
  * evaluates the residuals against its own line (one fma + one subtraction per sample), their absolute sum, and
    decides every reject whose distance from the threshold exceeds the enclosure's half-width; a pixel with an
    undecidable sample is handed to the bit-exact cascade WITH its alive mask and counters (it continues there, it
    does not start again).
Reported: pixels still iterating, handed over per iteration (first time), removals inside the alive range, checks
that the reference's slope / sigma always lie inside the enclosures and that every decided reject equals the
reference's, and an instruction-count model of a cascade of such stages against today's bit-exact cascade.
usage: linfit_guard_sim.py [pixels] [frames] [kappa] [scale]"""
import sys
import numpy as np

P_ = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 128
kappa = np.float32(sys.argv[3]) if len(sys.argv) > 3 else np.float32(3.0)
scale = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
f32, f64 = np.float32, np.float64
u = 2.0 ** -24
rng = np.random.default_rng(7)


def gamma(k):
    return k * u / (1.0 - k * u)


k = np.arange(N)
bg = 1000.0 + 5.0 * np.sin(k)
gain = 1.0 + 0.02 * np.cos(1.7 * k)
sig = 30.0 * (1.0 + 0.5 * (k % 7) / 6.0)
sky = 200.0 * rng.uniform(0.0, 1.0, P_)
v = bg[None, :] + gain[None, :] * sky[:, None] + sig[None, :] * rng.standard_normal((P_, N))
uo = rng.uniform(size=(P_, N))
um = rng.uniform(size=(P_, N))
v = np.where(uo < 0.004, v + 300.0 + 19700.0 * um, np.where(uo < 0.005, v - (100.0 + 800.0 * um), v))
ys = np.sort((v * scale).astype(f32), axis=1)
n = np.full(P_, N, np.int64)
idx = np.arange(N, dtype=f32)

xm_t = np.zeros(N + 1, f32)
xs_t = np.zeros(N + 1, f32)
for m in range(1, N + 1):
    xs = np.arange(m, dtype=f32)
    s = np.cumsum(xs, dtype=f32)[-1]
    mean = f32(s / f32(m))
    d = (xs - mean).astype(f32)
    var = f32(np.cumsum((d * d).astype(f32), dtype=f32)[-1] / f32(m))
    xm_t[m], xs_t[m] = mean, f32(np.sqrt(np.float64(var)))

iterating = np.ones(P_, bool)
handed = np.zeros(P_, bool)
hand_iter = np.zeros(P_, np.int64)         # iteration at which a pixel was handed over (0: never)
iters_needed = np.zeros(P_, np.int64)      # the reference's iterations
mid_iter = []                              # per iteration: pixels (still in the guarded stage) with a removal inside the alive range
c_shift = ys[:, N // 2].astype(f64)
enc_fail = dec_fail = 0
print("pixels %d, frames %d, kappa %g, scale %g" % (P_, N, kappa, scale))
print("iter  iterating   in guard   handed now   (%% of all)  handed so far   mid-removal px   band (median ADU)  thr (median)")
hist = []
for it in range(1, 60):
    if not iterating.any():
        break
    valid = idx[None, :] < n[:, None]
    fn = n.astype(f32)
    y0 = np.where(valid, ys, f32(0))
    ym = (np.cumsum(y0, axis=1, dtype=f32)[:, -1] / fn).astype(f32)
    d = np.where(valid, (ys - ym[:, None]).astype(f32), f32(0))
    yvar = (np.cumsum((d * d).astype(f32), axis=1, dtype=f32)[:, -1] / fn).astype(f32)
    ysd = np.sqrt(yvar.astype(f64)).astype(f32)
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
    keep = valid & ~rej

    # ---------------- the guarded stage's view (fp64 moments; its own fp32 line) ----------------
    mm = n.astype(f64)
    z = np.where(valid, ys.astype(f64) - c_shift[:, None], 0.0)
    D = z.sum(axis=1)
    Q = (z * z).sum(axis=1)
    Pm = (idx[None, :].astype(f64) * z).sum(axis=1)
    xm64, xsd64 = xm.astype(f64), xsd.astype(f64)
    Cg = Pm - xm64 * D
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        K = 1.0 / (xsd64 * xsd64 * (mm + 1.0))
        slope0 = (Cg * K).astype(f32).astype(f64)                     # the kernel's fp32 slope
    i32 = (ym.astype(f64) - slope0 * xm64).astype(f32)
    lin_g = (idx[None, :].astype(f64) * slope0[:, None] + i32.astype(f64)[:, None]).astype(f32)   # one fma
    res_g = np.where(valid, (ys - lin_g).astype(f32), f32(0))
    S_abs = np.abs(res_g).astype(f64).sum(axis=1)
    sg_g = S_abs / mm
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        xvar = (mm * mm - 1.0) / 12.0
        # first a provisional A_s (needs Lmax, which needs A_s: one fixed-point step with a generous start)
        Lmax0 = np.abs(ym.astype(f64)) + np.abs(slope0) * (mm - 1.0) * 0.51
        T = (mm - 1.0) / 2.0 * S_abs * (1 + gamma(40)) + np.abs(slope0) * mm * xvar + 3 * u * Lmax0 * mm * mm / 4.0
        A_C = gamma(N + 2) * T + gamma(17) * (T + np.abs(ym.astype(f64) - c_shift) * mm * mm / 4.0) + 1e-30
        A_s = 1.0001 * K * A_C + 16 * u * np.abs(slope0)
        Lmax = np.abs(ym.astype(f64)) + (np.abs(slope0) + A_s) * (mm - 1.0) / 2.0
        E = u * (1.5 * (np.abs(slope0) + A_s) * (mm - 1.0) + 5.0 * Lmax) * 1.001
    half = np.abs(idx[None, :].astype(f64) - xm64[:, None])
    W = A_s * (mm - 1.0) / 2.0 + E                                      # uniform over the samples
    A_sg = (A_s * mm / 4.0 + E) + (gamma(N) + gamma(N // 4 + 4) + 6 * u) * sg_g
    kap = abs(float(kappa))
    lbg = kap * sg_g
    band1 = 1.001 * (W + kap * A_sg + 4 * u * lbg) + 1e-30
    band = np.broadcast_to(band1[:, None], ys.shape)
    t_lo = -res_g.astype(f64) - float(kappa) * sg_g[:, None]           # > 0: low reject
    t_hi = res_g.astype(f64) - float(kappa) * sg_g[:, None]
    und = valid & ((np.abs(t_lo) <= band) | (np.abs(t_hi) <= band))
    ymax = np.abs(np.where(valid, ys, 0)).max(axis=1).astype(f64)
    finite = np.isfinite(slope0) & np.isfinite(sg_g) & (np.abs(slope0) > 2.0 ** -60) & (ymax < 2.0 ** 60) & (mm >= 2)
    doubt = und.any(axis=1) | ~finite
    # rigor checks on the pixels the guarded stage still owns and would decide
    own = iterating & ~handed
    ok = own & finite & np.isfinite(sl)
    enc_fail += int((ok & (np.abs(sl.astype(f64) - slope0) > A_s)).sum())
    enc_fail += int((ok & (np.abs(sigma.astype(f64) - sg_g) > A_sg)).sum())
    dec = own & ~doubt
    g_lo = valid & (t_lo > 0)
    g_hi = valid & ~g_lo & (t_hi > 0)
    dec_fail += int((dec & ((g_lo != rej_lo) | (g_hi != rej_hi)).any(axis=1)).sum())
    # removals strictly inside the remaining alive range
    first = np.argmax(keep, axis=1)
    last = N - 1 - np.argmax(keep[:, ::-1], axis=1)
    inside = (idx[None, :] > first[:, None]) & (idx[None, :] < last[:, None])
    mid = (rej & inside).any(axis=1) & keep.any(axis=1)
    now = own & doubt
    hand_iter[now] = it
    handed |= now
    mid_iter.append(own & ~doubt & mid)
    med_band = np.median(band[own][valid[own]]) if own.any() else 0.0
    print("%4d  %8d   %8d   %8d   (%5.2f %%)     %5.2f %%        %8d         %.2e          %.2f"
          % (it, iterating.sum(), own.sum(), now.sum(), 100.0 * now.sum() / P_, 100.0 * handed.mean(),
             (own & ~doubt & mid).sum(), med_band, np.median(lbg[own]) if own.any() else 0.0))
    nrej = rej.sum(axis=1)
    stop = iterating & ((nrej == 0) | (n < 3))
    iters_needed[iterating] = it
    order = np.argsort(~keep, axis=1, kind="stable")
    ys = np.take_along_axis(np.where(keep, ys, f32(np.inf)), order, axis=1)
    upd = iterating & ~stop
    n = np.where(upd, keep.sum(axis=1), n)
    iterating &= ~stop
print("enclosure violations: %d, wrong decided rejects: %d" % (enc_fail, dec_fail))
print("handed to the bit-exact cascade with state: %.2f %% of the pixels; mean iterations %.2f, max %d"
      % (100.0 * handed.mean(), iters_needed.mean(), iters_needed.max()))

# ---------------- instruction model: cascades of lock-step waves ----------------
# a wave runs an iteration if any of its lanes needs it; stages repack unfinished pixels
def cascade(need_from, need_to, quotas, per_iter, per_stage, label):
    """pixels enter at iteration need_from[p]+1 and finish after need_to[p]; returns wave-instructions"""
    left = need_to - need_from
    active = np.flatnonzero(left > 0)
    total = 0.0
    for q in quotas:
        if active.size == 0:
            break
        nw = (active.size + 63) // 64
        pad = np.zeros(nw * 64, np.int64)
        pad[:active.size] = left[active]
        w = pad.reshape(nw, 64).max(axis=1)
        run = np.minimum(w, q) if q else w
        total += float(run.sum()) * per_iter + nw * per_stage
        left[active] -= (q if q else left[active].max() + 1)
        active = active[left[active] > 0]
    return total


zero = np.zeros(P_, np.int64)
SORT = 2500.0                    # gather + network + start-up per wave and stage
for per_exact, per_guard, mid_extra in ((3700.0, 1300.0, 260.0), (3700.0, 1000.0, 260.0)):
    base = cascade(zero.copy(), iters_needed.copy(), (8, 6, 8, 0), per_exact, SORT, "exact")
    g_to = np.where(hand_iter > 0, hand_iter - 1, iters_needed)       # iterations the guarded stages decide
    guard = cascade(zero.copy(), g_to.copy(), (8, 8, 0), per_guard, SORT + 600.0, "guard")
    # waves that execute the suffix pass: any lane with a removal inside the range, ~ per iteration and wave of 64
    midw = sum(int(np.add.reduceat(mi.astype(np.int64), np.arange(0, P_, 64)).astype(bool).sum()) for mi in mid_iter)
    guard += midw * mid_extra
    ex_from = np.where(hand_iter > 0, hand_iter - 1, iters_needed)
    rest = cascade(ex_from.copy(), iters_needed.copy(), (8, 0), per_exact, SORT, "exact tail")
    print("model (exact %.0f / guard %.0f per wave-iteration): today %.0f k wave-instr per 64 px; guarded %.0f k + exact tail %.0f k = %.0f k  -> %.2f x"
          % (per_exact, per_guard, base / (P_ / 64) / 1e3, guard / (P_ / 64) / 1e3, rest / (P_ / 64) / 1e3,
             (guard + rest) / (P_ / 64) / 1e3, base / (guard + rest)))
