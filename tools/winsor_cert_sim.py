#!/usr/bin/env python3
"""Host simulation (float64): how many winsorization rounds the reference's loop (stack.go:649-672) runs per pixel
and pass on the bench's synthetic distribution, and at which round an INVARIANT-INTERVAL certificate would end it:
a trial value L <= std_k with 1.134*stddev(copy clamped at median -/+ 1.5 L) >= L bounds every later std from below
(the clamps only tighten, the variance is monotone in the clamp), std_k bounds them from above; the loop may be left
as soon as the clip decisions agree over [L, std_k]."""
import sys
import numpy as np

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
P = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
first_trial = int(sys.argv[3]) if len(sys.argv) > 3 else 3
every = int(sys.argv[4]) if len(sys.argv) > 4 else 2
mu = float(sys.argv[5]) if len(sys.argv) > 5 else 0.5
sig = 3.0
rng = np.random.default_rng(7)
k = np.arange(n)
sigma = 30.0 * (1.0 + 0.5 * (k % 7) / 6.0)
x = 1000.0 + 5.0 * np.sin(k) + (1.0 + 0.02 * np.cos(1.7 * k)) * 100.0 + sigma * rng.standard_normal((P, n))
uo = rng.random((P, n)); um = rng.random((P, n))
x = np.where(uo < 0.004, x + 300 + 19700 * um, np.where(uo < 0.005, x - 100 - 800 * um, x))
x = np.float32(x).astype(np.float64)

def g(vals, alive, med, s):
    lo = med - 1.5 * s; hi = med + 1.5 * s
    w = np.clip(vals, lo[:, None], hi[:, None])
    cnt = alive.sum(1)
    m = (w * alive).sum(1) / cnt
    var = (((w - m[:, None]) ** 2) * alive).sum(1) / cnt      # MeanStdDev: population variance
    return 1.134 * np.sqrt(var)

alive = np.ones((P, n), bool)
STAT = [0, 0, 0]
STAT0 = [0, 0]
C0 = float(sys.argv[6]) if len(sys.argv) > 6 else 0.0
live = np.ones(P, bool)
tot_rounds = np.zeros(P, int); tot_cert = np.zeros(P, int); passes = np.zeros(P, int)
hist_faith = []; hist_cert = []
for pas in range(12):
    idx = np.nonzero(live)[0]
    if len(idx) == 0: break
    v = x[idx]; al = alive[idx]
    cnt = al.sum(1)
    big = np.where(al, v, np.inf); srt = np.sort(big, 1)
    med = np.where(cnt % 2 == 1, srt[np.arange(len(idx)), cnt // 2], 0.5 * (srt[np.arange(len(idx)), cnt // 2 - 1] + srt[np.arange(len(idx)), np.minimum(cnt // 2, n - 1)]))
    m0 = (v * al).sum(1) / cnt
    s = np.sqrt((((v - m0[:, None]) ** 2) * al).sum(1) / cnt)
    # faithful loop with composed clamps
    L_eff = np.full(len(idx), -np.inf); H_eff = np.full(len(idx), np.inf)
    inner = np.ones(len(idx), bool)
    rounds = np.zeros(len(idx), int)
    cert_round = np.full(len(idx), -1)            # rounds executed (faithful + trials) when the certificate ended the loop
    trials = np.zeros(len(idx), int)
    s_hist = [s.copy()]
    s_cur = s.copy()
    final = s.copy()
    if C0 > 0:
        Ltry = C0 * s
        gl = g(v, al, med, Ltry)
        okL = gl >= Ltry * (1 + 1e-5)
        amb_lo = ((v < (med - sig * Ltry)[:, None]) & (v >= (med - sig * s)[:, None]) & al).sum(1)
        amb_hi = ((v > (med + sig * Ltry)[:, None]) & (v <= (med + sig * s)[:, None]) & al).sum(1)
        ok = okL & (amb_lo == 0) & (amb_hi == 0)
        trials = trials + 1
        cert_round = np.where(ok, 1, cert_round)
        STAT0[0] += len(idx); STAT0[1] += ok.sum()
    for r in range(1, 400):
        if not inner.any(): break
        lo = med - 1.5 * s_cur; hi = med + 1.5 * s_cur
        w = np.clip(v, L_eff[:, None], H_eff[:, None])
        changed = (((w < lo[:, None]) | (w > hi[:, None])) & al).sum(1)
        L_eff = np.where(inner, np.maximum(L_eff, lo), L_eff); H_eff = np.where(inner, np.minimum(H_eff, hi), H_eff)
        w = np.clip(v, L_eff[:, None], H_eff[:, None])
        mm = (w * al).sum(1) / cnt
        s_new = 1.134 * np.sqrt((((w - mm[:, None]) ** 2) * al).sum(1) / cnt)
        factor = np.abs(s_new - s_cur) / np.where(s_cur > 0, s_cur, 1)
        stop = (changed == 0) | (factor <= 0.0005) | (s_cur == 0)
        rounds = np.where(inner, r, rounds)
        s_prev = s_cur
        s_cur = np.where(inner, s_new, s_cur)
        s_hist.append(s_cur.copy())
        inner_next = inner & ~stop
        # certificate trial after this round for lanes still inside
        if r >= first_trial and (r - first_trial) % every == 0:
            todo = inner_next & (cert_round < 0)
            if todo.any():
                s2, s1, s0 = s_hist[-1], s_hist[-2], s_hist[-3] if len(s_hist) >= 3 else s_hist[-2]
                d1 = s1 - s2; d0 = s0 - s1
                rr = np.clip(np.where(d0 > 0, d1 / np.where(d0 > 0, d0, 1), 0.9), 0.0, 0.95)
                rest = d1 * rr / (1 - rr)
                Ltry = np.maximum(s2 - (1 + mu) * rest - 1e-4 * s2, 0.0)
                gl = g(v, al, med, Ltry)
                okL = gl >= Ltry * (1 + 1e-5)
                amb_lo = ((v < (med - sig * Ltry)[:, None]) & (v >= (med - sig * s2)[:, None]) & al).sum(1)
                amb_hi = ((v > (med + sig * Ltry)[:, None]) & (v <= (med + sig * s2)[:, None]) & al).sum(1)
                ok = todo & okL & (amb_lo == 0) & (amb_hi == 0)
                STAT[0] += todo.sum(); STAT[1] += (todo & ~okL).sum(); STAT[2] += (todo & okL & ~ok).sum()
                trials = trials + todo
                cert_round = np.where(ok, r + trials, cert_round)
        inner = inner_next
    eff = np.where(cert_round >= 0, cert_round, rounds + trials)
    hist_faith.append(rounds); hist_cert.append(eff)
    tot_rounds[idx] += rounds; tot_cert[idx] += eff; passes[idx] += 1
    # clip
    lo = med - sig * s_cur; hi = med + sig * s_cur
    clip = ((v < lo[:, None]) | (v > hi[:, None])) & al
    anyclip = clip.any(1)
    alive[idx] = al & ~clip
    live[idx] = anyclip & (alive[idx].sum(1) > 1)
    print("pass %d: pixels %d  faithful rounds mean %.2f q90 %d q99 %d max %d | with certificate mean %.2f q90 %d q99 %d max %d | certified %.3f"
          % (pas + 1, len(idx), rounds.mean(), np.quantile(rounds, .9), np.quantile(rounds, .99), rounds.max(),
             eff.mean(), np.quantile(eff, .9), np.quantile(eff, .99), eff.max(), (cert_round >= 0).mean()))
def wave(a):   # max over groups of 64
    m = len(a) // 64 * 64
    return a[:m].reshape(-1, 64).max(1).mean()
print("n %d: per pixel total rounds faithful %.2f, certificate %.2f; passes %.2f" % (n, tot_rounds.mean(), tot_cert.mean(), passes.mean()))
print("   per wave (64 px, pass 1 only): faithful %.1f, certificate %.1f" % (wave(hist_faith[0]), wave(hist_cert[0])))
print("   per wave total rounds (sum over passes, lockstep): faithful %.1f, certificate %.1f" % (wave(tot_rounds), wave(tot_cert)))
print("   trials %d: L not invariant %.3f, invariant but clip ambiguous %.3f" % (STAT[0], STAT[1] / STAT[0], STAT[2] / STAT[0]))
if C0 > 0: print("   round-0 trials %d, certified %.3f" % (STAT0[0], STAT0[1] / STAT0[0]))
