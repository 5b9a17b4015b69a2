"""Second, independent restatement of the stacking path in pure Python with
numpy float32 scalars (every operation rounds to fp32, no FMA).

TEST INFRASTRUCTURE ONLY.  Written from the language-neutral operation order
in SURVEY.md Appendix B (B0-B8) and the reference lines cited there; it shares
no code with oracle/nl_oracle.c.  It is slow (Python loops) and only used on
small cases: tests require the C oracle and this file to agree bit for bit,
which is the strongest pin available for the Stack* functions (the reference
has no tests for them and no Go toolchain exists here).
"""
import math

import numpy as np

F = np.float32
_ZERO = F(0.0)


def _isnan(v):
    return v != v


# B1 -- internal/qsort/qsort.go:94-126
def hoare_select(a, lo, hi, k):
    """k-th smallest (1-based) of a[lo..hi], in place; returns the value."""
    left, right = lo, hi
    while left < right:
        pivot = a[(left + right) >> 1]
        l, r = left - 1, right + 1
        while True:
            l += 1
            while not (a[l] >= pivot):
                l += 1
            r -= 1
            while not (a[r] <= pivot):
                r -= 1
            if l >= r:
                break
            a[l], a[r] = a[r], a[l]
        index = r
        offset = index - left + 1
        if k <= offset:
            right = index
        else:
            left = index + 1
            k -= offset
    return a[left]


# qsort.go:68-82
def select_median(a, n):
    k = (n >> 1) + 1
    upper = hoare_select(a, 0, n - 1, k)
    if n & 1:
        return upper
    lower = a[0]
    for i in range(1, k - 1):
        if a[i] > lower:
            lower = a[i]
    return F(0.5) * (lower + upper)


# qsort.go:26-56
def hoare_sort(a, lo, hi):
    """sort a[lo..hi] inclusive, same partition order as the reference."""
    stack = [(lo, hi)]
    while stack:
        lo, hi = stack.pop()
        if hi - lo + 1 <= 1:
            continue
        pivot = a[(lo + hi) >> 1]
        l, r = lo - 1, hi + 1
        while True:
            l += 1
            while not (a[l] >= pivot):
                l += 1
            r -= 1
            while not (a[r] <= pivot):
                r -= 1
            if l >= r:
                break
            a[l], a[r] = a[r], a[l]
        # the two halves are disjoint, so their order of processing is irrelevant
        stack.append((lo, r))
        stack.append((r + 1, hi))


# B2 -- internal/stats/stats.go:246-261
def mean_std(a, lo, n):
    s = _ZERO
    for i in range(lo, lo + n):
        s = F(s + a[i])
    mean = F(s / F(n))
    v = _ZERO
    for i in range(lo, lo + n):
        d = F(a[i] - mean)
        v = F(v + F(d * d))
    v = F(v / F(n))
    return mean, F(math.sqrt(float(v)))


# B0 -- stack.go:380-387
def gather(frames, i, weights=None):
    vals, ws = [], []
    for li in range(len(frames)):
        v = F(frames[li][i])
        if not _isnan(v):
            vals.append(v)
            if weights is not None:
                ws.append(F(weights[li]))
    return vals, ws


def _clip(a, w, n, lo, hi, cnt):
    j = 0
    while j < n:
        g = a[j]
        if g < lo:
            a[j] = a[n - 1]
            if w:
                w[j] = w[n - 1]
            n -= 1
            cnt[0] += 1
        elif g > hi:
            a[j] = a[n - 1]
            if w:
                w[j] = w[n - 1]
            n -= 1
            cnt[1] += 1
        else:
            j += 1
    return n


def _wmean(a, w, n):
    s, ws = _ZERO, _ZERO
    for i in range(n):
        s = F(s + F(a[i] * w[i]))
        ws = F(ws + w[i])
    return F(s / ws)


def _winsor_std(a, n, med, std):
    # B4 inner loop -- stack.go:646-672
    wz = list(a[:n])
    while True:
        t = F(F(1.5) * std)
        lo, hi = F(med - t), F(med + t)
        changed = 0
        for i in range(n):
            if wz[i] < lo:
                wz[i] = lo
                changed += 1
            elif wz[i] > hi:
                wz[i] = hi
                changed += 1
        old = std
        _, std = mean_std(wz, 0, n)
        std = F(F(1.134) * std)
        with np.errstate(all="ignore"):
            factor = F(F(abs(float(F(std - old)))) / old)
        if changed == 0 or factor <= F(0.0005):
            return std


def stack_pixel(mode, vals, ws, sig_lo, sig_hi, cnt):
    """one pixel; vals = gathered non-NaN values (list of np.float32),
    ws = matching weights or []; cnt = [clipLow, clipHigh] updated in place."""
    a = list(vals)
    w = list(ws)
    n = len(a)
    sig_lo, sig_hi = F(sig_lo), F(sig_hi)
    if mode == 0:                      # median, stack.go:274-303
        return select_median(a, n)
    if mode == 1:                      # mean, stack.go:307-366
        if w:
            return _wmean(a, w, n)
        s = _ZERO
        for v in a:
            s = F(s + v)
        return F(s / F(n))
    if mode in (2, 3):                 # sigma / winsorized sigma, B3 / B4
        while True:
            med = select_median(a, n)
            mean, std = mean_std(a, 0, n)
            if mode == 3:
                std = _winsor_std(a, n, med, std)
            lo = F(med - F(sig_lo * std))
            hi = F(med + F(sig_hi * std))
            prev = cnt[0] + cnt[1]
            n = _clip(a, w, n, lo, hi, cnt)
            if cnt[0] + cnt[1] == prev or n <= 1:
                return _wmean(a, w, n) if w else mean
    if mode == 4:                      # MAD sigma, B5
        med = select_median(a, n)
        ad = []
        for i in range(n):
            d = F(a[i] - med)
            if d < 0:
                d = F(-d)
            ad.append(d)
        mad = select_median(ad, n)
        std = F(mad * F(1.4826))
        lo = F(med - F(sig_lo * std))
        hi = F(med + F(sig_hi * std))
        n = _clip(a, None, n, lo, hi, cnt)
        s = _ZERO
        for i in range(n):
            s = F(s + a[i])
        with np.errstate(all="ignore"):
            return F(s / F(n))
    if mode == 5:                      # linear fit, B6
        base = 0
        mean = _ZERO
        while True:
            hoare_sort(a, base, base + n - 1)
            xs = [F(i) for i in range(n)]
            xm, xsd = mean_std(xs, 0, n)
            ym, ysd = mean_std(a, base, n)
            c = _ZERO
            for i in range(n):
                c = F(c + F(F(xs[i] - xm) * F(a[base + i] - ym)))
            with np.errstate(all="ignore"):
                c = F(c / F(F(xsd * ysd) * F(F(n) + F(1))))
                slope = F(F(c * ysd) / xsd)
            icpt = F(ym - F(slope * xm))
            mean = ym
            sg = _ZERO
            for i in range(n):
                lin = F(F(F(i) * slope) + icpt)
                sg = F(sg + F(abs(float(F(a[base + i] - lin)))))
            sg = F(sg / F(n))
            left = 0
            lb, hb = F(sig_lo * sg), F(sig_hi * sg)
            for i in range(n):
                g = a[base + i]
                lin = F(F(F(i) * slope) + icpt)
                if F(lin - g) > lb:
                    a[base + i] = a[base + left]
                    left += 1
                    cnt[0] += 1
                elif F(g - lin) > hb:
                    a[base + i] = a[base + left]
                    left += 1
                    cnt[1] += 1
            if left == 0 or n < 3:
                return mean
            base += left
            n -= left
    raise ValueError("invalid stacking mode")


def stack(mode, frames, weights=None, sig_lo=2.75, sig_hi=2.75, ref_loc=0.0):
    """frames [N, P] -> (result[P], clipLow, clipHigh).  Linear fit ignores
    weights (stack.go:188-189)."""
    frames = np.asarray(frames, dtype=np.float32)
    npix = frames.shape[1]
    if mode == 5:
        weights = None
    out = np.empty(npix, np.float32)
    cnt = [0, 0]
    for i in range(npix):
        vals, ws = gather(frames, i, weights)
        if not vals:
            out[i] = F(ref_loc)
            continue
        out[i] = stack_pixel(mode, vals, ws, sig_lo, sig_hi, cnt)
    return out, cnt[0], cnt[1]
