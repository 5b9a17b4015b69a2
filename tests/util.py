"""Shared helpers for the parity tests: seeded sub-exposure stacks with the
features the reference's input contract allows (NaN = no data after
alignment, hot/cold outliers, ties, per-frame gain and noise)."""
import numpy as np


def make_frames(n, width, height, seed, nan_frac=0.06, hot=0.01, cold=0.004,
                ties=False, nan_border=True, all_nan_patch=True):
    rng = np.random.default_rng(seed)
    p = width * height
    yy, xx = np.mgrid[0:height, 0:width]
    sky = (200.0 * (0.5 * xx / width + 0.5 * yy / height)).astype(np.float32).reshape(-1)
    frames = np.empty((n, p), np.float32)
    for k in range(n):
        bg = np.float32(1000.0 + 5.0 * np.sin(k))
        gain = np.float32(1.0 + 0.02 * np.cos(1.7 * k))
        sigma = np.float32(30.0 * (1.0 + 0.5 * (k % 7) / 6.0))
        f = bg + gain * sky + sigma * rng.standard_normal(p).astype(np.float32)
        u = rng.random(p)
        f = np.where(u < hot, f + np.float32(300.0) + np.float32(19700.0) * rng.random(p).astype(np.float32), f)
        f = np.where((u >= hot) & (u < hot + cold), f - np.float32(100.0) - np.float32(800.0) * rng.random(p).astype(np.float32), f)
        f = f.astype(np.float32)
        if ties:
            f = (np.round(f / np.float32(16.0)) * np.float32(16.0)).astype(np.float32)
        f[rng.random(p) < nan_frac] = np.nan
        if nan_border:
            img = f.reshape(height, width)
            img[: (k % 9) % max(height, 1), :] = np.nan
            if k % 5:
                img[:, width - (k % 5):] = np.nan
        frames[k] = f
    if all_nan_patch and p >= 4:
        frames[:, p // 2: p // 2 + 3] = np.nan      # pixels with no data in any frame
    return frames


def bits_equal(a, b):
    """bit-exact equality of two float32 arrays (NaN payloads included)."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def same_values(a, b):
    """equal as numbers, NaN == NaN (sign of zero / NaN payload ignored)."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


def describe_mismatch(got, want, limit=5):
    bad = np.flatnonzero(~((got == want) | (np.isnan(got) & np.isnan(want))))
    rows = ["%d: got %r want %r" % (i, got[i], want[i]) for i in bad[:limit]]
    return "%d mismatching pixels; %s" % (bad.size, "; ".join(rows))
