#!/usr/bin/env python3
"""What the host of this box lets a process use, and how the CPU oracle scales on it.

Prints one JSON object: hardware threads, the affinity mask, the cgroup CPU quota (v2 cpu.max / v1 cfs quota), NUMA nodes,
load average, and a thread sweep of the oracle (oracle/nl_oracle.c, sigma clipping, 128 frames) on a strip first-touched by
worker threads -- Msamples/s per thread count.  bench.py's cpu_baseline uses the same helpers (host_limits, thread_sweep).

    python tools/cpu_probe.py [--frames 128] [--rows 64] [--width 4096] [--mode 2]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _read(path):
    try:
        return open(path).read().strip()
    except OSError:
        return None


def host_limits():
    """Every limit that can hold a process below os.cpu_count() threads of real CPU."""
    out = {"os_cpu_count": os.cpu_count()}
    try:
        out["affinity"] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        out["affinity"] = None
    quota = None
    v2 = _read("/sys/fs/cgroup/cpu.max")
    if v2:
        out["cgroup_cpu_max"] = v2
        a = v2.split()
        if a[0] != "max":
            quota = float(a[0]) / float(a[1])
    else:
        q, p = _read("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), _read("/sys/fs/cgroup/cpu/cpu.cfs_period_us")
        if q and p:
            out["cgroup_cfs_quota_us"], out["cgroup_cfs_period_us"] = q, p
            if int(q) > 0:
                quota = float(q) / float(p)
    out["cgroup_cpu_quota"] = quota
    out["cgroup_cpuset"] = _read("/sys/fs/cgroup/cpuset.cpus.effective") or _read("/sys/fs/cgroup/cpuset/cpuset.effective_cpus")
    try:
        out["numa_nodes"] = len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()])
    except OSError:
        out["numa_nodes"] = None
    try:
        out["loadavg"] = [round(x, 2) for x in os.getloadavg()]
    except OSError:
        out["loadavg"] = None
    usable = out["affinity"] or out["os_cpu_count"] or 1
    if quota:
        usable = max(1, min(usable, int(quota + 0.5)))
    out["usable_threads"] = usable
    return out


def sweep_counts(usable, physical=None):
    c = {1, 2, 4, 8, 16, 32, 64, 128, 256, usable}
    if physical:
        c.add(physical)
    return sorted(x for x in c if x <= max(usable, 1))


def make_strip(n, rows, width, workers):
    """N separately allocated host frames (as fits.Image.Data is), each first-touched by a worker thread -- the reference's
    frames are allocated by its loader goroutines (operator.go:73-116), not by one thread on one NUMA node."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    frames = [None] * n

    def fill(i):
        rng = np.random.default_rng(1000 + i)
        a = np.empty(rows * width, np.float32)
        a[:] = rng.standard_normal(rows * width, dtype=np.float32) * 30.0 + 1000.0
        # a few outliers and missing samples, as the synthetic stacks have
        idx = rng.integers(0, a.size, a.size // 200)
        a[idx] += 2000.0
        a[rng.integers(0, a.size, a.size // 1000)] = np.nan
        frames[i] = a
    with ThreadPoolExecutor(max(1, workers)) as ex:
        list(ex.map(fill, range(n)))
    return frames


def thread_sweep(frames, mode, kappa, counts, weights=None, repeats=2, budget_s=40.0, pin=True):
    """Oracle rate per thread count: best of `repeats` runs each; stops when the time budget is spent.  pin: worker t of
    the pool sits on the t-th allowed CPU (a fresh pthread pool is otherwise at the mercy of the load balancer)."""
    from oracle import oracle
    oracle.set_pin_workers(pin)
    n, npix = len(frames), frames[0].size
    rows, t_start = [], time.perf_counter()
    for c in counts:
        best = None
        for _ in range(repeats):
            t0 = time.perf_counter()
            rc, _, _, _, _ = oracle.stack_apply(mode, frames, weights, kappa, kappa, 0.0, num_cpu=c)
            dt = time.perf_counter() - t0
            assert rc == 0
            best = dt if best is None else min(best, dt)
        rows.append({"threads": c, "s": round(best, 4), "msamples_per_s": round(n * npix / best / 1e6, 1),
                     "per_thread": round(n * npix / best / 1e6 / c, 2)})
        if time.perf_counter() - t_start > budget_s:
            break
    oracle.set_pin_workers(False)
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=128)
    ap.add_argument("--rows", type=int, default=64)
    ap.add_argument("--width", type=int, default=4096)
    ap.add_argument("--mode", type=int, default=2)
    ap.add_argument("--budget", type=float, default=60.0)
    a = ap.parse_args()
    lim = host_limits()
    frames = make_strip(a.frames, a.rows, a.width, min(lim["usable_threads"], 32))
    lim["sweep_pinned"] = thread_sweep(frames, a.mode, 3.0, sweep_counts(lim["os_cpu_count"] or 1), budget_s=a.budget / 2)
    lim["sweep_unpinned"] = thread_sweep(frames, a.mode, 3.0, sweep_counts(lim["os_cpu_count"] or 1), budget_s=a.budget / 2, pin=False)
    lim["sample"] = "%d frames x %d rows x %d px, mode %d" % (a.frames, a.rows, a.width, a.mode)
    print(json.dumps(lim, indent=1))


if __name__ == "__main__":
    main()
