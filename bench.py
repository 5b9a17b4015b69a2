#!/usr/bin/env python3
"""bench.py -- stacked Mpixels/s of the per-pixel stacking hot path on MI355X.

One "step" = one full stack pass (OpStack.Apply's numeric core,
internal/ops/stack/stack.go:142-218) over a synthetic sub-exposure stack that
is already resident in HBM.  Default workload = BASELINE.json configs[1]:
128 x 4096 x 4096 fp32 frames, sigma-clipped mean, kappa = 3.

--gpus N: one rank per GPU (launched by torch.distributed.run; when WORLD_SIZE
is not set this script re-executes itself under it).  STRONG scaling, as
BASELINE.json's metric asks: the image stays 4096 x 4096 and rank g owns the
rows tile_rows(4096, N, g) of all frames (stack.go:142-152 split by pixel
range).  The only exchange is the sum of the two clip counters per pass
(stack.go:193-198): a 16-byte RCCL all-reduce on the device -- the counters are
copied device-to-device behind the pass and reduced on the handle's own stream,
no host round trip per pass.  --weak keeps the tile per GPU fixed instead.

Prints ONE JSON line on rank 0 (contract in the task description) including
  roofline     algorithmic HBM bytes of one launch (4*tile_pixels*(N+1)) / the
               dominant kernel's mean duration over the TIMED passes (HIP events
               on the stream the kernel runs on, read back after the timed region)
  cpu_baseline the CPU oracle (C restatement of the Go reference) on a bounded
               strip of the same stack: 1 warm-up + median of 3, N separately
               allocated host frames, all host threads.
  also         the other stack depths of the north star (sigma clipping, 32 and 512 frames) and, on one GPU,
               the remaining BASELINE.json configurations (C3 tile with its goal-seek, C4, C5), winsorized and
               weighted sigma clipping at 128 frames and what the reference's auto mode picks (winsorized 16 / 24
               frames, linear fit 32 frames): same protocol, each with the dominant kernel's fraction, the pass's
               fraction, the measured HBM traffic where the workload was profiled, and a parity flag (a strip of the
               same stack through the C ABI against the oracle).
  fresh_handle what ONE OpStack.Apply pays on a new handle (stack.go:131-138: the drop-in creates a handle per Apply):
               create, first pass, destroy -- wall-clock milliseconds; with the grid hints the library carries over from
               the last handle of the geometry, and ("without_inherited_hints") without them; a winsorized and a weighted
               pass on the same frames as well (their scratch goes through the same buffer cache).
  ranks        (--gpus N) every rank's wall clock per step, pass and dominant-kernel GPU time, and their min / max.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
MODE_NAMES = {0: "median", 1: "mean", 2: "sigma-clip", 3: "winsorized sigma-clip",
              4: "MAD sigma-clip", 5: "linear-fit"}
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 2.0      # wave-instructions/s: 1024 SIMDs, one wave64 VALU instruction per 2 cycles (MI355X_MICROARCH.md), 2.4 GHz
TRAFFIC_FILES = ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=128)
    ap.add_argument("--width", type=int, default=4096)
    ap.add_argument("--height", type=int, default=4096,
                    help="image rows (strong scaling: split over the GPUs; --weak: rows per GPU)")
    ap.add_argument("--mode", type=int, default=2)
    ap.add_argument("--kappa", type=float, default=3.0)
    ap.add_argument("--weak", action="store_true",
                    help="weak scaling: every rank owns a --height rows tile, the image grows with --gpus")
    ap.add_argument("--row0", type=int, default=0,
                    help="1-GPU runs only: first image row of the tile (with --image-height: one GPU's share of a larger image)")
    ap.add_argument("--image-height", type=int, default=0,
                    help="1-GPU runs only: height of the whole image the --height rows tile is cut from")
    ap.add_argument("--cpu-rows", type=int, default=0,
                    help="rows of the stack timed on the CPU (0 = auto, about 3 s per run)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the host-side legs: cpu_baseline and apply_from_host")
    ap.add_argument("--apply", action="store_true", help="run the apply_from_host leg even with --no-cpu")
    ap.add_argument("--no-apply", action="store_true", help="skip the apply_from_host leg (OpStack.Apply from host memory)")
    ap.add_argument("--preheat-steps", type=int, default=64,
                    help="untimed passes in front of the W warm-up steps (the same number on every rank): a box fresh from idle "
                         "runs its first ~30 ms of work at a lower clock and 5 warm-up passes are 9 ms -- measured 1.91 ms per pass "
                         "with --steps 5 --warmup 2, 1.81 with 10 / 3, 1.76 with 20 / 5, 1.74 from 100 passes on; reported in config")
    ap.add_argument("--no-also", action="store_true",
                    help="default workload only: skip the 32- and 512-frame stacks reported under \"also\"")
    ap.add_argument("--weighted", action="store_true",
                    help="per-frame weights in [0.2, 1] (the shape of inverse-noise weights, stack.go:246-253)")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the multi-rank path)")
    ap.add_argument("--force-dist", action="store_true",
                    help="--gpus 1 only: initialise the process group (world size 1) and drive every pass through the "
                         "device-side counter reduction all the same -- the RCCL code path of the scaling run on one GPU")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal on a 1-GPU box: every rank uses device 0 (needs --backend gloo)")
    return ap.parse_args()


def cpu_info():
    """(model name, hardware threads, physical cores) of the host."""
    model, pairs, threads = "unknown CPU", set(), 0
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown CPU":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("processor"):
                threads += 1
            elif line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
                pairs.add((phys, core))
    except OSError:
        pass
    threads = threads or (os.cpu_count() or 1)
    return model, threads, (len(pairs) or threads)


def cpu_baseline(st, args, rows, weights=None, frames_n=None, mode=None, width=None):
    """Times the oracle (oracle/nl_oracle.c, a C restatement of the Go reference: same batching rule stack.go:134-138, a
    pool of worker threads as the reference's NumCPU goroutines) on the first `rows` rows of the resident stack.

    What "all host cores" means is measured, not assumed (round 6): the GPU boxes of this pool show 256 hardware threads
    to os.cpu_count() and to the affinity mask, but run the container under a cgroup CPU quota (cpu.max) of 16 -- 256
    threads on 16 CPUs' worth of time is how round 5 got 2.5 Msamples/s per thread.  tools/cpu_probe.py:host_limits reads
    affinity, quota and NUMA nodes; the headline figure is the best of a thread sweep (1 thread, half the usable threads,
    the usable threads, twice that, every hardware thread -- the reference would start NumCPU = every hardware thread),
    each run long enough to span several quota periods.  Frames are N separate host allocations (as fits.Image.Data is),
    each first-touched by a worker thread (the reference's loader goroutines allocate them, operator.go:73-116); workers
    are pinned to the allowed CPUs in order (a fresh pthread pool is otherwise spread lazily by the scheduler; Go's
    long-lived Ms are spread already).  Extra workloads (frames_n given): one run at the usable threads, parity only."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle
    from tools.cpu_probe import host_limits
    n, w = frames_n or args.frames, width or args.width
    mode = args.mode if mode is None else mode
    model, threads, physical = cpu_info()
    lim = host_limits()
    usable = lim["usable_threads"]
    frames = [None] * n
    with ThreadPoolExecutor(max(1, min(usable, 16))) as ex:           # first touch by worker threads
        futs = []
        for i in range(n):
            tmp = st.download_rows(i, 0, rows)
            futs.append(ex.submit(lambda i=i, tmp=tmp: frames.__setitem__(i, tmp.copy())))
        for f in futs:
            f.result()
    ow = None if mode in (0, 5) else weights          # median / linear fit take no weights (stack.go:158,188)

    def run(c, sub_rows=None):
        fr = frames if sub_rows is None else [f[:sub_rows * w] for f in frames]
        t0 = time.perf_counter()
        rc, res, cl, ch, _ = oracle.stack_apply(mode, fr, ow, args.kappa, args.kappa, 0.0, num_cpu=c)
        assert rc == 0
        return time.perf_counter() - t0, res, cl, ch

    oracle.set_pin_workers(True)
    try:
        if frames_n is not None:
            dt, res, cl, ch = run(usable)
            return {"value": round(rows * w / dt / 1e6, 3), "unit": "Mpixels/s", "cores": usable}, res, (cl, ch)
        # thread sweep; the 1-thread run on a sixteenth of the strip (it only has to give the per-thread rate)
        one_rows = max(1, rows // 16)
        dt1, _, _, _ = run(1, one_rows)
        per_thread = n * one_rows * w / dt1 / 1e6
        sweep = [{"threads": 1, "msamples_per_s": round(per_thread, 1), "s": round(dt1, 3), "rows": one_rows}]
        counts = sorted({max(1, usable // 2), usable, min(threads, 2 * usable), threads} - {1}) or [1]
        best = None
        res = cc = None
        for c in counts:
            run(c)                                                         # warm-up (page cache, thread placement)
            ts = []
            for _ in range(2):
                dt, r_, cl, ch = run(c)
                ts.append(dt)
                res, cc = r_, (cl, ch)
            dt = min(ts)
            sweep.append({"threads": c, "msamples_per_s": round(n * rows * w / dt / 1e6, 1), "s": round(dt, 3), "rows": rows})
            if best is None or dt < best[0]:
                best = (dt, c)
    finally:
        oracle.set_pin_workers(False)
    dt, c_best = best
    limit = None
    if lim.get("cgroup_cpu_quota") and lim["cgroup_cpu_quota"] < threads:
        limit = "cgroup CPU quota %.4g of %d hardware threads (cpu.max = %s)" % (
            lim["cgroup_cpu_quota"], threads, lim.get("cgroup_cpu_max") or "%s/%s" % (lim.get("cgroup_cfs_quota_us"), lim.get("cgroup_cfs_period_us")))
    elif lim.get("affinity") and lim["affinity"] < threads:
        limit = "affinity mask of %d of %d hardware threads" % (lim["affinity"], threads)
    return {"value": round(rows * w / dt / 1e6, 3), "unit": "Mpixels/s", "cores": min(c_best, usable),
            "threads_of_best_run": c_best, "usable_threads": usable,
            "hardware_threads": threads, "physical_cores": physical, "numa_nodes": lim.get("numa_nodes"),
            "host_limit": limit, "host_limits": {k: lim[k] for k in lim if k != "usable_threads"},
            "msamples_per_s_one_thread": round(per_thread, 1),
            "msamples_per_s_best": round(n * rows * w / dt / 1e6, 1),
            "thread_sweep": sweep,
            "kind": "port",
            "sample": "first %d rows x %d px x %d frames of the same synthetic stack (N separate host allocations, "
                      "first-touched by worker threads), %s, C restatement of the Go reference (no Go toolchain), "
                      "thread sweep with pinned workers, per count 1 warm-up + best of 2: %.2f s on %d threads "
                      "(%d usable%s; %d hardware threads, %d physical cores of %s)"
                      % (rows, w, n, MODE_NAMES[mode], dt, c_best, usable,
                         (" -- " + limit) if limit else "", threads, physical, model)}, res, cc


PCIE_GEN5_X16_GIBS = 63.0 * 1e9 / 1024.0 ** 3          # 32 GT/s x 16 lanes, 128b/130b: 63.0 GB/s per direction before packet overhead


def apply_from_host(st, n, w, h, mode, kappa, device, want_result, want_counters):
    """What a Nightlight user waits for: OpStack.Apply (stack.go:115-227) from frames in HOST memory,
    through exactly the calls go/stackhip/stack_hip.go:92-121 makes -- nl_group_create, nl_group_upload_frame x N,
    nl_group_set_weights, nl_group_run (result into a host buffer), nl_group_destroy -- on ONE GPU.  Frames are N pageable,
    separately allocated host arrays (as fits.Image.Data is), first-touched by worker threads, holding the resident synthetic
    stack of `st` (so the result can be checked against the resident pass).  Two legs: fp32 frames, and the int16 FITS payload
    path (nl_group_upload_frame_fits: big-endian BITPIX 16 / BZERO 32768 as cameras write it, half the PCIe bytes, decoded on
    the device).  Each leg runs twice (a process's first Apply pays hipHostMalloc of the staging ring and cold buffers);
    the second is reported, the first kept as "first_apply_of_the_process".  Breakdown: create, the upload calls (host staging
    memcpy + DMA issue, overlapped), nl_group_run (tail of the DMAs + pass + 64 MiB result download), destroy; a second
    nl_group_run on the settled group separates pass + download from the upload tail."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from nightlight_amd import StackGroup
    from tools.cpu_probe import host_limits
    usable = host_limits()["usable_threads"]
    gib = 1024.0 ** 3
    frames = [None] * n
    with ThreadPoolExecutor(max(1, min(usable, 16))) as ex:
        futs = []
        for i in range(n):
            tmp = st.download_tile(i)
            futs.append(ex.submit(lambda i=i, tmp=tmp: frames.__setitem__(i, tmp.copy())))
        for f in futs:
            f.result()
        # int16 payloads: 8 distinct quantised frames, copied into N separate allocations
        def payload(i):
            q = np.clip(np.rint(np.nan_to_num(frames[i], nan=0.0)), 0, 65535).astype(np.int32) - 32768
            return np.frombuffer(q.astype(">i2").tobytes(), np.uint8)
        distinct = list(ex.map(payload, range(min(8, n))))
        raws = list(ex.map(lambda i: distinct[i % len(distinct)].copy(), range(n)))

    def one(kind):
        t0 = time.perf_counter()
        g = StackGroup(n, w, h, devices=[device])
        t1 = time.perf_counter()
        for i in range(n):
            if kind == "fp32":
                g.upload_frame(i, frames[i])
            else:
                g.upload_frame_fits(i, raws[i], 16, 1.0, 32768.0)
        g.set_weights(None)
        t2 = time.perf_counter()
        out, cl, ch = g.run(mode, kappa, kappa, 0.0)
        t3 = time.perf_counter()
        out2, _, _ = g.run(mode, kappa, kappa, 0.0)              # settled group: pass + download only
        t4 = time.perf_counter()
        pass_ms = g.tile(0).pass_times(0)[0]
        g.close()
        t5 = time.perf_counter()
        nbytes = n * w * h * (4 if kind == "fp32" else 2)
        resident_ms = (t4 - t3) * 1e3
        run_ms = (t3 - t2) * 1e3
        upload_s = (t2 - t1) + max(0.0, run_ms - resident_ms) / 1e3
        wall_ms = ((t3 - t0) + (t5 - t4)) * 1e3                    # create + uploads + run + destroy (the extra run excluded)
        r = {"wall_ms": round(wall_ms, 2), "ms_create": round((t1 - t0) * 1e3, 3),
             "ms_upload_calls": round((t2 - t1) * 1e3, 2), "ms_run": round(run_ms, 2),
             "ms_run_on_settled_group": round(resident_ms, 2), "ms_pass_device": round(pass_ms, 3),
             "ms_result_download": round(resident_ms - pass_ms, 2), "ms_upload_tail_in_run": round(max(0.0, run_ms - resident_ms), 2),
             "ms_destroy": round((t5 - t4) * 1e3, 3),
             "host_bytes": nbytes, "upload_gib_s": round(nbytes / gib / upload_s, 2),
             "upload_frac_of_pcie_gen5_x16": round(nbytes / gib / upload_s / PCIE_GEN5_X16_GIBS, 3),
             "mpixels_per_s_end_to_end": round(w * h / (wall_ms * 1e-3) / 1e6, 2),
             "share": {"upload": round(upload_s * 1e3 / wall_ms, 3), "pass": round(pass_ms / wall_ms, 4),
                       "download": round((resident_ms - pass_ms) / wall_ms, 4),
                       "create_destroy": round(((t1 - t0) + (t5 - t4)) * 1e3 / wall_ms, 4)},
             "clip_counters": [int(cl), int(ch)]}
        return r, out

    res = {"note": "OpStack.Apply from host memory on 1 GPU: nl_group_create / nl_group_upload_frame x %d / nl_group_run / "
                   "nl_group_destroy as go/stackhip/stack_hip.go calls them; %d pageable, separately allocated frames of "
                   "%d x %d; pcie peak = Gen5 x16 %.1f GiB/s" % (n, n, h, w, PCIE_GEN5_X16_GIBS)}
    for kind in ("fp32", "fits_int16"):
        first, _ = one(kind)
        second, out = one(kind)
        second["first_apply_of_the_process"] = {k: first[k] for k in ("wall_ms", "ms_create", "ms_upload_calls", "ms_run", "ms_destroy", "upload_gib_s")}
        if kind == "fp32":
            second["result_equals_resident_pass"] = bool(np.array_equal(out, want_result, equal_nan=True))
            second["counters_equal_resident_pass"] = bool(tuple(second["clip_counters"]) == tuple(want_counters))
        res[kind] = second
    return res


def measured_traffic(kernel, frames, width, rows, mode, want_issue=False):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this workload
    (profiles/rNN_traffic.json: FETCH_SIZE / WRITE_SIZE with the guide's gfx950
    corrections; newest round first), or (None, None) when this workload -- this kernel at
    this geometry -- was not profiled.  want_issue: the entry itself (its SQ_INSTS_VALU / SQ_INSTS_SALU per launch feed
    issue_roofline)."""
    for name in TRAFFIC_FILES:
        try:
            doc = json.load(open(os.path.join(ROOT, "profiles", name)))
        except Exception:
            continue
        for e in doc.get("entries", []):
            if (e["kernel"] == kernel and e["frames"] == frames and e["width"] == width
                    and e["rows"] == rows and e["mode"] == mode and e.get("traffic_bytes")):
                src = "profiles/%s (separate rocprofv3 --pmc passes of this workload)" % name
                if want_issue:
                    return e, src
                return e["traffic_bytes"], src
    return None, None


def issue_roofline(kernel, frames, width, rows, mode, kernel_ms):
    """Issue-rate roofline of the dominant kernel: vector + scalar wave-instructions per launch from the
    committed PMC pass (SQ_INSTS_VALU, SQ_INSTS_SALU) / the kernel's live-measured duration, against one wave64 VALU
    instruction per SIMD every 2 cycles at 2.4 GHz.  'valu_frac' is what the HBM roofline cannot say about a kernel that
    moves few bytes per instruction: how close the SIMDs are to issuing a vector instruction whenever they could.  None
    when the workload's instruction counts are not on record."""
    e, src = measured_traffic(kernel, frames, width, rows, mode, want_issue=True)
    if not e or not e.get("insts_valu"):
        return None
    valu, salu = float(e["insts_valu"]), float(e.get("insts_salu") or 0.0)
    rate = valu / (kernel_ms * 1e-3)
    return {"bound": "valu_issue", "achieved": round(rate / 1e9, 2), "peak": round(VALU_ISSUE_PEAK / 1e9, 2),
            "unit": "G wave-instructions/s", "valu_frac": round(rate / VALU_ISSUE_PEAK, 4),
            "insts_valu_per_launch": valu, "insts_salu_per_launch": salu,
            "cycles_per_valu_instruction_and_simd": round(VALU_ISSUE_PEAK * 2.0 / rate, 3), "source": src}


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks under torch.distributed.run."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world

    import numpy as np
    import torch
    from nightlight_amd import StackHandle
    from nightlight_amd.dist import tile_rows

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    if world > 1 and not args.share_device and torch.cuda.device_count() < world:
        raise SystemExit("bench.py --gpus %d: only %d HIP device(s) visible" % (world, torch.cuda.device_count()))
    device = 0 if args.share_device else local_rank
    torch.cuda.set_device(device)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        kw = {}
        if "MASTER_ADDR" not in os.environ:              # --force-dist without a launcher: a one-rank group of its own
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            kw = dict(init_method="tcp://127.0.0.1:%d" % sk.getsockname()[1], rank=0, world_size=1)
            sk.close()
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device), **kw)
        else:
            dist.init_process_group(backend=args.backend, **kw)
    on_device = args.backend == "nccl"

    n, w = args.frames, args.width
    if args.weak:
        image_rows = args.height * world
        row0, rows = rank * args.height, args.height
    elif world == 1 and args.image_height:
        image_rows, row0, rows = args.image_height, args.row0, args.height
    else:
        image_rows = args.height
        row0, rows = tile_rows(image_rows, world, rank)

    # global clip totals (the log line of stack.go:214-218 needs them on every pass)
    totals = torch.zeros(2, dtype=torch.int64, device="cuda" if on_device else "cpu")

    def time_stack(frames, mode, steps, warmup, weights=None, geo=None):
        """One workload on this rank's row tile: `warmup` untimed passes, then exactly `steps` timed ones
        bracketed by barrier + synchronize.  Returns the handle (still open) and the timings.
        geo = (width, image_rows, row0, rows) of another geometry than the headline's (1-GPU extras)."""
        gw, gh, gr0, grows = geo or (w, image_rows, row0, rows)
        st = StackHandle(frames, gw, gh, row0=gr0, rows=grows, device=device)
        st.fill_synthetic()
        if os.environ.get("NL_DEV_FLAGS"):
            st.set_dev_flags(int(os.environ["NL_DEV_FLAGS"]))
        if weights is not None:
            st.set_weights(weights)
        stream = torch.cuda.ExternalStream(st.stream_ptr, device=device) if (dist is not None and on_device) else None
        # Device-side reduction without a copy kernel: pass i leaves its counters in ring[i % 3] (the caller's buffer,
        # nl_stack_set_counters_buffer), the all-reduce of pass i runs in place on its first two words while pass i + 1
        # has the device, and a buffer is handed out again only behind the collective that used it last.
        ring = torch.zeros((3, 4), dtype=torch.int64, device="cuda") if (dist is not None and on_device) else None
        works = [None, None, None]
        seq = [0]
        # The collectives are issued from a stream of their own that waits for pass i through the library's fence-free event
        # (nl_stack_order_stream_after): torch's bookkeeping for an asynchronous collective -- an event on the issuing stream,
        # a wait when the buffer is reused -- then lands on that stream and not on the one the passes run on (0.2634 -> 0.2605 ms
        # per step of a 512-row share, world-size-1 group; tools/host_step_probe.py: the host enqueues a step in 49 us).
        comm = torch.cuda.Stream(device=device) if (dist is not None and on_device) else None

        def step():
            if dist is not None and on_device:
                k = seq[0] % 3
                if works[k] is not None:
                    # (three passes old: long done -- a host-side query then; a wait on the pass's stream only if it really is
                    # still running: it orders the reuse of the buffer)
                    if not works[k].is_completed():
                        with torch.cuda.stream(stream):
                            works[k].wait()
                    works[k] = None
                st.set_counters_buffer(ring[k].data_ptr())
            st.run_async(mode, args.kappa, args.kappa, 0.0)
            if dist is None:
                return
            if on_device:
                st.order_stream_after(comm.cuda_stream)      # comm waits for this pass; the pass's stream waits for nothing
                with torch.cuda.stream(comm):
                    works[seq[0] % 3] = dist.all_reduce(ring[seq[0] % 3][:2], async_op=True)
                seq[0] += 1
            else:                                   # gloo rehearsal: counters through the host
                cl, ch = st.finish()
                totals.copy_(torch.tensor([cl, ch], dtype=torch.int64))
                dist.all_reduce(totals)

        def fence():
            if stream is not None:
                with torch.cuda.stream(comm):
                    for k in range(3):
                        if works[k] is not None:
                            works[k].wait()
                            works[k] = None
                comm.synchronize()
            st.finish()
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
                torch.cuda.synchronize()

        for _ in range(args.preheat_steps):                  # device clock ramp, untimed (see --preheat-steps)
            step()
        fence()
        for _ in range(warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        dt = time.perf_counter() - t0

        # HIP-event times of the timed passes, read back from the handle's ring after the fact
        timed = min(steps, 64)
        times = [st.pass_times(b) for b in range(timed)]
        pass_ms = float(np.mean([t[0] for t in times]))
        k_ms = float(np.mean([t[1] for t in times]))
        # one more pass, synchronous as OpStack.Apply runs it (enqueue, wait, read the counters back)
        if ring is not None:
            st.set_counters_buffer(None)                     # (back to the handle's own counters)
        t1 = time.perf_counter()
        st.run_async(mode, args.kappa, args.kappa, 0.0)
        cl, ch = st.finish()
        sync_ms = (time.perf_counter() - t1) * 1e3
        ranks = None
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda" if on_device else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            # every rank's own clock and GPU times, so that a straggler shows in the scaling record
            mine = torch.tensor([dt, pass_ms, k_ms], dtype=torch.float64, device="cuda" if on_device else "cpu")
            every = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
            dist.all_gather(every, mine)
            rows_ = [[float(x) for x in e.tolist()] for e in every]
            ranks = {"wall_ms_per_step": [round(r[0] * 1e3 / steps, 4) for r in rows_],
                     "pass_ms": [round(r[1], 4) for r in rows_], "kernel_ms": [round(r[2], 4) for r in rows_],
                     "pass_ms_min": round(min(r[1] for r in rows_), 4), "pass_ms_max": round(max(r[1] for r in rows_), 4),
                     "kernel_ms_min": round(min(r[2] for r in rows_), 4), "kernel_ms_max": round(max(r[2] for r in rows_), 4)}
            dt = float(t.item())
            if on_device:
                st.copy_counters_async(totals.data_ptr())
                torch.cuda.synchronize()
            else:
                totals.copy_(torch.tensor([cl, ch], dtype=torch.int64))
            dist.all_reduce(totals)
            cl, ch = int(totals[0].item()), int(totals[1].item())
        return st, {"dt": dt, "pass_ms": pass_ms, "k_ms": k_ms, "timed": timed, "cl": cl, "ch": ch, "sync_ms": sync_ms, "ranks": ranks}

    def strip_parity(frames, mode, geo, weights_, strip_rows, res, cc):
        """The first strip_rows rows of the tile through the C ABI against the oracle's result `res` / counters `cc`
        for the same rows: counters must be equal, values within the north star's 1e-5 (bit-exact for every kernel but
        the register-resident ones)."""
        gw, gh, gr0, _ = geo
        with StackHandle(frames, gw, gh, row0=gr0, rows=strip_rows, device=device) as strip:
            strip.fill_synthetic()
            strip.set_weights(weights_)
            strip.run_async(mode, args.kappa, args.kappa, 0.0)
            gl, gh_ = strip.finish()
            got = strip.download_rows(-1, 0, strip_rows)
        ok = ~np.isnan(res) & (res != 0)
        same_nan = bool(np.array_equal(np.isnan(got), np.isnan(res)))
        rel = float(np.max(np.abs(got[ok].astype(np.float64) - res[ok]) / np.abs(res[ok]))) if ok.any() else 0.0
        counted = mode >= 2
        return {"clip_counters_equal": bool((gl, gh_) == cc) if counted else True,
                "clip_counters": [int(gl), int(gh_)],
                "max_rel_err": rel,
                "bit_exact": bool(np.array_equal(got, res, equal_nan=True)),
                "within_1e-5": bool(same_nan and rel <= 1e-5)}

    def fresh_handle_cost(st, frames, mode, weights_, no_hints=False):
        """What ONE Apply pays on a new handle (the drop-in creates a handle per Apply, go/stackhip/stack_hip.go): create
        (device buffers, streams, events), the first pass and destroy; the frames are the resident ones of `st` (attached,
        not uploaded again).  The first pass of a handle has no list lengths of its own to size its replay grids with; the
        library remembers those of the last handle of the same geometry in the process (hints_load / hints_store,
        nlstack_api.hip), so "warm" numbers are WITH those inherited hints and the fused protocol they allow;
        no_hints=True (developer switch 512) times a first pass without them, as the first Apply of a process runs."""
        def cycle():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            h2 = StackHandle(frames, w, image_rows, row0=row0, rows=rows, device=device)
            t1 = time.perf_counter()
            h2.attach_device_frames(st.frames_device_ptr(), st.frame_stride())
            h2.set_weights(weights_)
            if no_hints:
                h2.set_dev_flags(512)
            t2 = time.perf_counter()
            h2.run_async(mode, args.kappa, args.kappa, 0.0)
            cl2, ch2 = h2.finish()
            t3 = time.perf_counter()
            h2.run_async(mode, args.kappa, args.kappa, 0.0)
            h2.finish()
            t4 = time.perf_counter()
            h2.attach_device_frames(None)
            h2.close()
            t5 = time.perf_counter()
            return (t1 - t0) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, int(cl2), int(ch2)
        cold = cycle()          # nothing parked yet: hipMalloc / hipFree of every buffer
        warm = cycle()          # the buffers the first cycle's destroy parked (nl_release_cached_memory)
        return {"ms_create": round(warm[0], 3), "ms_first_pass_fresh_handle": round(warm[1], 3),
                "ms_second_pass_synchronous": round(warm[2], 3), "ms_destroy": round(warm[3], 3),
                "ms_create_destroy": round(warm[0] + warm[3], 3),
                "first_handle_of_the_process": {"ms_create": round(cold[0], 3), "ms_first_pass_fresh_handle": round(cold[1], 3),
                                                "ms_destroy": round(cold[3], 3), "ms_create_destroy": round(cold[0] + cold[3], 3)},
                "clip_counters": [warm[4], warm[5]],
                "note": "one handle per OpStack.Apply: create + first pass + destroy on frames already resident "
                        "(uploads excluded); the library parks the large buffers of a destroyed handle for the next one "
                        "of the same geometry and hands its list lengths on as grid hints"
                        + (" -- here the first pass ran WITHOUT inherited hints (developer switch 512)" if no_hints else
                           " -- the first pass ran WITH the hints the previous handle left") +
                        "; the timed steps above re-use one handle"}

    default_workload_early = (args.frames == 128 and args.mode == 2 and args.width == 4096 and args.height == 4096 and
                              not args.weak and not args.weighted and not args.image_height)
    weights = None
    if args.weighted:
        weights = np.array([0.2 + 0.8 * ((k * 37) % 101) / 100.0 for k in range(n)], np.float32)
    st, tm = time_stack(n, args.mode, args.steps, args.warmup, weights)
    dt, pass_ms, k_ms, timed, cl, ch = tm["dt"], tm["pass_ms"], tm["k_ms"], tm["timed"], tm["cl"], tm["ch"]

    if rank == 0:
        pixels_per_step = image_rows * w
        ms_per_step = dt * 1e3 / args.steps
        value = pixels_per_step * args.steps / dt / 1e6
        alg_bytes = 4.0 * rows * w * (n + 1)          # per launch (rank 0's tile), SURVEY 8d
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        traffic, traffic_source = measured_traffic(st.last_kernel_name, n, w, rows, args.mode)
        if args.weak:
            workload = "%d x %dx%d fp32 frames per GPU (weak scaling), %s%s kappa=%g, frames resident in HBM" % (
                n, rows, w, MODE_NAMES[args.mode], " (weighted)" if args.weighted else "", args.kappa)
        else:
            workload = "%d x %dx%d fp32 frames, %s%s kappa=%g, rows split over %d GPU(s), frames resident in HBM" % (
                n, image_rows, w, MODE_NAMES[args.mode], " (weighted)" if args.weighted else "", args.kappa, world)
            if world == 1 and rows != image_rows:
                workload = "rows [%d,%d) of " % (row0, row0 + rows) + workload
        out = {
            "metric": "stacked Mpixels/sec (%s, %dx%dx%d fp32)" % (MODE_NAMES[args.mode], n, image_rows, w),
            "value": round(value, 3), "unit": "Mpixels/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak" if args.weak else "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload,
                       "frames": n, "width": w, "image_rows": image_rows, "rows_per_gpu": rows, "mode": args.mode, "preheat_steps": args.preheat_steps,
                       "sharding": "row tiles, %d rank(s); per pass one all-reduce of 2 int64 clip counters%s"
                                   % (world, " on the device (RCCL, the pass's own stream)" if (dist is not None and on_device)
                                      else ""),
                       "clip_low": cl, "clip_high": ch},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": st.last_kernel_name,
                         "kernel_ms": round(k_ms, 4), "algorithmic_bytes": alg_bytes,
                         "timed_passes_averaged": timed,
                         "pass_ms": round(pass_ms, 4),
                         "pass_frac": round(alg_bytes / (pass_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                         "pixels_redone_by_exact_kernel": st.last_fallback_pixels,
                         "issue": issue_roofline(st.last_kernel_name, n, w, rows, args.mode, k_ms)},
            # ms_per_step times passes queued back to back (one host sync at the end of the timed region);
            # this is one pass run the way OpStack.Apply runs it: enqueue, wait, read the counters back
            "ms_per_step_synchronous": round(tm["sync_ms"], 4),
        }
        if tm["ranks"] is not None:
            out["ranks"] = tm["ranks"]            # per rank: wall clock per step, pass and dominant-kernel GPU time (HIP events)
        if world == 1 and not args.no_cpu:
            cpu_rows = args.cpu_rows
            if cpu_rows <= 0:
                # about 2 s per run at the usable threads (several cgroup quota periods): the oracle does ~50e6 samples/s
                # per thread (sigma clip); 13 runs of the sweep then stay within ~30 s
                from tools.cpu_probe import host_limits
                cpu_rows = max(8, min(rows, int(host_limits()["usable_threads"] * 2.0 * 50.0e6 / (n * w))))
            cpu_rows = min(cpu_rows, rows)
            base, res, cc = cpu_baseline(st, args, cpu_rows, weights)
            # parity in the same run: the same strip through the C ABI vs the oracle
            base["parity_with_gpu"] = strip_parity(n, args.mode, (w, image_rows, row0, rows), weights, cpu_rows, res, cc)
            out["cpu_baseline"] = base
        if world == 1 and dist is None:
            out["fresh_handle"] = fresh_handle_cost(st, n, args.mode, weights)
            out["fresh_handle"]["without_inherited_hints"] = {
                k: v for k, v in fresh_handle_cost(st, n, args.mode, weights, no_hints=True).items()
                if k in ("ms_create", "ms_first_pass_fresh_handle", "ms_second_pass_synchronous", "ms_destroy", "note")}
            if default_workload_early:
                # the scratch of the other pass protocols goes through the same buffer cache: a winsorized stack (the
                # cascade's lists) and a weighted one (the decision pass's thresholds) on the same resident frames
                w_ = np.array([0.2 + 0.8 * ((k * 37) % 101) / 100.0 for k in range(n)], np.float32)
                for tag, m_, wt_ in (("winsorized", 3, None), ("weighted_sigma", 2, w_)):
                    c = fresh_handle_cost(st, n, m_, wt_)
                    out["fresh_handle"][tag] = {k: c[k] for k in ("ms_create", "ms_first_pass_fresh_handle", "ms_destroy", "ms_create_destroy")}

    if rank == 0 and world == 1 and dist is None and weights is None and (
            args.apply or (default_workload_early and not (args.no_apply or args.no_cpu))):     # (--no-cpu skips both host-side legs; --apply: any geometry)
        want = st.download_rows(-1, 0, rows)          # result of the last resident pass (same frames, same kappa)
        out["apply_from_host"] = apply_from_host(st, n, w, image_rows, args.mode, args.kappa, device, want, (cl, ch))
    st.close()
    # The other stack depths the north star names (4096 x 4096 x {32, 512} fp32, sigma clipping), same protocol
    # and the same row tiles, reported beside the headline (which stays what `value` is).
    default_workload = (args.frames == 128 and args.mode == 2 and args.width == 4096 and args.height == 4096 and
                        not args.weak and not args.weighted and not args.image_height)
    if default_workload and not args.no_also:
        w128 = np.array([0.2 + 0.8 * ((k * 37) % 101) / 100.0 for k in range(128)], np.float32)
        # (tag, frames, mode, weights, geometry (width, image rows, first row, rows) or None = the headline's tile, goal-seek)
        extras = [("sigma32", 32, 2, None, None, False), ("sigma512", 512, 2, None, None, False)]
        if world == 1 and dist is None:
            # the remaining BASELINE.json configurations on one GPU -- C3 as one of its 8 row tiles (rows 1536 ... 2047 of
            # 512 x 4096 x 4096, winsorized clipping, and the goal-seek of stackfindsigma.go:48-98 on that tile), C4 (linear
            # fit; the reference's StackLinearFit takes no weights, stack.go:188-189), C5 -- and the two modes whose kernels
            # differ most from the headline's: winsorized and weighted sigma clipping at 128 frames
            extras += [("C3 tile (1 of 8 GPUs)", 512, 3, None, (w, 4096, 1536, 512), True),
                       ("C4", 128, 5, None, None, False),
                       ("C5", 64, 0, None, (6000, 4000, 0, 4000), False),
                       ("winsor128", 128, 3, None, None, False),
                       ("weighted sigma128", 128, 2, w128, None, False),
                       # what the reference's auto mode (-stMode 6, stack.go:45-55) picks for 15 ... 24 frames (winsorized
                       # clipping) and from 25 frames on (linear fit)
                       ("winsor16", 16, 3, None, None, False),
                       ("winsor24", 24, 3, None, None, False),
                       ("linfit32", 32, 5, None, None, False)]
        also = []
        for tag, frames, mode, wts, geo, goal_seek in extras:
            st2, t2 = time_stack(frames, mode, args.steps, args.warmup, wts, geo)
            if rank == 0:
                gw, gh, gr0, grows = geo or (w, image_rows, row0, rows)
                alg = 4.0 * grows * gw * (frames + 1)
                e = {"tag": tag,
                     "workload": "%s%d x %dx%d fp32 frames, %s%s kappa=%g, rows split over %d GPU(s)" % (
                         ("rows [%d,%d) of " % (gr0, gr0 + grows)) if geo and grows != gh else "", frames, gh, gw,
                         MODE_NAMES[mode], " (weighted)" if wts is not None else "", args.kappa, world),
                     "value": round((grows if geo else image_rows) * gw * args.steps / t2["dt"] / 1e6, 3), "unit": "Mpixels/s",
                     "ms_per_step": round(t2["dt"] * 1e3 / args.steps, 4),
                     "ms_per_step_synchronous": round(t2["sync_ms"], 4),
                     "kernel": st2.last_kernel_name, "kernel_ms": round(t2["k_ms"], 4), "pass_ms": round(t2["pass_ms"], 4),
                     "frac": round(alg / (t2["k_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                     "pass_frac": round(alg / (t2["pass_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                     "algorithmic_bytes": alg, "clip_low": t2["cl"], "clip_high": t2["ch"],
                     "pixels_redone_by_exact_kernel": st2.last_fallback_pixels}
                e["traffic"], e["traffic_source"] = measured_traffic(st2.last_kernel_name, frames, gw, grows, mode)
                e["roofline_issue"] = issue_roofline(st2.last_kernel_name, frames, gw, grows, mode, t2["k_ms"])
                if t2["ranks"] is not None:
                    e["ranks"] = t2["ranks"]
                if mode == 5:
                    # (the linear fit is a cascade of stages: kernel_ms is its first stage, the pass is what counts)
                    e["note"] = "kernel_ms = first stage of the cascade (stack_linfit.hip); pass_ms covers all stages"
                if mode == 3 and 12 <= frames <= 96:
                    e["note"] = ("kernel_ms = first stage of the winsorization cascade (stack_fast_sigma_impl.hpp); its "
                                 "continuation stages, the generic pass and the replays are in pass_ms")
                if goal_seek:
                    t0 = time.perf_counter()
                    _, gcl, gch, gsl, gsh, gpasses = st2.find_sigmas(mode, 0.5, 0.5, fetch=False)
                    e["goal_seek"] = {"target_percent": [0.5, 0.5], "passes": gpasses, "sigma_low": gsl, "sigma_high": gsh,
                                      "clip_low": gcl, "clip_high": gch,
                                      "total_ms": round((time.perf_counter() - t0) * 1e3, 3)}
                    # the same bisection step by step: what each step's pass costs inside the sequence and what a pass at the
                    # SAME sigmas costs in steady state (round 6: the two agree -- a goal-seek step carries no overhead of its
                    # own; the total differs from passes x the kappa-3 pass because the steps are not kappa-3 passes: the
                    # third one clips 5 % of the samples)
                    from tools.goalseek_probe import step_table
                    gsteps, _ = step_table(st2, mode)
                    e["goal_seek"]["steps"] = gsteps
                    e["goal_seek"]["sum_pass_ms_in_sequence"] = round(sum(x["pass_ms"] for x in gsteps), 3)
                    e["goal_seek"]["sum_steady_pass_ms_at_the_same_sigmas"] = round(sum(x["steady_pass_ms"] for x in gsteps), 3)
                if world == 1 and not args.no_cpu:
                    # parity flag: the first rows of the tile through the C ABI against the oracle (one run, all host threads)
                    prow = min(16, grows)
                    _, res2, cc2 = cpu_baseline(st2, args, prow, wts, frames_n=frames, mode=mode, width=gw)
                    e["parity_with_oracle"] = dict(strip_parity(frames, mode, (gw, gh, gr0, grows), wts, prow, res2, cc2),
                                                   rows_checked=prow)
                also.append(e)
            st2.close()
        if rank == 0:
            out["also"] = also
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
