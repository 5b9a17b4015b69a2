#!/usr/bin/env python3
"""bench.py -- stacked Mpixels/s of the per-pixel stacking hot path on MI355X.

One "step" = one full stack pass (OpStack.Apply's numeric core,
internal/ops/stack/stack.go:142-218) over a synthetic sub-exposure stack that
is already resident in HBM.  Default workload = BASELINE.json configs[1]:
128 x 4096 x 4096 fp32 frames, sigma-clipped mean, kappa = 3, one GPU.

With --gpus N (launched by torch.distributed.run, one rank per GPU) every rank
owns one row tile of the same size (weak scaling: the image grows with N); the
only exchange is the all-reduce of the two clip counters per pass (RCCL).

Prints ONE JSON line on rank 0 (contract in the task description) including
  roofline     algorithmic HBM bytes (4*P*(N+1)) / HIP-event kernel time
  cpu_baseline the CPU oracle (C restatement of the Go reference) timed on a
               bounded strip of the same stack with all host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
MODE_NAMES = {0: "median", 1: "mean", 2: "sigma-clip", 3: "winsorized sigma-clip",
              4: "MAD sigma-clip", 5: "linear-fit"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=128)
    ap.add_argument("--width", type=int, default=4096)
    ap.add_argument("--height", type=int, default=4096, help="rows per GPU")
    ap.add_argument("--mode", type=int, default=2)
    ap.add_argument("--kappa", type=float, default=3.0)
    ap.add_argument("--cpu-rows", type=int, default=0,
                    help="rows of the stack timed on the CPU (0 = auto, about 10-30 s)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--weighted", action="store_true",
                    help="per-frame weights in [0.2, 1] (the shape of inverse-noise weights, stack.go:246-253)")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the multi-rank path)")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal on a 1-GPU box: every rank uses device 0 (needs --backend gloo)")
    return ap.parse_args()


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(st, args, rows, weights=None):
    """Times the oracle (oracle/nl_oracle.c, a C restatement of the Go reference:
    same batching rule, one worker per host core) on the first `rows` rows."""
    import numpy as np
    from oracle import oracle
    n, w = args.frames, args.width
    frames = np.empty((n, rows * w), np.float32)
    for i in range(n):
        frames[i] = st.download_tile(i)[: rows * w]
    cores = os.cpu_count() or 1
    t0 = time.perf_counter()
    ow = None if args.mode in (0, 5) else weights          # median / linear fit take no weights (stack.go:158,188)
    rc, res, cl, ch, _ = oracle.stack_apply(args.mode, frames, ow, args.kappa, args.kappa,
                                            0.0, num_cpu=cores)
    dt = time.perf_counter() - t0
    assert rc == 0
    return {"value": round(rows * w / dt / 1e6, 3), "unit": "Mpixels/s", "cores": cores,
            "kind": "port",
            "sample": "first %d rows x %d px x %d frames of the same synthetic stack, %s, "
                      "C restatement of the Go reference (no Go toolchain), %.1f s on %d threads of %s"
                      % (rows, w, n, MODE_NAMES[args.mode], dt, cores, cpu_model())}, res, (cl, ch)


def measured_traffic(kernel, args):
    """HBM bytes per launch from the committed rocprofv3 PMC passes
    (profiles/r01_traffic.json: 2*FETCH_SIZE + WRITE_SIZE, gfx950 correction),
    or None when this workload was not profiled."""
    try:
        doc = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
    except Exception:
        return None
    for e in doc.get("entries", []):
        if (e["kernel"] == kernel and e["frames"] == args.frames and e["width"] == args.width
                and e["rows"] == args.height and e["mode"] == args.mode):
            return e["traffic_bytes"]
    return None


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world

    import numpy as np
    import torch
    from nightlight_amd import StackHandle

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    device = 0 if args.share_device else local_rank
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend=args.backend)
    comm_device = "cuda" if args.backend == "nccl" else "cpu"

    n, w, rows = args.frames, args.width, args.height
    total_rows = rows * world
    st = StackHandle(n, w, total_rows, row0=rank * rows, rows=rows, device=device)
    st.fill_synthetic()
    weights = None
    if args.weighted:
        weights = np.array([0.2 + 0.8 * ((k * 37) % 101) / 100.0 for k in range(n)], np.float32)
        st.set_weights(weights)
    counters = torch.zeros(2, dtype=torch.int64, device=comm_device)

    def step():
        st.run_async(args.mode, args.kappa, args.kappa, 0.0)
        cl, ch = st.finish()
        if dist is not None:     # global clip totals, as the log line of stack.go:214-218 needs
            counters.copy_(torch.tensor([cl, ch], dtype=torch.int64))
            dist.all_reduce(counters)
        return cl, ch

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    kernel_ms, pass_ms = [], []
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        cl, ch = step()
        kernel_ms.append(st.last_dominant_kernel_ms)     # HIP events around the dominant kernel
        pass_ms.append(st.last_kernel_ms)                # ... and around every kernel of the pass
    fence()
    dt = time.perf_counter() - t0

    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=comm_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        pixels_per_step = rows * w * world
        ms_per_step = dt * 1e3 / args.steps
        value = pixels_per_step * args.steps / dt / 1e6
        k_ms = float(np.mean(kernel_ms))
        alg_bytes = 4.0 * rows * w * (n + 1)          # per launch (one tile), SURVEY 8d
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        out = {
            "metric": "stacked Mpixels/sec (%s, %dx%dx%d fp32)" % (MODE_NAMES[args.mode], n, rows, w),
            "value": round(value, 3), "unit": "Mpixels/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%d x %dx%d fp32 frames per GPU, %s%s kappa=%g, frames resident in HBM"
                                   % (n, rows, w, MODE_NAMES[args.mode], " (weighted)" if args.weighted else "",
                                      args.kappa),
                       "frames": n, "width": w, "rows_per_gpu": rows, "mode": args.mode,
                       "sharding": "row tiles, %d rank(s); all-reduce of 2 int64 clip counters per pass" % world,
                       "clip_low": int(counters[0].item()) if dist is not None else cl,
                       "clip_high": int(counters[1].item()) if dist is not None else ch},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": measured_traffic(st.last_kernel_name, args),
                         "kernel": st.last_kernel_name,
                         "kernel_ms": round(k_ms, 4), "algorithmic_bytes": alg_bytes,
                         "pass_ms": round(float(np.mean(pass_ms)), 4),
                         "pass_frac": round(alg_bytes / (float(np.mean(pass_ms)) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                         "pixels_redone_by_exact_kernel": st.last_fallback_pixels},
        }
        if world == 1 and not args.no_cpu:
            cpu_rows = args.cpu_rows
            if cpu_rows <= 0:
                # about 10-30 s of CPU time: the oracle does ~5e6 samples/s per core
                # (sigma clip), so give every core ~12 s of work, capped at the tile
                cores = os.cpu_count() or 1
                cpu_rows = max(8, min(rows, int(cores * 12 * 5.0e6 / (n * w))))
            cpu_rows = min(cpu_rows, rows)
            base, res, cc = cpu_baseline(st, args, cpu_rows, weights)
            # parity in the same run: the same strip through the C ABI vs the oracle.
            # Clip counters must be equal; values within the north star's 1e-5
            # (bit-exact for every kernel but the register-resident sigma one).
            with StackHandle(n, w, total_rows, row0=0, rows=cpu_rows, device=device) as strip:
                strip.fill_synthetic()
                strip.set_weights(weights)
                got, gl, gh = strip.run(args.mode, args.kappa, args.kappa, 0.0)
                got = got[: cpu_rows * w]
            ok = ~np.isnan(res) & (res != 0)
            same_nan = bool(np.array_equal(np.isnan(got), np.isnan(res)))
            rel = float(np.max(np.abs(got[ok].astype(np.float64) - res[ok]) / np.abs(res[ok]))) if ok.any() else 0.0
            base["parity_with_gpu"] = {
                "clip_counters_equal": bool((gl, gh) == cc),
                "clip_counters": [int(gl), int(gh)],
                "max_rel_err": rel,
                "bit_exact": bool(np.array_equal(got, res, equal_nan=True)),
                "within_1e-5": bool(same_nan and rel <= 1e-5),
            }
            out["cpu_baseline"] = base
        print(json.dumps(out), flush=True)

    st.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
