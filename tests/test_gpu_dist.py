"""Multi-GPU forms of the path on ONE device (the GPU box has a single MI355X):

* two torch.distributed ranks (gloo, both on device 0) each run the HIP path on
  their own row tile and exchange only the two clip counters; the tiles
  reassemble the oracle's single-image result (stack.go:142-152, 193-198);
* nl_group_*: the single-process fan-out over n tiles (what the Go shim and the
  C++ operator mirror call), tiles all on device 0 here;
* bench.py --gpus 2 launches its own ranks and reports strong scaling.
RCCL itself (backend nccl over xGMI) needs >= 2 GPUs and is exercised by the
driver's scaling run; everything around the collective is the same code."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from util import bits_equal, make_frames

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, N = 160, 90, 20          # 90 rows over 2 / 3 / 4 tiles: uneven splits


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    import nightlight_amd as nl
    from nightlight_amd.dist import ShardedStack
    from util import make_frames as mk
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = mk(N, W, H, seed=7100)

    def make_tile(row0, rows):
        st = nl.StackHandle(N, W, H, row0=row0, rows=rows, device=0)     # both ranks share device 0
        st.upload_frames(frames)
        return st

    sh = ShardedStack(H, make_tile, world=world, rank=rank, device="cpu")
    out = {}
    for mode in (2, 3, 5):
        res, cl, ch = sh.run(mode, 2.5, 2.5)
        out["res%d" % mode] = res
        out["clip%d" % mode] = np.array([cl, ch])
    gs = sh.find_sigmas(2, 1.0, 1.0)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), row0=sh.row0, rows=sh.rows,
             gs_res=gs[0], gs=np.array(gs[1:], np.float64), **out)
    sh.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_gloo_ranks_shard_the_hip_path(tmp_path, oracle, nl):
    import torch.multiprocessing as mp
    world, port = 2, 29600 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    frames = make_frames(N, W, H, seed=7100)
    ranks = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    for mode in (2, 3, 5):
        rc, want, wl, wh, _ = oracle.stack_apply(mode, frames, None, 2.5, 2.5)
        full = np.zeros(W * H, np.float32)
        for d in ranks:
            r0, rows = int(d["row0"]), int(d["rows"])
            full[r0 * W:(r0 + rows) * W] = d["res%d" % mode][r0 * W:(r0 + rows) * W]
            assert tuple(d["clip%d" % mode]) == (wl, wh)          # every rank holds the GLOBAL totals
        assert np.array_equal(np.isnan(full), np.isnan(want))
        ok = ~np.isnan(want)
        assert np.all(np.abs(full[ok].astype(np.float64) - want[ok]) <= 1e-5 * np.abs(want[ok]))
    passes, gres, gcl, gch, gsl, gsh = oracle.find_sigmas_bisect(2, frames, 1.0, 1.0)
    gfull = np.zeros(W * H, np.float32)
    for d in ranks:
        r0, rows = int(d["row0"]), int(d["rows"])
        gfull[r0 * W:(r0 + rows) * W] = d["gs_res"][r0 * W:(r0 + rows) * W]
        assert tuple(d["gs"][:2]) == (gcl, gch) and int(d["gs"][4]) == passes
        assert (np.float32(d["gs"][2]), np.float32(d["gs"][3])) == (gsl, gsh)
    ok = ~np.isnan(gres)
    assert np.all(np.abs(gfull[ok].astype(np.float64) - gres[ok]) <= 1e-5 * np.abs(gres[ok]))


@pytest.mark.parametrize("tiles", [1, 3, 4])
def test_group_fans_one_stack_out_over_tiles(nl, oracle, tiles):
    frames = make_frames(N, W, H, seed=7200 + tiles)
    w = (0.3 + 0.7 * ((np.arange(N) * 13) % 17) / 16.0).astype(np.float32)
    with nl.StackGroup(N, W, H, devices=[0] * tiles) as g:
        assert g.size == tiles
        spans = [g.tile_rows(t) for t in range(tiles)]
        assert spans[0][0] == 0 and sum(r for _, r in spans) == H
        assert all(a[0] + a[1] == b[0] for a, b in zip(spans, spans[1:]))
        g.upload_frames(frames)
        for mode, weights, exact in ((0, None, True), (1, w, True), (2, None, False), (2, w, True), (3, None, False),
                                     (4, None, False), (5, None, True)):
            g.set_weights(weights)
            got, cl, ch = g.run(mode, 2.0, 2.5)
            ow = None if mode in (0, 5) else weights
            rc, want, wl, wh, _ = oracle.stack_apply(mode, frames, ow, 2.0, 2.5)
            assert rc == 0
            if mode >= 2:
                assert (cl, ch) == (wl, wh), (mode, tiles)
            if exact:
                assert bits_equal(got, want), (mode, tiles)
            else:
                ok = ~np.isnan(want)
                assert np.array_equal(np.isnan(got), np.isnan(want))
                assert np.all(np.abs(got[ok].astype(np.float64) - want[ok]) <= 1e-5 * np.abs(want[ok]))
        # goal-seek with host-summed counters takes the oracle's branches
        g.set_weights(None)
        res, cl, ch, sl, sh, passes = g.find_sigmas(2, 1.0, 1.5)
        op, ores, ocl, och, osl, osh = oracle.find_sigmas_bisect(2, frames, 1.0, 1.5)
        assert (cl, ch, np.float32(sl), np.float32(sh), passes) == (ocl, och, osl, osh, op)
        # stack of stacks across the tiles
        g.run(1)
        g.accumulate(3.0, True)
        g.accumulate(2.0, False)
        acc = g.accumulate_finalize(5.0)
        rc, mean, _, _, _ = oracle.stack_apply(1, frames, None)
        a = oracle.stack_incremental(np.zeros_like(mean), mean, 3.0, True)
        a = oracle.stack_incremental(a, mean, 2.0, False)
        assert bits_equal(acc, oracle.stack_incremental_finalize(a, 5.0))


@pytest.mark.parametrize("tiles", [2, 5])
def test_group_finish_on_worker_threads_gives_the_serial_result(nl, oracle, monkeypatch, tiles):
    # nl_group_run finishes its tiles on worker threads when they sit on several devices (the copies of the result rows then
    # run over separate links); forced here on one device, against the serial finish and the oracle
    frames = make_frames(N, W, H, seed=7300 + tiles)
    res = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("NL_GROUP_PARALLEL_FINISH", flag)
        with nl.StackGroup(N, W, H, devices=[0] * tiles) as g:
            g.upload_frames(frames)
            res[flag] = [g.run(mode, 2.0, 2.5) for mode in (1, 2, 3, 5)]
            g.set_weights(np.ones(N, np.float32))
            with pytest.raises(Exception):
                g.run(4)                                     # (a failing pass leaves nothing pending in either flavour)
            g.set_weights(None)
            again = g.run(2, 2.0, 2.5)
            assert again[1:] == res[flag][1][1:] and bits_equal(again[0], res[flag][1][0])
    for (a, al, ah), (b, bl, bh) in zip(res["0"], res["1"]):
        assert (al, ah) == (bl, bh) and bits_equal(a, b)
    rc, want, wl, wh, _ = oracle.stack_apply(2, frames, None, 2.0, 2.5)
    assert rc == 0 and res["1"][1][1:] == (wl, wh)


def test_group_errors_follow_the_handle(nl):
    from nightlight_amd import capi
    with nl.StackGroup(4, 32, 16, devices=[0, 0]) as g:
        g.set_weights(np.ones(4, np.float32))
        with pytest.raises(capi.NlError) as e:
            g.run(4)                      # weighted MAD: the reference panics, we return its message
        assert e.value.code == capi.ERR_WEIGHTED_MAD
        with pytest.raises(capi.NlError) as e:
            g.run(9)
        assert e.value.message == "invalid stacking mode"
    with pytest.raises(capi.NlError):
        nl.StackGroup(4, 32, 16, devices=[0, 99])


def test_bench_self_launches_ranks_and_reports_strong_scaling():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-device",
           "--frames", "16", "--width", "256", "--height", "96", "--steps", "3", "--warmup", "1", "--no-cpu"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    doc = json.loads(lines[0])
    assert doc["n_gpus"] == 2 and doc["scaling"] == "strong"
    assert doc["config"]["image_rows"] == 96 and doc["config"]["rows_per_gpu"] == 48
    # same stack on one rank: same global counters, twice the rows per GPU
    p1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--frames", "16", "--width", "256",
                         "--height", "96", "--steps", "3", "--warmup", "1", "--no-cpu"],
                        env=env, capture_output=True, text=True, timeout=280)
    assert p1.returncode == 0, p1.stderr[-2000:]
    one = json.loads([ln for ln in p1.stdout.splitlines() if ln.startswith("{")][0])
    assert one["n_gpus"] == 1 and one["config"]["rows_per_gpu"] == 96
    assert (one["config"]["clip_low"], one["config"]["clip_high"]) == (doc["config"]["clip_low"], doc["config"]["clip_high"])
    assert one["roofline"]["timed_passes_averaged"] == 3 and one["roofline"]["kernel_ms"] > 0


def test_eight_ranks_tile_the_headline_stack_on_one_device():
    # rehearsal of the driver's 8-GPU run on this 1-GPU box: 8 gloo ranks share device 0, each owns 512 rows
    # (1 GiB) of the 128 x 4096 x 4096 stack; the bench line must report the 8-way strong-scaling split and the
    # global counters of the 1-rank run (profiles/: 6 836 157 / 13 270 993)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo", "--share-device",
           "--steps", "3", "--warmup", "1", "--no-cpu", "--no-also"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    doc = json.loads(lines[0])
    assert doc["n_gpus"] == 8 and doc["scaling"] == "strong"
    assert doc["config"]["frames"] == 128 and doc["config"]["image_rows"] == 4096 and doc["config"]["rows_per_gpu"] == 512
    assert (doc["config"]["clip_low"], doc["config"]["clip_high"]) == (6836157, 13270993)
    assert doc["roofline"]["algorithmic_bytes"] == 4.0 * 512 * 4096 * 129


# ---- RCCL itself, on the one GPU there is: a process group of world size 1 with backend "nccl" ----------------------
# (internal/ops/stack/stack.go:193-198: the clip totals every pass, and every goal-seek step, reduces.)  The scaling run
# is the driver's; what can be executed here is every line of it except the second rank: communicator set-up, the
# device-to-device copy of the counters behind a pass, an asynchronous all-reduce on the handle's own stream
# (ExternalStream), and the goal-seek's reduction callback backed by the same collective.

def _rccl_worker(_index, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import nightlight_amd as nl
    from nightlight_amd.dist import ShardedStack
    from util import make_frames as mk
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    frames = mk(N, W, H, seed=7300)

    def make_tile(row0, rows):
        st = nl.StackHandle(N, W, H, row0=row0, rows=rows, device=0)
        st.upload_frames(frames)
        return st

    sh = ShardedStack(H, make_tile, world=1, rank=0, device="cuda")       # counters all-reduced as a CUDA tensor: RCCL
    out = {}
    for mode in (2, 3, 5):
        res, cl, ch = sh.run(mode, 2.5, 2.5)
        out["res%d" % mode] = res
        out["clip%d" % mode] = np.array([cl, ch])
    gs = sh.find_sigmas(2, 1.0, 1.0)                                      # nl_reduce_fn -> dist.all_reduce on the device
    # the bench's per-pass protocol: counters copied device-to-device behind the pass, reduced asynchronously on the
    # handle's own stream, the next pass enqueued meanwhile
    st = sh.tile
    totals = torch.zeros(2, dtype=torch.int64, device="cuda")
    stream = torch.cuda.ExternalStream(st.stream_ptr, device=0)
    pending = None
    seen = []
    for it in range(4):
        st.run_async(2, 2.5 + 0.25 * it, 2.5)
        with torch.cuda.stream(stream):
            if pending is not None:
                pending.wait()
                seen.append(totals.clone())
            st.copy_counters_async(totals.data_ptr())
            pending = dist.all_reduce(totals, async_op=True)
    with torch.cuda.stream(stream):
        pending.wait()
    st.finish()
    torch.cuda.synchronize()
    seen.append(totals.clone())
    # round 5, what bench.py does now: no copy kernel -- pass i leaves its counters in ring[i % 3] (the caller's buffer,
    # nl_stack_set_counters_buffer), the all-reduce runs in place on it while pass i + 1 has the device, a buffer is
    # handed out again behind the collective that used it last
    # ... and the collectives are issued from a stream of their own that waits for pass i through nl_stack_order_stream_after
    # (the pass's stream waits for nothing but a buffer's last collective, and that only if a host-side query finds it running)
    ring = torch.zeros((3, 4), dtype=torch.int64, device="cuda")
    works = [None, None, None]
    comm = torch.cuda.Stream()
    for it in range(7):
        k = it % 3
        if works[k] is not None and (it == 4 or not works[k].is_completed()):      # (it == 4: the waiting branch, once)
            with torch.cuda.stream(stream):
                works[k].wait()
        st.set_counters_buffer(ring[k].data_ptr())
        st.run_async(2, 2.5 + 0.25 * (it % 4), 2.5)
        st.order_stream_after(comm.cuda_stream)
        with torch.cuda.stream(comm):
            works[k] = dist.all_reduce(ring[k][:2], async_op=True)
    with torch.cuda.stream(comm):
        for w_ in works:
            w_.wait()
    comm.synchronize()
    st.finish()
    torch.cuda.synchronize()
    ring_totals = ring[:, :2].cpu().numpy()                 # passes 6, 4, 5 (it = 6 -> k = 0, 4 -> 1, 5 -> 2)
    st.set_counters_buffer(None)
    own = st.run(2, 2.5, 2.5)[1:]
    np.savez(out_path, gs_res=gs[0], gs=np.array(gs[1:], np.float64), backend=np.array([dist.get_backend()]),
             async_totals=np.stack([t.cpu().numpy() for t in seen]), ring_totals=ring_totals, own_after=np.array(own), **out)
    sh.close()
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_world_size_one_runs_the_device_side_reduction(tmp_path, oracle, nl):
    import torch.multiprocessing as mp
    port = 31600 + (os.getpid() % 2000)
    out_path = os.path.join(str(tmp_path), "rccl.npz")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    mp.spawn(_rccl_worker, args=(port, out_path), nprocs=1, join=True)
    d = np.load(out_path)
    assert str(d["backend"][0]) == "nccl"
    frames = make_frames(N, W, H, seed=7300)
    for mode in (2, 3, 5):
        rc, want, wl, wh, _ = oracle.stack_apply(mode, frames, None, 2.5, 2.5)
        assert tuple(d["clip%d" % mode]) == (wl, wh)
        got = d["res%d" % mode]
        assert np.array_equal(np.isnan(got), np.isnan(want))
        ok = ~np.isnan(want)
        assert np.all(np.abs(got[ok].astype(np.float64) - want[ok]) <= 1e-5 * np.abs(want[ok]))
    passes, gres, gcl, gch, gsl, gsh = oracle.find_sigmas_bisect(2, frames, 1.0, 1.0)
    assert tuple(d["gs"][:2]) == (gcl, gch) and int(d["gs"][4]) == passes
    assert (np.float32(d["gs"][2]), np.float32(d["gs"][3])) == (gsl, gsh)
    # the asynchronous per-pass totals: pass i's counters, reduced while pass i+1 ran
    want_by_it = {}
    for it in range(4):
        rc, _, wl, wh, _ = oracle.stack_apply(2, frames, None, 2.5 + 0.25 * it, 2.5)
        want_by_it[it] = (wl, wh)
        assert tuple(int(x) for x in d["async_totals"][it]) == (wl, wh), it
    # the ring protocol (nl_stack_set_counters_buffer): buffer k holds the totals of the last pass that wrote it
    for k, it in ((0, 6), (1, 4), (2, 5)):
        assert tuple(int(x) for x in d["ring_totals"][k]) == want_by_it[it % 4], (k, it)
    assert tuple(int(x) for x in d["own_after"]) == want_by_it[0]            # back on the handle's own buffer


def test_bench_force_dist_reports_the_rccl_protocol_on_one_gpu():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--frames", "16", "--width", "256", "--height", "96",
            "--steps", "3", "--warmup", "1", "--no-cpu"]
    p = subprocess.run(base + ["--force-dist"], env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    doc = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert doc["n_gpus"] == 1
    assert "on the device (RCCL, the pass's own stream)" in doc["config"]["sharding"]
    p1 = subprocess.run(base, env=env, capture_output=True, text=True, timeout=280)
    assert p1.returncode == 0, p1.stderr[-2000:]
    one = json.loads([ln for ln in p1.stdout.splitlines() if ln.startswith("{")][0])
    assert "RCCL" not in one["config"]["sharding"]
    assert (one["config"]["clip_low"], one["config"]["clip_high"]) == (doc["config"]["clip_low"], doc["config"]["clip_high"])
    assert "fresh_handle" in one and one["fresh_handle"]["ms_first_pass_fresh_handle"] > 0
    assert one["fresh_handle"]["clip_counters"] == [one["config"]["clip_low"], one["config"]["clip_high"]]


def test_bench_apply_from_host_leg_matches_the_resident_pass():
    # bench.py apply_from_host (round 6): OpStack.Apply from pageable host frames through nl_group_create /
    # nl_group_upload_frame[_fits] / nl_group_run / nl_group_destroy, as go/stackhip/stack_hip.go calls them -- here on a small
    # stack (--apply runs the leg for any geometry): the fp32 leg must reproduce the resident pass's result and counters
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--frames", "12", "--width", "512", "--height", "96",
           "--steps", "2", "--warmup", "1", "--preheat-steps", "1", "--no-cpu", "--no-also", "--apply"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    doc = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    a = doc["apply_from_host"]
    assert a["fp32"]["result_equals_resident_pass"] and a["fp32"]["counters_equal_resident_pass"]
    for leg in ("fp32", "fits_int16"):
        assert a[leg]["wall_ms"] > 0 and a[leg]["upload_gib_s"] > 0
        assert set(a[leg]["share"]) == {"upload", "pass", "download", "create_destroy"}
    assert a["fits_int16"]["host_bytes"] * 2 == a["fp32"]["host_bytes"]
