"""The C++ host mirror of the reference's operator surface
(nightlight_amd/host/, mirroring internal/ops/operator.go and
internal/ops/stack/stack.go:66-227): JSON defaults, error messages and -- on
the GPU -- results, log lines and exposure bookkeeping against the oracle."""
import json

import numpy as np
import pytest

from util import make_frames


def test_json_defaults_follow_new_op_stack_default():
    # stack.go:77 NewOpStackDefault = (StAuto, none, 2.75, 2.75); :92-99 missing keys keep them
    from nightlight_amd import operator as op
    assert json.loads(op.op_stack_roundtrip_json('{"type":"stack"}')) == {
        "type": "stack", "mode": 6, "weighting": 0, "sigmaLow": 2.75, "sigmaHigh": 2.75}
    got = json.loads(op.op_stack_roundtrip_json('{"type":"stack","mode":3,"sigmaHigh":4.5,"weighting":1}'))
    assert got == {"type": "stack", "mode": 3, "weighting": 1, "sigmaLow": 2.75, "sigmaHigh": 4.5}


def test_error_messages_without_touching_a_device():
    from nightlight_amd import operator as op
    f = [np.ones(16, np.float32)] * 3
    with pytest.raises(op.OperatorError, match="^stack operator needs inputs$"):           # stack.go:103
        op.op_stack_apply_json('{"type":"stack"}', [], 4, 4)
    with pytest.raises(op.OperatorError, match="^invalid stacking mode$"):                 # stack.go:119
        op.op_stack_apply_json('{"type":"stack","mode":7}', f, 4, 4)
    with pytest.raises(op.OperatorError,
                       match="^1: Missing exposure information for exposure-weighted stacking$"):   # :238
        op.op_stack_apply_json('{"type":"stack","mode":1,"weighting":1}', f, 4, 4, exposure=[30, 0, 30])
    with pytest.raises(op.OperatorError, match="^Invalid weighting mode 9"):               # stack.go:267
        op.op_stack_apply_json('{"type":"stack","mode":1,"weighting":9}', f, 4, 4)
    with pytest.raises(op.OperatorError, match="Unknown operator type"):
        op.op_stack_apply_json('{"type":"stackX"}', f, 4, 4)


@pytest.mark.gpu
def test_operator_apply_matches_oracle_and_logs_like_the_reference(nl, oracle):
    from nightlight_amd import operator as op
    width, height, n = 40, 12, 20
    frames = make_frames(n, width, height, seed=21)
    exposure = np.full(n, 30.0, np.float32)
    out, exp_sum, log = op.op_stack_apply_json(
        '{"type":"stack","mode":3,"weighting":0,"sigmaLow":2.5,"sigmaHigh":3}', list(frames),
        width, height, exposure=exposure)
    rc, want, wl, wh, _ = oracle.stack_apply(3, frames, None, 2.5, 3.0)
    # default dispatch = register-resident winsor kernel: counters exact, values to summation order
    assert np.array_equal(np.isnan(out), np.isnan(want))
    np.testing.assert_allclose(out, want, rtol=1e-5, atol=0, equal_nan=True)
    assert exp_sum == 600.0                                                                # stack.go:220-225
    from nightlight_amd.dist import clipped_log_line
    assert log == ("Stacking %d frames with stacking mode 3 and sigma low 2.5 high 3:\n" % n
                   + clipped_log_line(wl, wh, width * height, n))                          # stack.go:124,214-218


@pytest.mark.gpu
def test_operator_auto_mode_exposure_weights_and_skipped_frames(nl, oracle):
    from nightlight_amd import operator as op
    width, height, n = 24, 10, 8
    frames = make_frames(n, width, height, seed=22)
    exposure = np.linspace(10, 80, n).astype(np.float32)
    # frame 3 was dropped upstream ("(nil, nil)", operator.go:119-131): stack the other 7
    ins = [None if i == 3 else frames[i] for i in range(n)]
    out, exp_sum, log = op.op_stack_apply_json('{"type":"stack","weighting":1}', ins, width, height,
                                               exposure=exposure)
    keep = np.delete(frames, 3, axis=0)
    w = np.delete(exposure, 3)
    rc, want, wl, wh, mode = oracle.stack_apply(6, keep, w, 2.75, 2.75)
    assert mode == 2 and log.startswith("Stacking 7 frames with stacking mode 2 and sigma low 2.75 high 2.75:\n")
    assert np.array_equal(out, want, equal_nan=True)
    assert exp_sum == float(np.float32(sum(np.float32(x) for x in w)))
    assert "Clipped low %d " % wl in log and " high %d " % wh in log


def _partition_py(num_frames, width, height, stack_memory_mb, max_threads):
    """stackbatches.go:139-186 in Python ints (no dark / flat frame)."""
    bytes_ = width * height * 4
    available = (stack_memory_mb * 1024 * 1024) // bytes_
    mt, bs, nb = max_threads, 0, 0
    while mt >= 1:
        bs = available - mt
        if bs >= 2:
            nb = (num_frames + bs - 1) // bs
            if nb > 1:
                bs -= 2
            if bs >= 2 and bs >= mt:
                break
        mt -= 1
    if mt < 1 or bs < 2:
        return None
    while (bs - 1) * nb >= num_frames:
        bs -= 1
    return nb, bs, mt


def test_stack_batches_errors_without_touching_a_device():
    from nightlight_amd import operator as op
    with pytest.raises(op.OperatorError, match="^No frames to batch process"):              # stackbatches.go:48
        op.op_stack_batches_apply_json('{"type":"stack"}', [], 8, 8, stack_memory_mb=64)
    f = [np.ones(64 * 64, np.float32)] * 10
    # 64x64 fp32 = 16 KiB per frame; 0 MiB of stack memory fits nothing (:181-183)
    with pytest.raises(op.OperatorError,
                       match="^Cannot find a stacking execution path within the given memory constraints."):
        op.op_stack_batches_apply_json('{"type":"stack"}', f, 64, 64, stack_memory_mb=0)
    assert _partition_py(10, 64, 64, 0, 4) is None


@pytest.mark.gpu
def test_stack_batches_partition_log_and_stack_of_stacks(nl, oracle):
    from nightlight_amd import operator as op
    width, height, n = 512, 512, 23                 # 1 MiB per frame
    frames = make_frames(n, width, height, seed=41)
    exposure = np.full(n, 10.0, np.float32)
    out, exp_sum, log = op.op_stack_batches_apply_json(
        '{"type":"stack","mode":2,"sigmaLow":2.5,"sigmaHigh":2.5}', list(frames), width, height,
        exposure=exposure, max_threads=2, memory_mb=20, stack_memory_mb=12)
    nb, bs, mt = _partition_py(n, width, height, 12, 2)
    assert (nb, bs, mt) == (3, 8, 2)
    assert "\nEstimating memory needs for 23 images from frame0.fits:\n" in log                     # :134
    assert "23 images of 512x512 pixels (0.3 MPixels), which each take 1 MiB in-memory as floating point.\n" in log
    assert "CPU has 2 threads. Physical memory is 20 MiB, -op.Memory is 12 MiB, this fits 12 frames.\n" in log
    assert "Using 3 random batches of size 8 with 2 images in parallel.\n" in log                   # :187
    assert "Randomizing input files into batches...\n" in log
    for b, k in ((1, 8), (2, 8), (3, 7)):
        assert "\nStarting batch %d of 3 with %d frames...\n" % (b, k) in log                       # :78
    assert exp_sum == float(np.float32(sum(np.float32(x) for x in exposure)))
    _check_batches_against_oracle(op, oracle, frames, width, height, exposure, (3, 8), devices=None)


def _oracle_stack_of_stacks(oracle, frames, perm, bs, mode, sl, sh):
    """The reference's batch loop (stackbatches.go:69-116) on the oracle for a given partition:
    per-batch stack with the frames in perm order, StackIncremental weighted by the batch frame
    count, StackIncrementalFinalize.  Returns (result, [(clipLow, clipHigh) per batch])."""
    acc, clips, total = None, [], 0
    for start in range(0, len(perm), bs):
        idx = perm[start:start + bs]
        rc, res, cl, ch, _ = oracle.stack_apply(mode, np.ascontiguousarray(frames[idx]), None, sl, sh, 0.0, num_cpu=4)
        assert rc == 0
        clips.append((cl, ch))
        acc = oracle.stack_incremental(np.zeros_like(res) if acc is None else acc, res, float(len(idx)),
                                       first=acc is None)
        total += len(idx)
    return oracle.stack_incremental_finalize(acc, float(total)), clips


def _check_batches_against_oracle(op, oracle, frames, width, height, exposure, shape, devices):
    import re
    nb, bs = shape
    n = frames.shape[0]
    op.set_devices(devices)
    try:
        # (a) mean per batch: every kernel on the way is bit-exact, so the whole stack of stacks must
        # equal the oracle's for the SAME partition bit for bit -- this sees the frame order inside a
        # batch (fp32 sums in frame order) and the device-side StackIncremental / Finalize
        out, _, log, perm = op.op_stack_batches_apply_json('{"type":"stack","mode":1}', list(frames), width, height,
                                                           exposure=exposure, max_threads=2, memory_mb=20,
                                                           stack_memory_mb=12, return_perm=True)
        assert sorted(perm) == list(range(n))                            # every frame in exactly one batch
        assert perm != list(range(n))                                    # batches are random ...
        for start in range(0, n, bs):                                    # ... and sorted inside (stackbatches.go:199-209)
            assert perm[start:start + bs] == sorted(perm[start:start + bs])
        want, _ = _oracle_stack_of_stacks(oracle, frames, perm, bs, 1, 0.0, 0.0)
        assert np.array_equal(out.view(np.uint32), want.view(np.uint32)), \
            "%d pixels differ" % np.count_nonzero(out.view(np.uint32) != want.view(np.uint32))
        # (b) sigma clipping per batch (register-resident kernels: counters exact, values 1e-5): the
        # per-batch "Clipped low" log lines carry the oracle's counters for that partition
        out, _, log, perm2 = op.op_stack_batches_apply_json(
            '{"type":"stack","mode":2,"sigmaLow":2.5,"sigmaHigh":2.5}', list(frames), width, height,
            exposure=exposure, max_threads=2, memory_mb=20, stack_memory_mb=12, return_perm=True)
        assert perm2 == perm                                             # fixed-seed permutation
        want, clips = _oracle_stack_of_stacks(oracle, frames, perm, bs, 2, 2.5, 2.5)
        got_clips = [(int(a), int(b)) for a, b in re.findall(r"Clipped low (\d+) \([0-9.]+%\) high (\d+) ", log)]
        assert got_clips == clips
        ok = ~np.isnan(want) & (want != 0)
        assert np.array_equal(np.isnan(out), np.isnan(want))
        assert np.max(np.abs(out[ok].astype(np.float64) - want[ok]) / np.abs(want[ok])) <= 1e-5
    finally:
        op.set_devices(None)


@pytest.mark.gpu
def test_stack_batches_over_several_device_tiles(nl, oracle):
    # the same batch loop with every stack fanned out over 3 row tiles (nl_group_*; all on device 0
    # here): tiles never change a pixel's arithmetic, so the same oracle comparison must hold
    from nightlight_amd import operator as op
    width, height, n = 512, 512, 23
    frames = make_frames(n, width, height, seed=41)
    _check_batches_against_oracle(op, oracle, frames, width, height, np.full(n, 10.0, np.float32), (3, 8),
                                  devices=[0, 0, 0])


@pytest.mark.gpu
def test_op_stack_apply_over_several_device_tiles(nl, oracle):
    from nightlight_amd import operator as op
    width, height, n = 96, 50, 12
    frames = make_frames(n, width, height, seed=43)
    op.set_devices([0, 0, 0, 0])
    try:
        out, _, log = op.op_stack_apply_json('{"type":"stack","mode":3}', list(frames), width, height)
    finally:
        op.set_devices(None)
    rc, want, wl, wh, _ = oracle.stack_apply(3, frames, None, 2.75, 2.75)
    assert "Clipped low %d (" % wl in log and " high %d (" % wh in log           # host-summed over the tiles
    ok = ~np.isnan(want) & (want != 0)
    assert np.max(np.abs(out[ok].astype(np.float64) - want[ok]) / np.abs(want[ok])) <= 1e-5


@pytest.mark.gpu
def test_stack_batches_single_batch_is_the_plain_stack(nl, oracle):
    from nightlight_amd import operator as op
    width, height, n = 64, 32, 9
    frames = make_frames(n, width, height, seed=42)
    out, _, log = op.op_stack_batches_apply_json('{"type":"stack","mode":0}', list(frames), width, height,
                                                 max_threads=2, memory_mb=64, stack_memory_mb=32)
    assert "Using 1 random batches of size 9 with 2 images in parallel.\n" in log
    assert "Randomizing" not in log
    rc, want, _, _, _ = oracle.stack_apply(0, frames, None, 0, 0)
    assert np.array_equal(out, want, equal_nan=True)
