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
