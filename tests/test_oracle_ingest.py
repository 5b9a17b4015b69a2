"""The oracle's restatement of the formats / steps either side of the stack
(fits/read.go, fits/write.go, fits/pixelops.go:601-605, fits/project.go,
star/coord.go) against hand-derived answers and against an independent numpy
formulation (struct / numpy big-endian views; per-pixel Python loop for the
projection).  The reference has no tests or fixtures for these: parity is pinned
by these known answers only."""
import struct

import numpy as np
import pytest


def test_decode_int16_with_bzero_known_answer(oracle):
    # the usual unsigned-16 convention: BZERO = 32768 (read.go:234-235)
    raw = bytes([0x80, 0x00, 0x7F, 0xFF, 0x00, 0x01, 0xFF, 0xFF])       # -32768, 32767, 1, -1
    rc, v, mn, mx, mean = oracle.fits_decode(raw, 16, 1.0, 32768.0)
    assert rc == 0
    assert v.tolist() == [0.0, 65535.0, 32769.0, 32767.0]
    assert (mn, mx) == (0.0, 65535.0)
    assert mean == np.float32((0.0 + 65535.0 + 32769.0 + 32767.0) / 4)


def test_decode_uint8_and_unknown_bitpix(oracle):
    rc, v, mn, mx, mean = oracle.fits_decode(bytes([0, 1, 255]), 8, 2.0, -1.0)
    assert rc == 0 and v.tolist() == [-1.0, 1.0, 509.0] and (mn, mx) == (-1.0, 509.0)
    assert oracle.fits_decode(bytes(8), 24)[0] == -1                      # read.go:169


@pytest.mark.parametrize("bitpix,fmt,dtype", [(8, "B", np.uint8), (16, ">h", ">i2"), (32, ">i", ">i4"),
                                              (64, ">q", ">i8"), (-32, ">f", ">f4"), (-64, ">d", ">f8")])
def test_decode_matches_numpy_big_endian_views(oracle, bitpix, fmt, dtype):
    rng = np.random.default_rng(abs(bitpix))
    n = 1003
    if bitpix > 0:
        info = np.iinfo(np.dtype(dtype).newbyteorder("="))
        vals = rng.integers(info.min, info.max, n, dtype=np.dtype(dtype).newbyteorder("="), endpoint=True)
    else:
        vals = (rng.standard_normal(n) * 1e3).astype(np.dtype(dtype).newbyteorder("="))
    raw = vals.astype(dtype).tobytes()
    assert raw[:struct.calcsize(fmt)] == struct.pack(fmt, vals[0].item())
    bscale, bzero = np.float32(0.25), np.float32(-7.5)
    rc, v, mn, mx, mean = oracle.fits_decode(raw, bitpix, bscale, bzero)
    want = (vals.astype(np.float32) * bscale + bzero).astype(np.float32)   # two fp32 roundings, like the reference
    assert rc == 0 and np.array_equal(v, want)
    assert mn == want.min() and mx == want.max()
    assert mean == np.float32(np.sum(want.astype(np.float64)) / n) or abs(mean - want.astype(np.float64).mean()) < 1e-3


def test_encode_is_big_endian_and_zeroes_nans(oracle):
    x = np.array([1.5, np.nan, -2.0, np.inf], np.float32)
    raw = oracle.fits_encode(x, True)
    assert raw.tobytes() == struct.pack(">4f", 1.5, 0.0, -2.0, float("inf"))        # write.go:191
    keep = oracle.fits_encode(x, False)
    assert np.isnan(np.frombuffer(keep.tobytes(), ">f4")[1])
    rc, back, *_ = oracle.fits_decode(raw, -32)
    assert back.tolist() == [1.5, 0.0, -2.0, float("inf")]


def test_affine_is_multiply_then_add_in_fp32(oracle):
    x = np.array([0.1, 3.0, np.nan, -1e30], np.float32)
    m, o = np.float32(1.1), np.float32(0.3)
    got = oracle.affine(x, m, o)
    want = ((x * m).astype(np.float32) + o).astype(np.float32)
    assert np.array_equal(got, want, equal_nan=True)


def test_transform_invert_known_answers(oracle):
    rc, inv = oracle.transform_invert([1, 0, 5, 0, 1, -3])                # pure shift
    assert rc == 0 and inv.tolist() == [1.0, -0.0, -5.0, -0.0, 1.0, 3.0]
    rc, inv = oracle.transform_invert([0, -1, 0, 1, 0, 0])                # 90 degree rotation
    assert rc == 0 and [abs(x) for x in inv.tolist()] == [0.0, 1.0, 0.0, 1.0, 0.0, 0.0]
    assert inv[1] == 1.0 and inv[3] == -1.0
    assert oracle.transform_invert([1, 2, 0, 2, 4, 0])[0] == -1           # singular (coord.go:160-163)


def _project_py(src, sw, sh, dw, dh, inv, oob):
    """project.go:40-73 in numpy-float32 scalars, one pixel at a time."""
    f = np.float32
    out = np.empty(dw * dh, np.float32)
    for row in range(dh):
        for col in range(dw):
            X = f(f(f(inv[0] * f(col)) + f(inv[1] * f(row))) + inv[2])
            Y = f(f(f(inv[3] * f(col)) + f(inv[4] * f(row))) + inv[5])
            xl, yl = int(np.floor(X)), int(np.floor(Y))
            if xl < 0 or xl + 1 >= sw or yl < 0 or yl + 1 >= sh:
                out[col + row * dw] = oob
                continue
            xr, yr = f(X - f(xl)), f(Y - f(yl))
            p = xl + yl * sw
            vyl = f(f(src[p] * f(1 - xr)) + f(src[p + 1] * xr))
            vyh = f(f(src[p + sw] * f(1 - xr)) + f(src[p + sw + 1] * xr))
            out[col + row * dw] = f(f(vyl * f(1 - yr)) + f(vyh * yr))
    return out


@pytest.mark.parametrize("trans", [[1, 0, 0.5, 0, 1, 0.25],                       # sub-pixel shift
                                   [0.99, 0.05, -3.2, -0.05, 0.99, 4.7],          # small rotation + shift
                                   [1, 0, 0, 0, 1, 0]])                           # identity: last row/column fall out
def test_project_matches_scalar_python_restatement(oracle, trans):
    rng = np.random.default_rng(3)
    sw, sh, dw, dh = 23, 17, 25, 15
    src = rng.standard_normal(sw * sh).astype(np.float32)
    rc, inv = oracle.transform_invert(trans)
    assert rc == 0
    rc, got = oracle.project_bilinear(src, sw, sh, dw, dh, trans, np.nan)
    want = _project_py(src, sw, sh, dw, dh, inv, np.float32(np.nan))
    assert rc == 0 and np.array_equal(got, want, equal_nan=True)
    assert np.isnan(got).any() and (~np.isnan(got)).any()


def test_project_identity_known_answer(oracle):
    # identity transform: every pixel whose right / lower neighbour exists is reproduced
    # exactly (weights 1 and 0); the last row and column become out of bounds (project.go:56)
    src = np.arange(12, dtype=np.float32)
    rc, got = oracle.project_bilinear(src, 4, 3, 4, 3, [1, 0, 0, 0, 1, 0], -1.0)
    assert rc == 0
    assert got.tolist() == [0, 1, 2, -1, 4, 5, 6, -1, -1, -1, -1, -1]
