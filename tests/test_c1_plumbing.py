"""BASELINE.json configs[0] -- the plumbing check BASELINE.md section 3 promises:
write 16 synthetic 1024x1024 BITPIX -32 FITS files, read them back, mean-stack,
compare with a numpy fp32 sequential mean.

CPU leg (no GPU): files -> oracle.fits_decode (read.go:351-395) -> oracle
StackMean (stack.go:307-332) == numpy sequential fp32 mean.
GPU leg: the same files' payload bytes -> nl_stack_upload_frame_fits (decode on
the device) -> mean stack -> bit-identical to the CPU leg; the result written
back through nl_stack_download_result_fits is byte-identical to the oracle
writer's payload (write.go:182-200, NaN -> 0).
"""
import os

import numpy as np
import pytest

from util import bits_equal

N, W, H = 16, 1024, 1024


def synth_frames(seed=2024):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    sky = (200.0 * (0.5 * xx / W + 0.5 * yy / H)).astype(np.float32).reshape(-1)
    frames = []
    for k in range(N):
        f = (np.float32(1000.0 + 5.0 * np.sin(k)) + sky +
             np.float32(30.0) * rng.standard_normal(W * H).astype(np.float32)).astype(np.float32)
        frames.append(f)
    return frames


def numpy_sequential_mean(frames):
    """StackMean for NaN-free frames: fp32 sum in frame order, one fp32 divide (stack.go:318-331)."""
    s = np.zeros(W * H, np.float32)
    for f in frames:
        s = (s + f).astype(np.float32)
    return (s / np.float32(len(frames))).astype(np.float32)


@pytest.fixture(scope="module")
def fits_files(tmp_path_factory):
    from oracle import fitsio
    d = tmp_path_factory.mktemp("c1")
    frames = synth_frames()
    paths = []
    for k, f in enumerate(frames):
        p = str(d / ("light_%02d.fits" % k))
        fitsio.write_f32(p, f, [W, H], exposure=300.0)
        paths.append(p)
    return paths, frames


def test_c1_files_are_wellformed(fits_files):
    from oracle import fitsio
    paths, frames = fits_files
    for p in paths[:3]:
        size = os.path.getsize(p)
        assert size % 2880 == 0                                    # header and payload padded to blocks
        info, off = fitsio.read_header(p)
        assert off == 2880 and size == 2880 + (W * H * 4 + 2879) // 2880 * 2880
        assert (info["bitpix"], info["naxisn"], float(info["bzero"]), float(info["bscale"]),
                float(info["exposure"]), info["program"]) == (-32, [W, H], 0.0, 1.0, 300.0, "nightlight")
    head = open(paths[0], "rb").read(160).decode()
    assert head[:80] == "SIMPLE  =                    T /     FITS standard 4.0" + " " * 26
    assert head[80:160].startswith("BITPIX  =                  -32 /     32-bit floating point")


def test_c1_cpu_mean_stack_from_fits(fits_files, oracle):
    from oracle import fitsio
    paths, frames = fits_files
    decoded = []
    for p, f in zip(paths, frames):
        info, raw = fitsio.read_payload(p)
        rc, data, mn, mx, mean = oracle.fits_decode(raw, info["bitpix"], info["bscale"], info["bzero"])
        assert rc == 0 and bits_equal(data, f)                      # write -> read round trip is lossless
        assert (mn, mx) == (f.min(), f.max())
        decoded.append(data)
    rc, got, _, _, mode = oracle.stack_apply(oracle.ST_MEAN, np.stack(decoded))
    assert rc == 0 and mode == oracle.ST_MEAN
    assert bits_equal(got, numpy_sequential_mean(frames))


def test_c1_nan_becomes_zero_on_write(tmp_path, oracle):
    from oracle import fitsio
    f = np.arange(12, dtype=np.float32)
    f[5] = np.nan
    p = str(tmp_path / "nan.fits")
    fitsio.write_f32(p, f, [4, 3])
    info, raw = fitsio.read_payload(p)
    rc, data, *_ = oracle.fits_decode(raw, -32)
    want = f.copy()
    want[5] = 0.0
    assert bits_equal(data, want) and bits_equal(raw, oracle.fits_encode(f, True))
    assert os.path.getsize(p) == 2 * 2880


@pytest.mark.gpu
def test_c1_gpu_mean_stack_from_fits(fits_files, oracle, nl):
    from oracle import fitsio
    paths, frames = fits_files
    with nl.StackHandle(N, W, H) as st:
        for k, p in enumerate(paths):
            info, raw = fitsio.read_payload(p)
            stats = st.upload_frame_fits(k, raw, info["bitpix"], info["bscale"], info["bzero"])
            assert (stats[0], stats[1]) == (frames[k].min(), frames[k].max())
        got, _, _ = st.run(nl.ST_MEAN)
        assert st.last_kernel_name.startswith("stack_mean")
        raw_out = st.download_result_fits()
    want = numpy_sequential_mean(frames)
    assert bits_equal(got, want)
    rc, cpu, _, _, _ = oracle.stack_apply(oracle.ST_MEAN, np.stack(frames))
    assert bits_equal(got, cpu)
    assert np.array_equal(raw_out, np.frombuffer(fitsio.payload_bytes(want)[: W * H * 4], np.uint8))
