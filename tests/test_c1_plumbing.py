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


def test_product_fits_framing_matches_the_reference_restatement(fits_files, tmp_path):
    # include/nlstack.h nl_fits_write_header / nl_fits_parse_header (host code of the product library, no device needed)
    # against oracle/fitsio.py: same header bytes, same fields, same payload offset; the reference's error messages
    import nightlight_amd as nla
    from nightlight_amd import capi
    from oracle import fitsio
    paths, frames = fits_files
    for naxisn, bz, bs, ex in (([W, H], 0.0, 1.0, 300.0), ([4, 3], 0.0, 1.0, 0.0), ([6000, 4000], 32768.0, 1.0, 12.5),
                               ([7, 5, 3], 0.5, 2.0, 1e6), ([16], -1.0, 0.25, 1.0 / 3.0)):
        # (oracle/fitsio.py formats %g only for the values its own tests write: integral ones and short decimals)
        want = fitsio.header_bytes(naxisn, bz, bs, ex) if ex in (300.0, 0.0, 12.5) else None
        got = nla.fits_write_header(naxisn, bz, bs, ex)
        assert len(got) % 2880 == 0 and got[:30] == b"SIMPLE  =                    T"
        if want is not None:
            assert got == want, (naxisn, bz, bs, ex)
        info = nla.fits_parse_header(got + b" " * 2880)
        assert (info["bitpix"], info["naxisn"], info["header_bytes"]) == (-32, list(naxisn), len(got))
        # (the reference cannot read back what its own %g writes as 1e+06: its float grammar wants a decimal point and
        # an upper-case exponent, read.go:545 -- the card is skipped with a warning and the exposure stays 0; mirrored)
        assert (info["bzero"], info["bscale"], info["exposure"]) == (np.float32(bz), np.float32(bs), np.float32(ex if ex < 1e6 else 0))
        assert info["payload_bytes"] == 4 * int(np.prod(naxisn))
        assert info["padded_payload_bytes"] == nla.fits_padded_bytes(info["payload_bytes"]) == (info["payload_bytes"] + 2879) // 2880 * 2880
    assert b"EXPOSURE=                1e+06 /" in nla.fits_write_header([7, 5, 3], 0.5, 2.0, 1e6)          # fmt %g of a float32
    assert b"EXPOSURE=           0.33333334 /" in nla.fits_write_header([16], -1.0, 0.25, 1.0 / 3.0)     # shortest digits that round-trip
    assert b"BZERO   =                   -1 /" in nla.fits_write_header([16], -1.0, 0.25, 1.0 / 3.0)
    # strconv decides %e for the shortest form with a precision of 6 whatever the digit count (ftoa.go): seven and eight
    # significant digits at exponent >= 6 still take the exponent form
    assert b"EXPOSURE=         1.234567e+06 /" in nla.fits_write_header([16], 0.0, 1.0, 1234567.0)
    assert b"EXPOSURE=        1.6777216e+07 /" in nla.fits_write_header([16], 0.0, 1.0, 16777216.0)
    assert b"EXPOSURE=               123456 /" in nla.fits_write_header([16], 0.0, 1.0, 123456.0)
    assert b"EXPOSURE=             999999.9 /" in nla.fits_write_header([16], 0.0, 1.0, 999999.9)
    assert b"BZERO   =               0.0001 /" in nla.fits_write_header([16], 1e-4, 1.0, 0.0)
    assert b"BZERO   =                1e-05 /" in nla.fits_write_header([16], 1e-5, 1.0, 0.0)
    raw = open(paths[0], "rb").read()
    info = nla.fits_parse_header(raw, 7)
    oinfo, off = fitsio.read_header(paths[0])
    assert info["header_bytes"] == off and info["naxisn"] == oinfo["naxisn"] and info["bitpix"] == oinfo["bitpix"]
    assert (info["bzero"], info["bscale"], info["exposure"]) == (oinfo["bzero"], oinfo["bscale"], oinfo["exposure"])
    # EXPTIME stands in for EXPOSURE (read.go:135-139); an integer card is accepted for a float key (:77-86); a D exponent is dropped
    hdr = bytearray(nla.fits_write_header([8, 2], 0.0, 1.0, 0.0))
    card = ("%-8s= %20s / %-47s" % ("EXPTIME", "120", "")).encode()
    end = hdr.index(b"END" + b" " * 77)
    hdr[end:end + 80] = card
    hdr[end + 80:end + 160] = b"END" + b" " * 77
    assert nla.fits_parse_header(bytes(hdr))["exposure"] == np.float32(120)
    hdr[end:end + 80] = ("%-8s= %20s / %-47s" % ("EXPTIME", "1.2D2", "")).encode()
    assert nla.fits_parse_header(bytes(hdr))["exposure"] == np.float32(0)
    # errors, with the reference's texts (read.go:103-105, 65-70, 452-454)
    bad = bytearray(raw[:2880])
    bad[0:80] = ("%-8s= %20s / %-47s" % ("SIMPLE", "F", "")).encode()
    with pytest.raises(capi.NlError) as e:
        nla.fits_parse_header(bytes(bad), 3)
    assert e.value.message == "3: Not a valid FITS file; SIMPLE=T missing in header"
    bad = bytearray(raw[:2880])
    i = bad.index(b"NAXIS2")
    bad[i:i + 80] = b" " * 80
    with pytest.raises(capi.NlError) as e:
        nla.fits_parse_header(bytes(bad), 5)
    assert e.value.message == "5: FITS header does not contain key NAXIS2"
    with pytest.raises(capi.NlError) as e:
        nla.fits_parse_header(raw[:1000], 1)                     # no END inside the bytes given
    assert e.value.message == "1: unexpected EOF"
    # untrusted headers: a negative axis, a BITPIX the decoder does not know, axes whose product overflows 63 bits
    def with_card(key, value):
        b = bytearray(raw[:2880])
        i = b.index(key.ljust(8).encode() + b"=")
        b[i:i + 80] = ("%-8s= %20s / %-47s" % (key, value, "")).encode()
        return bytes(b)
    for key, value, text in (("NAXIS1", "-4", "negative NAXIS1 -4"), ("BITPIX", "-16", "unsupported BITPIX -16"),
                             ("BITPIX", "24", "unsupported BITPIX 24")):
        with pytest.raises(capi.NlError) as e:
            nla.fits_parse_header(with_card(key, value), 2)
        assert e.value.message == "2: " + text
    big = bytearray(nla.fits_write_header([2147483647] * 8, 0.0, 1.0, 0.0))
    with pytest.raises(capi.NlError) as e:
        nla.fits_parse_header(bytes(big), 4)
    assert "overflow" in e.value.message


@pytest.mark.gpu
def test_c1_gpu_file_to_file_through_the_product(fits_files, oracle, nl, tmp_path):
    # the whole C1 path without oracle/fitsio.py: file image -> nl_fits_parse_header -> payload decoded on the device ->
    # mean stack -> result encoded on the device -> nl_fits_write_header + padding -> file; the oracle only checks
    from oracle import fitsio
    paths, frames = fits_files
    with nl.StackHandle(N, W, H) as st:
        exposure = np.float32(0)
        for k, p in enumerate(paths):
            raw = open(p, "rb").read()
            info = nl.fits_parse_header(raw, k)
            assert info["naxisn"] == [W, H] and info["bitpix"] == -32
            payload = np.frombuffer(raw, np.uint8, info["payload_bytes"], info["header_bytes"])
            st.upload_frame_fits(k, payload, info["bitpix"], info["bscale"], info["bzero"])
            exposure = np.float32(exposure + info["exposure"])       # stack.go:221-225
        st.run(nl.ST_MEAN)
        body = np.asarray(st.download_result_fits(), np.uint8).tobytes()
    out = str(tmp_path / "stack.fits")
    with open(out, "wb") as f:
        f.write(nl.fits_write_header([W, H], 0.0, 1.0, float(exposure)))
        f.write(body + b" " * (nl.fits_padded_bytes(len(body)) - len(body)))
    # checker: the reference restatement reads the file back and finds the sequential fp32 mean
    oinfo, raw_back = fitsio.read_payload(out)
    assert (oinfo["naxisn"], float(oinfo["exposure"]), os.path.getsize(out) % 2880) == ([W, H], 300.0 * N, 0)
    rc, data, *_ = oracle.fits_decode(raw_back, -32)
    assert rc == 0 and bits_equal(data, numpy_sequential_mean(frames))


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
