"""Minimal FITS image writer / reader -- TEST INFRASTRUCTURE ONLY (the C1
plumbing check of BASELINE.md section 3), never imported by the product.

Restates, for 2-D BITPIX -32 images (and reading BITPIX 8/16/32/-32/-64):
  Image.Write            /root/reference/internal/fits/write.go:54-89   header cards, END, 2880 padding
  writeBool/Int32/Float32/String  write.go:104-171                       "%-8s= %20s / %-47s"
  writeFloat32Array      write.go:182-215                                big-endian fp32, NaN -> 0, padding
  Image.Read             /root/reference/internal/fits/read.go:97-147    mandatory + optional keys
The payload arithmetic (float32(val)*BSCALE+BZERO) is oracle.fits_decode / the
device's nl_stack_upload_frame_fits; this module only moves bytes and cards.
"""
import re

import numpy as np

BLOCK = 2880


def _go_g(v):
    """fmt's %g of a float32 for the values the tests write (shortest repr; integral -> no point)."""
    v = float(np.float32(v))
    if v == int(v) and abs(v) < 1e21:
        return "%d" % int(v)
    return repr(np.float32(v).item())


def _card(key, value, comment):
    return "%-8s= %20s / %-47s" % (key[:8], value, comment[:47])


def header_bytes(naxisn, bzero=0.0, bscale=1.0, exposure=0.0):
    cards = [_card("SIMPLE", "T", "    FITS standard 4.0"),
             _card("BITPIX", "%d" % -32, "    32-bit floating point"),
             _card("NAXIS", "%d" % len(naxisn), "[1] Number of array dimensions")]
    for i, n in enumerate(naxisn):
        cards.append(_card("NAXIS%d" % (i + 1), "%d" % n, "[1] Array dimension"))
    cards.append(_card("BZERO", _go_g(bzero), "[1] Zero offset"))
    cards.append(_card("BSCALE", _go_g(bscale), "[1] Data scale"))
    if exposure != 0:
        cards.append(_card("EXPOSURE", _go_g(exposure), "[s] Exposure duration"))
    value = "nightlight"
    cards.append("%-8s= '%s'%s / %-47s" % ("PROGRAM", value, " " * (18 - len(value)),
                                            "    https://github.com/mlnoga/nightlight"))
    cards.append("END" + " " * 77)
    assert all(len(c) == 80 for c in cards)
    text = "".join(cards)
    if len(text) % BLOCK:
        text += " " * (BLOCK - len(text) % BLOCK)
    return text.encode("ascii")


def payload_bytes(data):
    """write.go:182-215: network byte order, NaN replaced by 0, last block padded with spaces."""
    d = np.ascontiguousarray(data, np.float32).reshape(-1).copy()
    d[np.isnan(d)] = 0.0
    raw = d.astype(">f4").tobytes()
    if len(raw) % BLOCK:
        raw += b" " * (BLOCK - len(raw) % BLOCK)
    return raw


def write_f32(path, data, naxisn, bzero=0.0, bscale=1.0, exposure=0.0):
    with open(path, "wb") as f:
        f.write(header_bytes(naxisn, bzero, bscale, exposure))
        f.write(payload_bytes(data))


_CARD = re.compile(r"^(?P<k>[A-Z0-9_-]+)\s*=\s*(?:(?P<b>[TF])|(?P<f>[+-]?[0-9]*\.[0-9]*(?:[ED][-+]?[0-9]+)?)"
                   r"|(?P<i>[+-]?[0-9]+)|'(?P<s>[^']*)')\s*(?:/.*)?$")


def read_header(path):
    """Returns (info dict, byte offset of the payload).  Keys as read.go:97-147 uses them."""
    hdr = {}
    with open(path, "rb") as f:
        off = 0
        while True:
            card = f.read(80).decode("ascii")
            off += 80
            if len(card) < 80:
                raise ValueError("FITS header without END")
            if card.startswith("END") and card[3:].strip() == "":
                break
            m = _CARD.match(card.rstrip())
            if not m:
                continue
            if m.group("b") is not None:
                hdr[m.group("k")] = m.group("b") == "T"
            elif m.group("f") is not None:
                hdr[m.group("k")] = np.float32(m.group("f").replace("D", "E"))
            elif m.group("i") is not None:
                hdr[m.group("k")] = int(m.group("i"))
            elif m.group("s") is not None:
                hdr[m.group("k")] = m.group("s").rstrip()
    if not hdr.get("SIMPLE"):
        raise ValueError("Not a valid FITS file; SIMPLE=T missing in header")
    off = (off + BLOCK - 1) // BLOCK * BLOCK
    naxis = hdr["NAXIS"]
    info = {"bitpix": hdr["BITPIX"], "naxisn": [hdr["NAXIS%d" % (i + 1)] for i in range(naxis)],
            "bzero": np.float32(hdr.get("BZERO", 0)), "bscale": np.float32(hdr.get("BSCALE", 1)),
            "exposure": np.float32(hdr.get("EXPOSURE", hdr.get("EXPTIME", 0))), "program": hdr.get("PROGRAM")}
    info["pixels"] = int(np.prod(info["naxisn"]))
    return info, off


def read_payload(path):
    """(info, raw big-endian payload bytes as uint8: exactly pixels*|bitpix|/8 bytes)."""
    info, off = read_header(path)
    nbytes = info["pixels"] * abs(info["bitpix"]) // 8
    with open(path, "rb") as f:
        f.seek(off)
        raw = np.frombuffer(f.read(nbytes), np.uint8)
    if raw.size != nbytes:
        raise ValueError("short FITS payload")
    return info, raw
