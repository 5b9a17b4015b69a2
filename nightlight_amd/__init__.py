"""nightlight_amd -- MI355X (gfx950) implementation of Nightlight's per-pixel
stacking hot path behind the C ABI in include/nlstack.h.

The product is libnlstack.so (hand-written HIP kernels + C ABI, csrc/); the
Python modules here are plumbing for tests, bench.py and multi-GPU sharding.
"""
from . import capi  # noqa: F401
from .capi import (NlError, ST_AUTO, ST_LINEAR_FIT, ST_MAD_SIGMA, ST_MEAN, ST_MEDIAN,  # noqa: F401
                   ST_SIGMA, ST_WINSOR_SIGMA, WEIGHT_EXPOSURE, WEIGHT_INVERSE_HFR,
                   WEIGHT_INVERSE_NOISE, WEIGHT_NONE)
from .stack import (StackGroup, StackHandle, fits_padded_bytes, fits_parse_header, fits_write_header,  # noqa: F401
                    median_filter_3x3, median_filter_mask, weights_from_scalars)

__version__ = "0.2.0"
