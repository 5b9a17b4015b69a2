#!/usr/bin/env python3
"""Exercise the ingest kernels (FITS decode, Project, affine, encode) on whole
4096x4096 frames; run under rocprofv3 --kernel-trace --stats to get their
durations (tools/gpu_profile.sh style).  Prints the algorithmic bytes per launch
so the summary can be turned into GB/s."""
import sys

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightlight_amd.stack import StackHandle

W = H = 4096
n = W * H
rng = np.random.default_rng(1)
raw16 = np.frombuffer(rng.integers(-32768, 32767, n, dtype=np.int16).astype(">i2").tobytes(), np.uint8)
raw32 = np.frombuffer((rng.standard_normal(n) * 100).astype(">f4").tobytes(), np.uint8)
src = (1000 + 30 * rng.standard_normal(n)).astype(np.float32)
with StackHandle(4, W, H) as st:
    for _ in range(3):
        st.upload_frame_fits(0, raw16, 16, 1.0, 32768.0, 1.01, -3.0)
        st.upload_frame_fits(1, raw32, -32, 1.0, 0.0)
        st.upload_frame_projected(2, src, W, H, [0.9999, 0.01, -3.2, -0.01, 0.9999, 4.7], float("nan"), 1.01, -3.0)
        st.upload_frame_projected(3, src, W, H, [1, 0, 0.5, 0, 1, 0.25])
        st.frame_affine(3, 1.01, 0.5)
    st.run(1, fetch=False)
    st.download_result_fits()
print("algorithmic bytes per launch: decode<16> %d, decode<-32> %d, project %d, affine %d, encode %d"
      % (n * 6, n * 8, n * 8, n * 8, n * 8))
