#!/usr/bin/env python3
"""PCIe-inclusive rate of the stacking boundary: 128 host frames of 4096x4096 fp32
(8 GiB) into the device buffer, blocking (pageable hipMemcpy) vs the overlapped
pinned-staging path, followed by one sigma-clip pass."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightlight_amd.stack import StackHandle

N, W, H = 128, 4096, 4096
rng = np.random.default_rng(0)
frames = [(1000 + 30 * rng.standard_normal(W * H)).astype(np.float32) for _ in range(8)]   # 8 distinct host frames, reused
gib = N * W * H * 4 / 2**30
with StackHandle(N, W, H) as st:
    for name, up in (("blocking nl_stack_upload_frame", st.upload_frame), ("overlapped nl_stack_upload_frame_async", st.upload_frame_async)):
        for rep in range(2):
            t0 = time.perf_counter()
            for k in range(N):
                up(k, frames[k % 8])
            t_issue = time.perf_counter() - t0
            st.run(2, 3.0, 3.0, fetch=False)
            t_all = time.perf_counter() - t0
        print("%-40s issue %.3f s, upload+pass %.3f s  -> %.1f GiB/s, %.1f Mpixel/s PCIe-inclusive"
              % (name, t_issue, t_all, gib / t_all, W * H / t_all / 1e6))

    raw16 = [np.frombuffer(rng.integers(-32768, 32767, W * H, dtype=np.int16).astype(">i2").tobytes(), np.uint8) for _ in range(4)]
    for rep in range(2):
        t0 = time.perf_counter()
        for k in range(N):
            st.upload_frame_fits(k, raw16[k % 4], 16, 1.0, 32768.0)
        st.run(2, 3.0, 3.0, fetch=False)
        t_all = time.perf_counter() - t0
    print("%-40s upload+decode+pass %.3f s  -> %.1f GiB/s of int16 payload, %.1f Mpixel/s PCIe-inclusive"
          % ("nl_stack_upload_frame_fits (BITPIX 16)", t_all, gib / 2 / t_all, W * H / t_all / 1e6))
