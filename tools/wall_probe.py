"""Wall-clock time of queued passes with a developer switch on and off, interleaved (run on the GPU box):
    python tools/wall_probe.py <mode> <frames> <rows> <flags>      (e.g. flags 32: passes without their timing events)"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightlight_amd import StackHandle

mode, n, rows, flags = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
with StackHandle(n, 4096, 4096, device=0, row0=0 if rows == 4096 else 1536, rows=rows) as st:
    st.fill_synthetic(seed=1)
    for _ in range(100):
        st.run_async(mode, 3.0, 3.0, 0.0)
    st.finish()
    res = {0: [], flags: []}
    for rep in range(5):
        for fl in (0, flags):
            st.set_dev_flags(fl)
            for _ in range(10):
                st.run_async(mode, 3.0, 3.0, 0.0)
            st.finish()
            t0 = time.perf_counter()
            for _ in range(100):
                st.run_async(mode, 3.0, 3.0, 0.0)
            st.finish()
            res[fl].append((time.perf_counter() - t0) * 10.0)
    for fl in res:
        print("mode %d frames %d rows %d flags %d: %.4f ms per pass (median of 5 x 100 queued passes; min %.4f)"
              % (mode, n, rows, fl, float(np.median(res[fl])), min(res[fl])))
