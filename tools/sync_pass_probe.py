"""Host view of one synchronous pass, as OpStack.Apply runs it (run on the GPU box): enqueue, wait, counters back.
    python tools/sync_pass_probe.py [frames] [rows]"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightlight_amd import StackHandle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
with StackHandle(n, 4096, 4096, device=0, row0=0 if rows == 4096 else 1536, rows=rows) as st:
    st.fill_synthetic(seed=1)
    for _ in range(20):
        st.run_async(2, 3.0, 3.0, 0.0)
    st.finish()
    enq, fin, gpu = [], [], []
    for _ in range(50):
        t0 = time.perf_counter()
        st.run_async(2, 3.0, 3.0, 0.0)
        t1 = time.perf_counter()
        st.finish()
        t2 = time.perf_counter()
        enq.append((t1 - t0) * 1e3); fin.append((t2 - t1) * 1e3); gpu.append(st.pass_times(0)[0])
    print("frames %d rows %d: synchronous pass %.4f ms = enqueue %.4f + finish %.4f; the pass on the device %.4f (medians of 50)"
          % (n, rows, np.median(enq) + np.median(fin), np.median(enq), np.median(fin), np.median(gpu)))
