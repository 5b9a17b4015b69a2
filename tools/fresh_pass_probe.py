"""What the FIRST pass of a fresh handle costs, host and device view (run on the GPU box): a handle per iteration on borrowed
frames (as bench.py's fresh_handle), one synchronous pass, then a second one.   python tools/fresh_pass_probe.py [frames]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightlight_amd import StackHandle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
with StackHandle(n, 4096, 4096, device=0) as st:
    st.fill_synthetic(seed=1)
    for _ in range(30):
        st.run_async(2, 3.0, 3.0, 0.0)
    st.finish()
    rows = []
    for it in range(12):
        h2 = StackHandle(n, 4096, 4096, device=0)
        h2.attach_device_frames(st.frames_device_ptr(), st.frame_stride())
        t0 = time.perf_counter(); h2.run_async(2, 3.0, 3.0, 0.0); t1 = time.perf_counter(); h2.finish(); t2 = time.perf_counter()
        p1 = h2.pass_times(0); proto1 = h2.last_pass_protocol
        t3 = time.perf_counter(); h2.run_async(2, 3.0, 3.0, 0.0); h2.finish(); t4 = time.perf_counter()
        p2 = h2.pass_times(0); proto2 = h2.last_pass_protocol
        h2.attach_device_frames(None); h2.close()
        rows.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, p1[0], p1[1], proto1, (t4 - t3) * 1e3, p2[0], p2[1], proto2))
    r = np.array([[x for x in row] for row in rows[2:]], dtype=np.float64)
    m = np.median(r, axis=0)
    print("frames %d: first pass enqueue %.3f + finish %.3f ms (device: pass %.3f, dominant kernel %.3f, protocol %d); second pass %.3f ms (device %.3f / %.3f, protocol %d)"
          % (n, m[0], m[1], m[2], m[3], int(m[4]), m[5], m[6], m[7], int(m[8])))
    st.run_async(2, 3.0, 3.0, 0.0); st.finish()
    print("  the long-lived handle, synchronous: device pass %.3f / dominant %.3f" % st.pass_times(0))
