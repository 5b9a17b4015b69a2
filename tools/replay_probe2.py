"""Dense bit-exact replay engines side by side (run on the GPU box):
    python tools/replay_probe2.py <mode> <frames> <rows> [weighted 0/1]
times nl_stack_set_exact(h, 2) (one pixel per wave), (h, 4) (four pixels per wave on 16-lane rows) and, up to 64
frames, (h, 3) (64 pixels per wave, lane per pixel) over the whole tile."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightlight_amd import StackHandle

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
weighted = (int(sys.argv[4]) if len(sys.argv) > 4 else 1) != 0
with StackHandle(n, 4096, rows, device=0) as st:
    st.fill_synthetic(seed=1)
    if weighted:
        st.set_weights(np.array([0.2 + 0.8 * ((k * 37) % 101) / 100.0 for k in range(n)], np.float32))
    ref = None
    for flavour in (2, 4, 3):
        if flavour == 3 and n > 64:
            continue
        st.set_exact(flavour)
        t = []
        for _ in range(4):
            st.run_async(mode, 3.0, 3.0, 0.0)
            cl, ch = st.finish()
            t.append(st.last_kernel_ms)
        out = st.result_tile()
        if ref is None:
            ref = (out.copy(), cl, ch)
        same = np.array_equal(out.view(np.uint32), ref[0].view(np.uint32)) and (cl, ch) == ref[1:]
        print("mode %d frames %d rows %d weighted %d  exact flavour %d: %.3f ms (%s)  %s  identical to flavour 2: %s"
              % (mode, n, rows, weighted, flavour, sorted(t)[1], st.last_kernel_name, (cl, ch), same))
