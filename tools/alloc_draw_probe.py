#!/usr/bin/env python3
"""Does the pass time of the LIBRARY depend on where a handle's frame buffer landed?  (run on the GPU box, NL_MEM_CACHE_MB=0;
the draw-at-create experiment of DESIGN 12.8 used a library build with nl_stack_frame_buffer_draws, not kept)
Eight handles of the headline geometry created in a row and kept alive (so that every one gets other memory); on each: mean,
median and sigma-clip passes, dominant-kernel ms (HIP events, mean of 10 after 5 warm-ups).  A bimodal column = the allocation
lottery of DESIGN 11.9 seen through the product kernels."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightlight_amd import StackHandle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
count = int(sys.argv[2]) if len(sys.argv) > 2 else 8
hs = []
for i in range(count):
    st = StackHandle(n, 4096, 4096)
    st.fill_synthetic(seed=1)
    hs.append(st)
for rnd in range(3):                       # interleaved rounds: is a handle's time its own, or the moment's?
    for i, st in enumerate(hs):
        row = []
        for mode in (1, 0, 2):
            for _ in range(5):
                st.run_async(mode, 3.0, 3.0, 0.0)
            st.finish()
            for _ in range(10):
                st.run_async(mode, 3.0, 3.0, 0.0)
            st.finish()
            t = [st.pass_times(b) for b in range(10)]
            row.append((float(np.mean([x[1] for x in t])), float(np.mean([x[0] for x in t]))))
        print("round %d handle %d frames at 0x%x: mean kernel %.4f ms (pass %.4f) | median %.4f (%.4f) | sigma %.4f (%.4f)"
              % (rnd, i, st.frames_device_ptr(), row[0][0], row[0][1], row[1][0], row[1][1], row[2][0], row[2][1]), flush=True)
for st in hs:
    st.close()
