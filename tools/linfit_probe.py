"""Linear-fit cascade statistics on the bench stack (run on the GPU box):
    python tools/linfit_probe.py [frames] [rows]
prints the list lengths of the cascade stages and the pass time."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightlight_amd import StackHandle, ST_LINEAR_FIT

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
with StackHandle(n, 4096, rows, device=0) as st:
    st.fill_synthetic(seed=1)
    for _ in range(3):
        got, cl, ch = st.run(ST_LINEAR_FIT, 3.0, 3.0)
    c = st.linfit_stage_counts
    print("pixels %d  stage hand-overs %r  exact %d  pass %.3f ms (dominant %.3f)  clips %d / %d"
          % (4096 * rows, c, st.last_fallback_pixels, st.last_kernel_ms, st.last_dominant_kernel_ms, cl, ch))
