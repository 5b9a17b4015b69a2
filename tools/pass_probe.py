"""Hand-over statistics of one pass on the bench stack (run on the GPU box):
    python tools/pass_probe.py <mode> <frames> <rows> [row0] [image_rows]
prints pixels sent to the generic pass / the exact replay, and the pass / dominant-kernel times."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightlight_amd import StackHandle

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
row0 = int(sys.argv[4]) if len(sys.argv) > 4 else 0
image_rows = int(sys.argv[5]) if len(sys.argv) > 5 else rows
with StackHandle(n, 4096, image_rows, device=0, row0=row0, rows=rows) as st:
    st.fill_synthetic(seed=1)
    for _ in range(3):
        got, cl, ch = st.run(mode, 3.0, 3.0)
    print("mode %d frames %d rows %d(+%d of %d): %s  generic %d  exact %d  pass %.3f ms (dominant %.3f)  clips %d / %d"
          % (mode, n, rows, row0, image_rows, st.last_kernel_name, st.last_generic_pixels, st.last_fallback_pixels,
             st.last_kernel_ms, st.last_dominant_kernel_ms, cl, ch))
