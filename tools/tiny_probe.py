"""Does the tail of a shallow winsorized pass come from the pixels with very few samples (NaN borders)?  (run on the GPU box)
The same 1024-row tile of the bench stack stacked as it is and with the border rows / columns cropped away."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightlight_amd import StackHandle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
W, H = 4096, 4096
def bench(st, tag):
    for _ in range(5): st.run_async(3, 3.0, 3.0, 0.0)
    st.finish()
    for _ in range(10): st.run_async(3, 3.0, 3.0, 0.0)
    cl, ch = st.finish()
    t = [st.pass_times(b) for b in range(10)]
    print("%s: pass %.3f ms dominant %.3f ms  generic %d exact %d" % (tag, np.mean([x[0] for x in t]), np.mean([x[1] for x in t]), st.last_generic_pixels, st.last_fallback_pixels))
with StackHandle(n, W, H, row0=0, rows=1032) as st:
    st.fill_synthetic(seed=1)
    bench(st, "with borders (rows 0..1031, all columns)")
    frames = [st.download_tile(i).reshape(1032, W) for i in range(n)]
crop = [np.ascontiguousarray(f[8:, :W - 8]) for f in frames]
h2, w2 = crop[0].shape
print("NaN fraction of the cropped stack: %.5f" % np.mean([np.isnan(c).mean() for c in crop]))
with StackHandle(n, w2, h2) as st:
    for i, c in enumerate(crop): st.upload_frame(i, c)
    bench(st, "cropped (%d x %d)" % (w2, h2))
