"""Cycle probes of the wave-per-pixel replay (library built with EXTRA=-DNL_PROBE, NLSTACK_LIB pointing at it):
    python tools/coop_probe.py <mode> <frames> <rows> [weighted 0/1]
prints the wave cycles per pixel spent in gather / decided select / undecided select + chains / clip / final."""
import ctypes
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightlight_amd import StackHandle, capi

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
weighted = (int(sys.argv[4]) if len(sys.argv) > 4 else 1) != 0
lib = capi.load()
lib.nl_debug_probe.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
buf = (ctypes.c_ulonglong * 8)()
with StackHandle(n, 4096, rows, device=0) as st:
    st.fill_synthetic(seed=1)
    if weighted:
        st.set_weights(np.array([0.2 + 0.8 * ((k * 37) % 101) / 100.0 for k in range(n)], np.float32))
    st.run_async(mode, 3.0, 3.0, 0.0); st.finish()
    lib.nl_debug_probe(buf, 1)
    st.run_async(mode, 3.0, 3.0, 0.0); st.finish()
    lib.nl_debug_probe(buf, 1)
    npix = 4096 * rows
    names = ["gather", "select(decided)", "select+chains(undecided)", "clip", "final", "whole wave", "whole wave (100 MHz ticks)"]
    print("mode %d frames %d rows %d weighted %d: %s  kernel %.3f ms %s" % (mode, n, rows, weighted,
          "  ".join("%s %.0f" % (nm, buf[i] / npix) for i, nm in enumerate(names)), st.last_kernel_ms, st.last_kernel_name))
    print("  average waves in flight %.0f, shader clock %.2f GHz, longest wave %.3f ms" % (buf[6] / 1e8 / (st.last_kernel_ms * 1e-3), buf[5] / (buf[6] / 1e8) / 1e9, buf[7] / 1e5))
