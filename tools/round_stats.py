#!/usr/bin/env python3
"""Developer probe: winsorization rounds / clip passes executed per wave vs needed per lane.
Needs a library built with  make EXTRA=-DNL_ROUND_STATS  (stack_fast.hip / stack_fast_ml.hip)."""
import ctypes
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightlight_amd import StackHandle, capi

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
h = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
flags = int(sys.argv[4]) if len(sys.argv) > 4 else 0          # developer switches (nl_stack_set_dev_flags)
lib = ctypes.CDLL(capi.LIB_PATH)
out = (ctypes.c_ulonglong * 16)()
with StackHandle(n, 4096, h, device=0) as st:
    st.fill_synthetic(seed=0x4E4C5354)
    st.set_dev_flags(flags)
    st.run(mode, 3.0, 3.0)
    name = "nl_debug_round_stats_ml" if n > 128 else "nl_debug_round_stats"
    fn = getattr(lib, name) if hasattr(lib, name) else (lambda o, r: 0)
    fn(out, 1)
    st.run(mode, 3.0, 3.0)
    fn(out, 1)
    v = list(out)
    waves = max(v[4], 1)
    lanes = 4096 * h                      # pixels (multi-lane kernels count one lane per pixel)
    print("mode %d n %d: kernel %s %.3f ms" % (mode, n, st.last_kernel_name, st.last_kernel_ms))
    print("  waves %d; winsor rounds per wave %.2f, per lane %.2f; clip passes per wave %.2f, per lane %.2f"
          % (waves, v[0] / waves, v[1] / lanes, v[2] / waves, v[3] / lanes))
    if n > 128 and hasattr(lib, "nl_debug_round_stats_mlz"):
        lib.nl_debug_round_stats_mlz(out, 1)
        print("  LDS-column kernel hand-overs: missing %d, c2>=8 %d, d2>=8 %d, low zone %d, high zone %d; shape -> exact %d; winsor bail %d, guard>100 %d" % tuple(list(out)[:8]))
        o = list(out)
        if o[12]:
            print("  sorting phase, cycles per block (wave 0): gather %.0f, network %.0f, ends %.0f, moments + window %.0f; rounds %.0f per block (%d)"
                  % (o[8] / o[12], o[9] / o[12], o[10] / o[12], o[11] / o[12], o[6] / max(o[7], 1), o[7]))
    if n > 128 and hasattr(lib, "nl_debug_round_stats_mlg"):
        lib.nl_debug_round_stats_mlg(out, 1)
        g = list(out)
        trips = max(g[4], 1)
        print("  generic pass: %d wave trips; rounds per trip %.2f (lanes %d); passes per trip %.2f (lanes %d)"
              % (trips, g[0] / trips, g[1], g[2] / trips, g[3]))
