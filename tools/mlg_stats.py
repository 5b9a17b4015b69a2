"""Developer probe (stats build of stack_fast_mlg.hip, -DNL_ROUND_STATS, NLSTACK_LIB pointing at it): cycles a wave
of the LDS generic pass spends in gather + sort, LDS column + tables, and rounds.
usage: mlg_stats.py <mode> <frames> <rows> [row0 image_rows]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightlight_amd import StackHandle, capi
mode, n, rows = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
row0 = int(sys.argv[4]) if len(sys.argv) > 4 else 0
img = int(sys.argv[5]) if len(sys.argv) > 5 else rows
lib = ctypes.CDLL(capi.LIB_PATH)
out = (ctypes.c_ulonglong * 8)()
with StackHandle(n, 4096, img, row0=row0, rows=rows) as st:
    st.fill_synthetic(seed=1)
    st.run(mode, 3.0, 3.0)
    lib.nl_debug_round_stats_mlg(out, 1)
    st.run(mode, 3.0, 3.0)
    lib.nl_debug_round_stats_mlg(out, 1)
    v = list(out); trips = max(v[4], 1)
    print("mode %d n %d rows %d: generic %d px; wave trips %d; rounds/trip %.1f passes/trip %.1f; cycles per trip: gather+sort %.0f, column+tables %.0f, rounds %.0f (100 MHz counter?)"
          % (mode, n, rows, st.last_generic_pixels, trips, v[0] / trips, v[2] / trips, v[5] / trips, v[6] / trips, v[7] / trips))
