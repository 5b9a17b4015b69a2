"""Host-side simulation behind two design decisions of the bit-exact replay; needs numpy only:
    python tools/qselect_sim.py [frames]
(1) a LANE-per-pixel replay runs 64 quickselects (qsort.go:94-126) in lock step, each pointer advancing up to `cand`
    positions per step: prints the steps of the mean lane and of the slowest of 64 lanes -- the wave pays for the latter;
(2) how many partition passes a select needs and on what range sizes (a wave-per-pixel pass costs the same whatever
    its range: two registers per lane from 64 elements on, one below)."""
import sys
import numpy as np

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rng = np.random.default_rng(1)


def select(a, k, cand):
    a = a.copy()
    left, right = 0, len(a) - 1
    steps, sizes = 0, []
    while left < right:
        sizes.append(right - left + 1)
        pivot = a[(left + right) >> 1]
        l, r = left - 1, right + 1
        lstop = rstop = False
        while True:
            steps += 1
            if not lstop:
                for _ in range(cand):
                    l += 1
                    if a[l] >= pivot or l >= right:
                        lstop = True
                        break
            if not rstop:
                for _ in range(cand):
                    r -= 1
                    if a[r] <= pivot or r <= left:
                        rstop = True
                        break
            if lstop and rstop:
                if l >= r:
                    break
                a[l], a[r] = a[r], a[l]
                lstop = rstop = False
        off = r - left + 1
        if k <= off:
            right = r
        else:
            left, k = r + 1, k - off
    return steps, sizes


for cand in (1, 2, 4):
    st = np.array([select(rng.normal(size=N).astype(np.float32), N // 2 + 1, cand)[0] for _ in range(64 * 20)]).reshape(20, 64)
    print("%d frames, %d candidates per pointer and step: %.0f steps per select on average, %.0f for the slowest of 64 lanes"
          % (N, cand, st.mean(), st.max(1).mean()))
sizes = []
for _ in range(3000):
    sizes += select(rng.normal(size=N).astype(np.float32), N // 2 + 1, 1)[1]
sizes = np.array(sizes)
print("partition passes per select: %.2f" % (len(sizes) / 3000.0))
for lo, hi in ((2, 2), (3, 4), (5, 16), (17, 63), (64, N)):
    print("  on ranges of %3d ... %3d elements: %.2f" % (lo, hi, ((sizes >= lo) & (sizes <= hi)).sum() / 3000.0))
