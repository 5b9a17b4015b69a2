"""Row F2 end to end (run on the GPU box): OpStackBatches over frames that start in HOST memory.
    python tools/batches_probe.py [frames] [batch] [width] [height]
Measures, for `frames` sub-exposures stacked in batches of `batch` (sigma clip, then the frame-count-weighted
stack of stacks, stackbatches.go:68-116):
  A  the C++ OpStackBatches mirror (nl_host_op_stack_batches_apply_json) on fp32 frames -- wall clock;
  B  the same pipeline through the C ABI on int16 FITS payloads (nl_group_upload_frame_fits: half the bytes
     cross PCIe, decode on the device), batch b+1 staged while batch b is stacked;
  C  the same on fp32 frames (nl_group_upload_frame);
  D  the resident pass alone (frames already in HBM), for comparison.
Prints PCIe-inclusive Mpixel/s (stacked output pixels x batches / wall), the sustained host->device rate and
how much of the wall clock the stack passes account for."""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightlight_amd import StackGroup
from nightlight_amd import operator as op

n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
w = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
h = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
nb = n // batch
rng = np.random.default_rng(3)
distinct = 8
f32 = [(1000 + 30 * rng.standard_normal(w * h)).astype(np.float32) for _ in range(distinct)]
i16 = [np.frombuffer((f - 32768).astype(">i2").tobytes(), np.uint8) for f in f32]     # BZERO = 32768 payloads
frames = [f32[i % distinct] for i in range(n)]
gib = 1024.0 ** 3

# A: the operator mirror (stack memory sized so that `batch` frames fit a batch: stackbatches.go:139-186)
# (batch size = available frames - threads - 2: reference frame and stack of stacks)
mem_mb = (batch + 4 + 2) * (w * h * 4 // (1 << 20))
t0 = time.perf_counter()
out, exp_sum, log = op.op_stack_batches_apply_json('{"type":"stack","mode":2,"sigmaLow":3,"sigmaHigh":3}', frames, w, h,
                                                   exposure=np.ones(n, np.float32), max_threads=4,
                                                   stack_memory_mb=mem_mb, memory_mb=4 * mem_mb)
ta = time.perf_counter() - t0
line = [l for l in log.splitlines() if "Using" in l]
# what materializing the input promises costs on its own: the mirror's promise allocates and fills a fresh
# 64 MiB Image per frame (the stand-in for loading + pre-processing a FITS file), 4 at a time
t0 = time.perf_counter()
for i in range(n):
    _ = frames[i].copy()
tm = time.perf_counter() - t0
print("A  OpStackBatches mirror, fp32 host frames: %.3f s wall for %d frames (%s)  %.1f GiB/s  %.1f stacked Mpixel/s, "
      "%.0f input Msamples/s; materializing %d fresh 64 MiB frames alone (single thread): %.3f s"
      % (ta, n, line[0].strip() if line else "?", n * w * h * 4 / gib / ta, w * h / ta / 1e6, w * h * n / ta / 1e6, n, tm))


def pipeline(kind):
    def upload(g, k, j):
        if kind == "i16":
            g.upload_frame_fits(k, i16[j % distinct], 16, 1.0, 32768.0)
        else:
            g.upload_frame(k, f32[j % distinct])

    with StackGroup(batch, w, h, n_tiles=1, devices=[0]) as g:
        for k in range(batch):                       # warm the ring and the kernels
            upload(g, k, k)
        g.run(2, 3.0, 3.0, download=False)
        t0 = time.perf_counter()
        t_pass = 0.0
        for b in range(nb):
            for k in range(batch):
                upload(g, k, b * batch + k)
            t1 = time.perf_counter()
            g.run(2, 3.0, 3.0, download=False)       # waits for the uploads on the device, then the pass
            g.accumulate(float(batch), first=(b == 0))
            t_pass += time.perf_counter() - t1
        res = g.accumulate_finalize(float(nb * batch))
        wall = time.perf_counter() - t0
        t_res = []
        for _ in range(5):
            t1 = time.perf_counter()
            g.run(2, 3.0, 3.0, download=False)
            t_res.append(time.perf_counter() - t1)
    return wall, t_pass, sorted(t_res)[2], res


for kind, label, bpp in (("i16", "B  int16 FITS payloads -> nl_group_upload_frame_fits", 2), ("f32", "C  fp32 frames -> nl_group_upload_frame", 4)):
    wall, t_pass, t_res, res = pipeline(kind)
    print("%s: %.3f s wall for %d frames in %d batches  %.1f GiB/s over PCIe  %.0f stacked Mpixel/s PCIe-inclusive (%.0f input Msamples/s); "
          "host waited %.3f s in run + accumulate (the tail of the uploads included); resident pass %.2f ms"
          % (label, wall, nb * batch, nb, nb * batch * w * h * bpp / gib / wall, w * h / wall / 1e6, w * h * nb * batch / wall / 1e6,
             t_pass, t_res * 1e3))
