"""Host time of one bench step of a rank (run on the GPU box): enqueue only, parts timed apart.
    python tools/host_step_probe.py [frames] [rows]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from nightlight_amd import StackHandle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 512
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
dist.init_process_group("nccl", rank=0, world_size=1)
with StackHandle(n, 4096, 4096, device=0, row0=1536, rows=rows) as st:
    st.fill_synthetic(seed=1)
    ring = torch.zeros((3, 4), dtype=torch.int64, device="cuda")
    comm = torch.cuda.Stream()
    for i in range(20):
        st.set_counters_buffer(ring[i % 3].data_ptr())
        st.run_async(2, 3.0, 3.0, 0.0)
        st.order_stream_after(comm.cuda_stream)
        with torch.cuda.stream(comm):
            dist.all_reduce(ring[i % 3][:2], async_op=True).wait()
    comm.synchronize()
    st.finish()
    t = {"set_buffer": 0.0, "run_async": 0.0, "order": 0.0, "all_reduce": 0.0}
    works = []
    reps = 200
    t_all = time.perf_counter()
    for i in range(reps):
        a = time.perf_counter()
        st.set_counters_buffer(ring[i % 3].data_ptr())
        b = time.perf_counter()
        st.run_async(2, 3.0, 3.0, 0.0)
        c = time.perf_counter()
        st.order_stream_after(comm.cuda_stream)
        d = time.perf_counter()
        with torch.cuda.stream(comm):
            works.append(dist.all_reduce(ring[i % 3][:2], async_op=True))
        e = time.perf_counter()
        t["set_buffer"] += b - a; t["run_async"] += c - b; t["order"] += d - c; t["all_reduce"] += e - d
    enq = time.perf_counter() - t_all
    with torch.cuda.stream(comm):
        for w in works:
            w.wait()
    comm.synchronize()
    st.finish()
    torch.cuda.synchronize()
    total = time.perf_counter() - t_all
    print("frames %d rows %d: host enqueue %.1f us per step (%s), wall %.1f us per step"
          % (n, rows, enq / reps * 1e6, ", ".join("%s %.1f" % (k, v / reps * 1e6) for k, v in t.items()), total / reps * 1e6))
dist.destroy_process_group()
