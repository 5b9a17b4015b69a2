"""Row-tile sharding across ranks on CPU: two gloo
processes, each owning one row tile of the same stack, exchange only the two
clip counters.  The tile runner here is the oracle (allowed in tests); on a GPU
rank it is nightlight_amd.StackHandle -- the sharding / reduction logic under
test (nightlight_amd/dist.py) is the same."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tile_rows_cover_the_image_exactly_once():
    from nightlight_amd.dist import tile_rows
    for height in (1, 7, 8, 4096, 4001):
        for world in (1, 2, 3, 8):
            spans = [tile_rows(height, world, r) for r in range(world)]
            assert spans[0][0] == 0
            for (r0, n), (r1, _) in zip(spans, spans[1:]):
                assert r0 + n == r1
            assert spans[-1][0] + spans[-1][1] == height
            sizes = [n for _, n in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        tile_rows(8, 2, 2)


def test_clipped_log_line_format():
    # stack.go:214-218, float32 arithmetic, %.2f
    from nightlight_amd.dist import clipped_log_line
    assert clipped_log_line(6836157, 13270993, 4096 * 4096, 128) == \
        "Clipped low 6836157 (0.32%) high 13270993 (0.62%)\n"


class OracleTile:
    """StackHandle-shaped tile runner backed by the CPU oracle."""

    def __init__(self, frames, width, row0, rows):
        from oracle import oracle
        self.o, self.width, self.row0, self.rows = oracle, width, row0, rows
        n = frames.shape[0]
        self.tile = np.ascontiguousarray(
            frames.reshape(n, -1, width)[:, row0:row0 + rows, :].reshape(n, -1))
        self.total = frames.size

    def run(self, mode, sl, sh, ref_loc=0.0, out=None, fetch=True):
        rc, res, cl, ch, _ = self.o.stack_apply(mode, self.tile, None, sl, sh, ref_loc)
        assert rc == 0
        if out is not None:
            out[self.row0 * self.width:(self.row0 + self.rows) * self.width] = res
        return res, cl, ch

    def find_sigmas(self, mode, perc_lo, perc_hi, ref_loc=0.0, reduce=None, fetch=True):
        # the bisection of stackfindsigma.go:48-98 with globally reduced counters
        lo_l, lo_r, hi_l, hi_r = 1.0, 11.0, 1.0, 11.0
        lo_m, hi_m = np.float32(6.0), np.float32(6.0)
        for i in range(100):
            res, cl, ch = self.run(mode, float(lo_m), float(hi_m), ref_loc)
            if reduce is not None:
                cl, ch = reduce(cl, ch)
            tot = np.float32(self.total)
            pl = np.float32(cl) * np.float32(100) / tot
            ph = np.float32(ch) * np.float32(100) / tot
            dl = int(np.float32(100) * pl + np.float32(0.5)) - int(100 * perc_lo)
            dh = int(np.float32(100) * ph + np.float32(0.5)) - int(100 * perc_hi)
            if (dl == 0 and dh == 0) or i >= 20:
                return res, cl, ch, float(lo_m), float(hi_m), i + 1
            if dl > 0:
                lo_l = float(lo_m)
            elif dl < 0:
                lo_r = float(lo_m)
            lo_m = np.float32(0.5) * (np.float32(lo_l) + np.float32(lo_r))
            if dh > 0:
                hi_l = float(hi_m)
            elif dh < 0:
                hi_r = float(hi_m)
            hi_m = np.float32(0.5) * (np.float32(hi_l) + np.float32(hi_r))

    def close(self):
        pass


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from nightlight_amd.dist import ShardedStack
    from util import make_frames
    dist.init_process_group("gloo", rank=rank, world_size=world)
    width, height, n = 48, 22, 18
    frames = make_frames(n, width, height, seed=4242)
    sh = ShardedStack(height, lambda r0, rows: OracleTile(frames, width, r0, rows),
                      world=world, rank=rank, device="cpu")
    res, cl, ch = sh.run(3, 2.5, 2.5)
    gs = sh.find_sigmas(2, 1.0, 1.0)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), res=res, row0=sh.row0, rows=sh.rows,
             clip=np.array([cl, ch]), gs_res=gs[0], gs=np.array(gs[1:], np.float64))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_reassemble_the_single_rank_result(tmp_path, oracle):
    import torch.multiprocessing as mp
    from util import make_frames
    world, port = 2, 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    width, height, n = 48, 22, 18
    frames = make_frames(n, width, height, seed=4242)
    rc, want, wl, wh, _ = oracle.stack_apply(3, frames, None, 2.5, 2.5)
    passes, gres, gcl, gch, gsl, gsh = oracle.find_sigmas_bisect(2, frames, 1.0, 1.0)
    full = np.zeros(width * height, np.float32)
    gfull = np.zeros(width * height, np.float32)
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        r0, rows = int(d["row0"]), int(d["rows"])
        full[r0 * width:(r0 + rows) * width] = d["res"]
        gfull[r0 * width:(r0 + rows) * width] = d["gs_res"]
        assert tuple(d["clip"]) == (wl, wh)          # every rank holds the GLOBAL totals
        assert tuple(d["gs"][:2]) == (gcl, gch) and int(d["gs"][4]) == passes
        assert (np.float32(d["gs"][2]), np.float32(d["gs"][3])) == (gsl, gsh)
    assert np.array_equal(full, want, equal_nan=True)
    assert np.array_equal(gfull, gres, equal_nan=True)
