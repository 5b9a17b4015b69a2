"""Row-tile sharding of one stack across the GPUs of a node.

Every output pixel depends only on the same pixel of the N frames
(internal/ops/stack/stack.go:142-152 already splits by contiguous pixel
ranges), so rank g of G owns rows [g*H/G, (g+1)*H/G) of ALL frames and no
pixel ever crosses GPUs.  The only exchange is the sum of the two clip
counters {clipLow, clipHigh} per pass (stack.go:193-198, and every goal-seek
step of stackfindsigma.go:56-97): one 16-byte all-reduce -- RCCL over xGMI
with backend "nccl", gloo on CPU in the tests.
"""
import numpy as np


def tile_rows(height, world, rank):
    """Rows [row0, row0+rows) owned by `rank`: contiguous, sizes differ by at
    most one row, all rows covered exactly once."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank %d of %d" % (rank, world))
    base, extra = divmod(int(height), int(world))
    rows = base + (1 if rank < extra else 0)
    row0 = rank * base + min(rank, extra)
    return row0, rows


def allreduce_counters(clip_low, clip_high, group=None, device=None):
    """Sum of (clipLow, clipHigh) over all ranks; identity without a process group."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return int(clip_low), int(clip_high)
    if device is None:
        device = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor([int(clip_low), int(clip_high)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    lo, hi = t.tolist()
    return int(lo), int(hi)


def clip_percentages(clip_low, clip_high, pixels, n_frames):
    """The two percentages of the reference's log line (stack.go:214-218),
    in its fp32 arithmetic: float32(clip)*100/float32(pixels*frames)."""
    total = np.float32(int(pixels) * int(n_frames))
    lo = np.float32(clip_low) * np.float32(100.0) / total
    hi = np.float32(clip_high) * np.float32(100.0) / total
    return float(lo), float(hi)


def clipped_log_line(clip_low, clip_high, pixels, n_frames):
    lo, hi = clip_percentages(clip_low, clip_high, pixels, n_frames)
    return "Clipped low %d (%.2f%%) high %d (%.2f%%)\n" % (clip_low, lo, clip_high, hi)


class ShardedStack:
    """One rank's share of a stack: a tile runner plus the counter all-reduce.

    `make_tile(row0, rows)` returns an object with the StackHandle interface
    (run / find_sigmas / close): nightlight_amd.StackHandle on a GPU rank.
    """

    def __init__(self, height, make_tile, world=1, rank=0, group=None, device=None):
        self.world, self.rank, self.group, self.device = world, rank, group, device
        self.row0, self.rows = tile_rows(height, world, rank)
        self.tile = make_tile(self.row0, self.rows)

    def run(self, mode, sigma_low, sigma_high, ref_loc=0.0, out=None, fetch=True):
        """One pass over this rank's tile; returns (result, global clipLow, global clipHigh)."""
        res, cl, ch = self.tile.run(mode, sigma_low, sigma_high, ref_loc, out=out, fetch=fetch)
        cl, ch = allreduce_counters(cl, ch, self.group, self.device)
        return res, cl, ch

    def find_sigmas(self, mode, clip_perc_low, clip_perc_high, ref_loc=0.0, fetch=True):
        """Goal-seek with globally reduced counters: every rank takes the same
        bisection branch without a host round trip through rank 0."""
        return self.tile.find_sigmas(
            mode, clip_perc_low, clip_perc_high, ref_loc,
            reduce=lambda lo, hi: allreduce_counters(lo, hi, self.group, self.device),
            fetch=fetch)

    def close(self):
        self.tile.close()
