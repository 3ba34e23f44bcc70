"""Data-parallel calibration helpers (new functionality: the reference is single-process).

One process per GPU; rank r owns calibration images [lo, hi) and their caches for every unit; per iteration the flat
alpha-gradient bucket of the unit is summed across ranks (RCCL all-reduce over xGMI; gloo in the CPU tests) and applied with
scale 1/world_size.  Nothing else moves between ranks."""
import torch
import torch.distributed as dist


def world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_range(n, rank, world_size):
    """Contiguous shard [lo, hi) of n calibration images for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard(cali_data, group=None):
    r, w = world(group)
    lo, hi = shard_range(cali_data.shape[0], r, w)
    return cali_data[lo:hi]


def allreduce_mean_(tensors, group=None):
    """In-place mean of a list of gradient tensors across ranks through ONE flat bucket (a single collective)."""
    _, w = world(group)
    if w == 1:
        return tensors
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat /= w
    off = 0
    for t in tensors:
        t.copy_(flat[off:off + t.numel()].view_as(t))
        off += t.numel()
    return tensors
