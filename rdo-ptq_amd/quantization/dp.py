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


class GradBucket:
    """The flat alpha-gradient bucket of one reconstruction unit and its collective.

    One persistent fp32 tensor holds d(rec + task)/d(alpha) of every weight tensor of the unit, back to back; `late` names the
    tensor whose weight gradient is the LAST kernel of the unit's backward pass -- it sits at the END of the bucket, so that the
    front (complete before that kernel starts) can be all-reduced while it runs.  The two halves are views made ONCE (the collective
    is always issued on the same tensors: no per-call slicing, a stable registration for RCCL).  SUM over the ranks; the consumer
    applies `scale` = 1 / world (the AdaRound apply kernel multiplies it in; the round-loss gradient is data independent and is
    added locally afterwards).  Used by `engine.UnitEngine` on device tensors over RCCL and by the CPU tests over gloo."""

    def __init__(self, sizes, late=None, device=None, group=None):
        names = list(sizes)
        if late is not None and late not in sizes:
            raise KeyError(f"gradient bucket: '{late}' is not a tensor of this unit ({names})")
        self.order = [n for n in names if n != late] + ([late] if late is not None else [])
        self.late, self.group = late, group
        total = sum(int(sizes[n]) for n in names)
        self.flat = torch.zeros(total, device=device, dtype=torch.float32)
        self.views, off = {}, 0
        for n in self.order:
            self.views[n] = self.flat[off:off + int(sizes[n])]
            off += int(sizes[n])
        self.early_numel = total - (int(sizes[late]) if late is not None else 0)
        self.front = self.flat[:self.early_numel] if late is not None else self.flat
        self.back = self.flat[self.early_numel:] if late is not None else None
        self.world = world(group)[1]
        self.scale = 1.0 / self.world
        self.n_collectives = 0              # issued so far (bench / tests)

    def view(self, name):
        return self.views[name]

    @property
    def comm(self):
        """Is there a process group to talk to?  (A forced split on one rank without one runs the same sequence, no collective.)"""
        return self.world > 1 or (dist.is_available() and dist.is_initialized())

    def reduce(self, between=None):
        """SUM the bucket over the ranks.  Without `late`: one all-reduce.  With it: all-reduce of the front (asynchronous, on the
        process group's stream), `between()` -- the caller enqueues the unit's last weight gradient --, all-reduce of the back, wait
        for both.  `between` is always called (exactly once) when given, so the caller's sequence does not depend on the world."""
        if self.back is None:
            if between is not None:
                between()
            if self.comm:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
                self.n_collectives += 1
            return
        w1 = dist.all_reduce(self.front, op=dist.ReduceOp.SUM, group=self.group, async_op=True) if self.comm else None
        if between is not None:
            between()
        if self.comm:
            w2 = dist.all_reduce(self.back, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            w1.wait()
            w2.wait()
            self.n_collectives += 2

    def nbytes(self):
        return self.flat.numel() * 4
