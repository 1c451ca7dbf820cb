"""Multi-GPU: rays shard embarrassingly by contiguous index ranges, one process per GPU (torch.distributed,
backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests).  Every rank keeps its shard resident for the whole
chain; the only exchange is at the detector: one all-reduce of 16 statistics (mean path, bounding box, weights)
and ONE gather of the per-ray read-out (X, Y, optical path, alive) to rank 0 -- a gather, not a ring collective:
on the xGMI mesh every peer has its own link into the root."""
import numpy as np
import torch
import torch.distributed as dist

# slots of the read-out statistics (include/art_hip.h: art_detector_stats [0..15], art_detector_readout [16..23]) by
# reduction operator
_MIN = [2, 4, 12]
_MAX = [3, 5, 13]


def shard_range(n_total, rank, world):
    """Contiguous global index range [lo, hi) of `rank`; concatenating ranks restores the global ray order."""
    return (n_total * rank) // world, (n_total * (rank + 1)) // world


class PendingStats:
    """Handle of an in-flight statistics exchange (allreduce_stats(..., async_op=True)): `.result()` waits for the
    collective (on the caller's stream, not the host) and folds the per-rank vectors."""

    def __init__(self, work, flat, world, nslots):
        self.work, self.flat, self.world, self.nslots = work, flat, world, nslots

    def result(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        allv = self.flat.view(self.world, self.nslots)
        out = allv.sum(dim=0)
        out[_MIN] = allv[:, _MIN].min(dim=0).values
        out[_MAX] = allv[:, _MAX].max(dim=0).values
        return out


def allreduce_stats(stats, device, async_op=False):
    """Combine per-shard read-out statistics (16 or 24 slots, layout of include/art_hip.h) into the global ones.
    ONE collective: an all-gather of the small vectors, folded on the device (sum slots added, min/max slots
    min/max-ed) -- cheaper than one all-reduce per operator.  Accepts a host array or a device tensor; returns a
    tensor on `device` (or, with async_op=True, a PendingStats whose collective runs on RCCL's stream while the
    caller's stream goes on tracing); nothing blocks the host."""
    t = stats if torch.is_tensor(stats) else torch.as_tensor(np.asarray(stats, dtype=np.float64))
    t = t.to(device)
    if not (dist.is_available() and dist.is_initialized()):
        return t
    world = dist.get_world_size()
    flat = torch.empty(world * t.numel(), dtype=torch.float64, device=device)
    work = dist.all_gather_into_tensor(flat, t.contiguous().reshape(-1), async_op=async_op)
    pending = PendingStats(work if async_op else None, flat, world, t.numel())
    return pending if async_op else pending.result()


def gather_readout(X, Y, opl, alive, dst=0, pack=None, sizes=None, async_op=False):
    """Gather the detector read-out of every shard to rank `dst` (rank order = global ray order).
    Returns (XYO [3, n_total] float64, alive [n_total] uint8) on dst, (None, None) elsewhere.
    Shards may differ in length (index ranges of a ray count not divisible by the world size): every rank sends
    a block padded to the longest shard and the root trims.  `sizes` = list of shard lengths if already known
    (otherwise one tiny all-gather); `pack` may hold preallocated {'send': [3,nmax], 'asend': [nmax],
    'recv': [world x [3,nmax]], 'arecv': [world x [nmax]]} buffers.
    With async_op=True (needs `pack` and `sizes`) the two collectives are only enqueued and a list of work handles
    is returned: the caller overlaps them with further kernels and calls `.wait()` on each handle before it
    touches the pack's buffers again (the result then sits in pack['recv'] / pack['arecv'] on dst)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if not dist.is_initialized():
        return torch.stack([X, Y, opl]), alive
    rank = dist.get_rank()
    n = X.numel()
    if sizes is None:
        t = torch.tensor([n], dtype=torch.int64, device=X.device)
        allsz = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allsz, t)
        sizes = [int(v.item()) for v in allsz]
    nmax = max(sizes)
    send = pack["send"] if pack else torch.zeros((3, nmax), dtype=torch.float64, device=X.device)
    asend = pack["asend"] if pack else torch.zeros(nmax, dtype=torch.uint8, device=X.device)
    send[0, :n], send[1, :n], send[2, :n] = X, Y, opl
    asend[:n] = alive
    if async_op:
        recv = pack["recv"] if rank == dst else None
        arecv = pack["arecv"] if rank == dst else None
        return [dist.gather(send, recv, dst=dst, async_op=True), dist.gather(asend, arecv, dst=dst, async_op=True)]
    if rank == dst:
        recv = pack["recv"] if pack else [torch.empty_like(send) for _ in range(world)]
        arecv = pack["arecv"] if pack else [torch.empty_like(asend) for _ in range(world)]
        dist.gather(send, recv, dst=dst)
        dist.gather(asend, arecv, dst=dst)
        if all(sz == nmax for sz in sizes):
            return torch.cat(recv, dim=1), torch.cat(arecv)
        return (torch.cat([r[:, :sz] for r, sz in zip(recv, sizes)], dim=1),
                torch.cat([a[:sz] for a, sz in zip(arecv, sizes)]))
    dist.gather(send, None, dst=dst)
    dist.gather(asend, None, dst=dst)
    return None, None
