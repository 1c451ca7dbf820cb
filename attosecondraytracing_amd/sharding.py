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


def shard_spec(n_total, rank, world, layout="blocks"):
    """(first, step, n) of `rank`'s shard: slot i holds global ray first + i * step.
    "blocks": contiguous index ranges (shard_range; rank order = global ray order).
    "strided": rank r holds rays r, r + world, r + 2 world, ... -- every rank samples the whole aperture of a Vogel
    spiral (which orders rays by radius), so masks and overfilled apertures cost every rank the same share of its rays;
    the global order is restored by interleaving (assemble)."""
    if layout == "blocks":
        lo, hi = shard_range(n_total, rank, world)
        return lo, 1, hi - lo
    if layout == "strided":
        return rank, world, (n_total - rank + world - 1) // world if n_total > rank else 0
    raise ValueError("layout must be 'blocks' or 'strided'")


def assemble(per_rank, layout="blocks"):
    """Global-order array from equal-length per-rank results [world, ..., n] (e.g. ReadoutGather.result):
    concatenation for "blocks", interleaving for "strided" -> [..., world * n]."""
    world, n = per_rank.shape[0], per_rank.shape[-1]
    if layout == "blocks":
        return torch.cat(list(per_rank), dim=-1)
    return torch.stack(list(per_rank), dim=-1).reshape(per_rank.shape[1:-1] + (n * world,))


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


def sample_slots(n, k, device):
    """`k` evenly spaced slot indices of a shard of n slots (all of them if n <= k), as an int64 tensor on `device`:
    slot j = floor(j (n-1) / (k-1)) in integer arithmetic -- strictly increasing for k <= n, first 0, last n-1, and
    never n (a float32 linspace rounds n-1 up to n above 2^24 slots)."""
    n, k = int(n), int(k)
    if n <= k:
        return torch.arange(n, dtype=torch.int64, device=device)
    return (torch.arange(k, dtype=torch.int64, device=device) * (n - 1)) // max(k - 1, 1)


def gather_sample(X, Y, opl, alive, slots, dst=0, pack=None):
    """Gather an evenly spaced SAMPLE of every shard's read-out to rank `dst`: what a spot diagram or delay graph
    consumes (the plots draw at most ~2e4 markers; the statistics they print are reduced over all rays by
    allreduce_stats).  A few hundred kB instead of 25 B x every ray, so it fits inside every step.
    `slots` = sample_slots(...) of this shard (every rank must use the same count); `pack` may hold preallocated
    {'recv': [world x [4, k]]} on dst.  Returns ([4, world*k] float64: X, Y, opl, alive-as-0/1) on dst, None elsewhere."""
    send = torch.stack([X.index_select(0, slots), Y.index_select(0, slots), opl.index_select(0, slots),
                        alive.index_select(0, slots).to(torch.float64)])
    if not (dist.is_available() and dist.is_initialized()):
        return send
    world, rank = dist.get_world_size(), dist.get_rank()
    if rank == dst:
        recv = pack["recv"] if pack else [torch.empty_like(send) for _ in range(world)]
        dist.gather(send, recv, dst=dst)
        return torch.cat(recv, dim=1)
    dist.gather(send, None, dst=dst)
    return None


class Exchange:
    """The per-step exchange of a sharded run in ONE collective: statistics of every shard + an evenly spaced sample
    of every shard's read-out, all-gathered (every rank gets the global statistics; any rank can draw the sample).
    Buffers are allocated once; per step it costs two tiny kernels (art_exchange_pack / art_exchange_fold) and the
    all-gather -- the host work of a step stays far below its GPU time.

    Two ways to use it.  `exchange(stats, X, Y, opl, alive)` does everything in place on the caller's stream.
    `start(b, ...)` / `finish(b)` split it over `buffers` independent buffer sets: `start` packs and enqueues the
    collective asynchronously (it runs on the communicator's stream behind the pack kernel), `finish` makes the caller's
    stream wait for it and folds -- so the all-gather of step i travels while step i+1 is traced, and its ~0.1 ms of
    latency does not add to a 0.7-ms step."""

    def __init__(self, backend, n_slots, sample=20000, buffers=2):
        self.be = backend
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        per_rank = 0 if sample <= 0 else max(1, sample // self.world)      # sample = 0: statistics only
        self.slots = (sample_slots(n_slots, per_rank, backend.device) if per_rank
                      else torch.empty(0, dtype=torch.int64, device=backend.device))
        self.k = int(self.slots.numel())
        # art_exchange_pack reads X/Y/opl/alive[slot] without a bounds check of its own
        assert self.k == 0 or (int(self.slots.min()) >= 0 and int(self.slots.max()) < int(n_slots)), "sample slot out of range"
        self.stride = 24 + 4 * self.k
        self.sends = [torch.empty(self.stride, dtype=torch.float64, device=backend.device) for _ in range(buffers)]
        self.recvs = [torch.empty(self.world * self.stride, dtype=torch.float64, device=backend.device) for _ in range(buffers)]
        self.statss = [torch.empty(24, dtype=torch.float64, device=backend.device) for _ in range(buffers)]
        self.work = [None] * buffers
        self.pending = [False] * buffers
        self.send, self.recv, self.stats = self.sends[0], self.recvs[0], self.statss[0]     # set 0 under the old names

    def start(self, b, stats_dev, X, Y, opl, alive):
        """Pack this rank's contribution into buffer set b and enqueue the all-gather (asynchronous where a process
        group exists).  Set b must have been finished since its last start."""
        assert not self.pending[b], "Exchange.start on a buffer set whose previous exchange was not finished"
        self.be.exchange_pack(stats_dev, X, Y, opl, alive, self.slots, self.sends[b])
        if self.world > 1 or (dist.is_available() and dist.is_initialized()):
            self.work[b] = dist.all_gather_into_tensor(self.recvs[b], self.sends[b], async_op=True)
        else:
            self.recvs[b].copy_(self.sends[b])
        self.pending[b] = True

    def finish(self, b):
        """-> (global statistics [24], sample [world, k, 4] = X, Y, opl, alive) of the exchange started on set b --
        views of reused buffers, valid until set b is started again."""
        assert self.pending[b], "Exchange.finish without a start"
        if self.work[b] is not None:
            self.work[b].wait()          # the caller's stream waits; the host does not (NCCL/RCCL), or blocks (gloo)
            self.work[b] = None
        self.pending[b] = False
        self.be.exchange_fold(self.recvs[b], self.world, self.stride, self.statss[b])
        return self.statss[b], self.recvs[b].view(self.world, self.stride)[:, 24:].reshape(self.world, self.k, 4)

    def __call__(self, stats_dev, X, Y, opl, alive):
        """Returns (global statistics [24], sample [world, k, 4] = X, Y, opl, alive) -- views of reused buffers."""
        self.start(0, stats_dev, X, Y, opl, alive)
        return self.finish(0)


class ReadoutGather:
    """The gather BASELINE.json's north_star names: every ray's read-out (X, Y, optical path: fp64; alive: 1 byte;
    25 B/ray) from every shard to rank `dst` in ONE collective.  The four arrays are packed into one byte buffer per
    rank ([3, n] fp64 followed by [n] uint8, padded to 8 bytes) so that a single `gather` moves them; on the xGMI mesh
    every peer has its own link into the root, so the collective is bound by one link per peer, not by a ring.

    `buffers` independent send/receive sets let the gather of step i (RCCL's stream) overlap the tracing of step i+1:
    `start(b, ...)` waits for the previous use of set b, packs (three row copies + one byte copy on the caller's
    stream) and enqueues the collective; `drain()` waits for everything in flight; `result(b)` returns, on dst, views
    ([world, 3, n] fp64, [world, n] uint8) of receive set b -- rank-major = global ray order for equal shards."""

    def __init__(self, n, world, rank, device, dst=0, buffers=2):
        self.n, self.world, self.rank, self.dst = int(n), int(world), int(rank), int(dst)
        self.nbytes = (25 * self.n + 7) // 8 * 8
        self.send = [torch.empty(self.nbytes, dtype=torch.uint8, device=device) for _ in range(buffers)]
        self.recv = [[torch.empty(self.nbytes, dtype=torch.uint8, device=device) for _ in range(self.world)]
                     if self.rank == self.dst else None for _ in range(buffers)]
        self.work = [None] * buffers

    def _split(self, buf):
        return buf[:24 * self.n].view(torch.float64).view(3, self.n), buf[24 * self.n:25 * self.n]

    def start(self, b, X, Y, opl, alive):
        if self.work[b] is not None:
            self.work[b].wait()
        xyo, al = self._split(self.send[b])
        xyo[0].copy_(X); xyo[1].copy_(Y); xyo[2].copy_(opl)
        al.copy_(alive)
        if dist.is_available() and dist.is_initialized():
            self.work[b] = dist.gather(self.send[b], self.recv[b], dst=self.dst, async_op=True)     # ONE collective
        elif self.recv[b] is not None:
            self.recv[b][0].copy_(self.send[b])

    def drain(self):
        for b, w in enumerate(self.work):
            if w is not None:
                w.wait()
                self.work[b] = None

    def result(self, b):
        if self.recv[b] is None:
            return None, None
        parts = [self._split(t) for t in self.recv[b]]
        return torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts])


def decode_survivors(buf, count, flags, spec):
    """One rank's survivor records (the byte layout of art_pack_survivors, include/art_hip.h) -> (number int64 [c], X, Y,
    path) as views of `buf` (uint8); the numbers of a dense shard (flags bit 0) are generated from its `spec` = (first,
    step, n)."""
    c = int(count)
    f64 = buf[16:16 + 24 * c].view(torch.float64).view(3, c)
    if flags & 1:
        first, step, _ = spec
        num = first + step * torch.arange(c, dtype=torch.int64, device=buf.device)
    else:
        num = buf[16 + 24 * c:16 + 28 * c].view(torch.int32).to(torch.int64)
    return num, f64[0], f64[1], f64[2]


class SurvivorGather:
    """The gather of SURVEY.md 8(e) as written: `(number:int32, X, Y, path)` = 28 B per SURVIVING ray from every shard to
    rank `dst` in ONE payload collective (the consumer, ART/ModuleDetector.py:254-279, sees survivors only; ReadoutGather
    above ships all slots, dead or alive).  Per step and rank:
      1. `art_pack_survivors` compacts the read-out of the alive slots into the rank's send buffer (header: count, flags;
         a shard whose every slot is alive and whose numbers are the implicit first + slot * step drops the number section:
         24 B/ray);
      2. the 16-byte headers are all-gathered on the device and copied to pinned host memory behind an event -- NOBODY
         WAITS for them in this step;
      3. ONE `gather` of `nbytes` bytes per rank, asynchronous on the communicator's stream.  A collective's size must be
         known on the host when it is issued: it is PREDICTED from the newest headers the host already holds (those of
         step i - 2 with two buffer sets: settled when their buffer set is taken again), with a margin (the larger of
         `margin` = 1/16 of the count and `slack` = 1024 records, explicit numbers assumed; a shard that was dense is predicted dense).  The
         count travels inside the payload's own header, so the root decodes exactly what was packed.  When the headers of
         a step land and show that some shard packed MORE than was shipped (overflow), that one step's gather is issued
         again with the exact size -- before its send buffer is packed again, on every rank alike (all ranks read the
         same all-gathered headers at the same point of the program).  Only the first step of a gather (nothing to
         predict from) reads its own headers synchronously.
    So a steady-state step costs two collectives -- a 16-byte all-gather nobody waits for and the payload gather -- and no
    host synchronisation; `host_syncs` counts the exceptions (first step, overflows).
    `buffers` independent sets let the gather of step i overlap the tracing of step i + 1 (start / drain / result as in
    ReadoutGather).  `result(b)` -> per-rank list of (number int64, X, Y, path) views on dst; `assemble(b)` -> the four
    arrays of the whole job in global ray order."""

    def __init__(self, backend, n, world, rank, dst=0, buffers=2, specs=None, margin=1.0 / 16, slack=1024, predict=True):
        self.be, self.n, self.world, self.rank, self.dst = backend, int(n), int(world), int(rank), int(dst)
        # (first, step, n) of every rank's shard: for the implicit numbers of dense shards on the root
        self.specs = specs if specs is not None else [(0, 1, self.n)] * self.world
        top = max(first + (cnt - 1) * step for first, step, cnt in self.specs if cnt > 0) if any(s[2] > 0 for s in self.specs) else 0
        if top > 2 ** 31 - 1:
            raise ValueError("ray numbers up to %d do not fit the int32 of a survivor record" % top)
        dev = backend.device
        self.margin, self.slack, self.predict = float(margin), int(slack), bool(predict)
        # capacity of every buffer: the longest shard with every slot alive and explicit numbers (28 B per ray)
        self.cap = backend.survivor_bytes(max([self.n] + [s[2] for s in self.specs]), False)
        self.send = [torch.empty(self.cap, dtype=torch.uint8, device=dev) for _ in range(buffers)]
        self.recv = [[torch.empty(self.cap, dtype=torch.uint8, device=dev) for _ in range(self.world)]
                     if self.rank == self.dst else None for _ in range(buffers)]
        self.hdr = [torch.zeros((self.world, 2), dtype=torch.int64, device=dev) for _ in range(buffers)]
        pin = torch.device(dev).type == "cuda"
        self._hdr_host = [torch.zeros((self.world, 2), dtype=torch.int64, pin_memory=pin) for _ in range(buffers)]
        self._side = torch.cuda.Stream(device=dev) if pin else None
        self._hdr_event = [None] * buffers    # the headers of set b are on the host once this event has completed
        self._settled = [True] * buffers      # headers read and the shipped size checked against them
        self.headers = [None] * buffers       # host copies: [[count, flags]] per rank (valid once set b is settled)
        self.nbytes = [0] * buffers           # size of the collective issued on set b
        self.work = [None] * buffers
        self._known = None                    # newest settled headers: what the next size is predicted from
        self.host_syncs = 0                   # steps that read their own headers synchronously + overflow repairs
        self.overflows = 0

    # ---- sizes ---------------------------------------------------------------------------------------------------------
    def _exact_bytes(self, headers):
        return max(self.be.survivor_bytes(c, bool(f & 1)) for c, f in headers)

    def _predicted_bytes(self, headers):
        need = 0
        for (c, f), (_, _, slots) in zip(headers, self.specs):
            if f & 1:                                   # dense last time: predicted dense (a shard cannot grow)
                b = self.be.survivor_bytes(c, True)
            else:
                b = self.be.survivor_bytes(min(int(slots), int(c + max(self.slack, self.margin * c))), False)
            need = max(need, b)
        return min(need, self.cap)

    # ---- collectives ---------------------------------------------------------------------------------------------------
    def _issue(self, b, nb):
        self.nbytes[b] = nb
        if dist.is_available() and dist.is_initialized():
            recv = None if self.recv[b] is None else [t[:nb] for t in self.recv[b]]
            self.work[b] = dist.gather(self.send[b][:nb], recv, dst=self.dst, async_op=True)     # ONE payload collective
        elif self.recv[b] is not None:
            self.recv[b][0][:nb].copy_(self.send[b][:nb])

    def _read_headers(self, b):
        ev = self._hdr_event[b]
        if ev is not None:
            ev.synchronize()
        self.headers[b] = self._hdr_host[b].tolist()

    def _settle(self, b):
        """Headers of set b on the host, the shipped size checked: a step that packed more than was shipped is gathered
        again with the exact size (its send buffer is still intact)."""
        if self._settled[b]:
            return
        self._read_headers(b)
        need = self._exact_bytes(self.headers[b])
        if need > self.nbytes[b]:
            self.overflows += 1
            self.host_syncs += 1
            if self.work[b] is not None:
                self.work[b].wait()
                self.work[b] = None
            self._issue(b, need)
            if self.work[b] is not None:
                self.work[b].wait()
                self.work[b] = None
        self._settled[b] = True
        self._known = self.headers[b]

    def start(self, b, X, Y, opl, alive, number=None):
        """Pack this rank's survivors into buffer set b and enqueue the gather; returns the bytes shipped per rank."""
        if self.work[b] is not None:
            self.work[b].wait()
            self.work[b] = None
        self._settle(b)                       # the previous use of this set (two steps ago with two sets)
        first, step, _ = self.specs[self.rank]
        self.be.pack_survivors(alive, X, Y, opl, number, first, step, self.send[b])
        mine = self.send[b][:16].view(torch.int64)
        if self._side is not None:
            # the headers travel on the communicator's stream BEHIND the previous step's payload gather, and their copy to
            # the host on a side stream behind that: the caller's stream (the next trace) never waits for either
            if dist.is_available() and dist.is_initialized():
                w = dist.all_gather_into_tensor(self.hdr[b].view(-1), mine, async_op=True)
                with torch.cuda.stream(self._side):
                    w.wait()                  # the SIDE stream waits for the collective
            else:
                self.hdr[b][0].copy_(mine)
                self._side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._side):
                self._hdr_host[b].copy_(self.hdr[b], non_blocking=True)
                self._hdr_event[b] = torch.cuda.Event()
                self._hdr_event[b].record()
        else:                                 # CPU tensors (gloo tests): everything is synchronous
            if dist.is_available() and dist.is_initialized():
                dist.all_gather_into_tensor(self.hdr[b].view(-1), mine)
            else:
                self.hdr[b][0].copy_(mine)
            self._hdr_host[b].copy_(self.hdr[b])
        self._settled[b] = False
        if self._known is None or not self.predict:
            self.host_syncs += 1              # nothing to predict from: this step's own headers, synchronously
            self._read_headers(b)
            nb = self._exact_bytes(self.headers[b])
            self._settled[b], self._known = True, self.headers[b]       # exact by construction
        else:
            nb = self._predicted_bytes(self._known)
        self._issue(b, nb)
        return nb

    def drain(self):
        for b, w in enumerate(self.work):
            if w is not None:
                w.wait()
                self.work[b] = None
            self._settle(b)

    def result(self, b):
        """On dst: [(number int64 [c], X [c], Y [c], path [c])] per rank (views of receive set b; the numbers of a dense
        shard are generated).  None elsewhere.  With more than one rank, call drain() on EVERY rank first: settling a set
        may re-issue its gather (overflow), which is a collective."""
        if not self._settled[b] and self.world > 1 and dist.is_available() and dist.is_initialized():
            raise RuntimeError("SurvivorGather.result: set %d is not settled yet -- call drain() on every rank first" % b)
        if self.work[b] is not None:
            self.work[b].wait()
            self.work[b] = None
        self._settle(b)
        if self.recv[b] is None:
            return None
        return [decode_survivors(buf, *self.headers[b][r], self.specs[r]) for r, buf in enumerate(self.recv[b])]

    def assemble(self, b):
        """On dst: (number, X, Y, path) of all survivors of the job in global ray order (rank order for contiguous
        shards; merged by ray number for strided ones)."""
        parts = self.result(b)
        if parts is None:
            return None
        num, X, Y, P = (torch.cat([p[k] for p in parts]) for k in range(4))
        if any(s[1] != 1 for s in self.specs):
            order = torch.argsort(num, stable=True)
            num, X, Y, P = num[order], X[order], Y[order], P[order]
        return num, X, Y, P
