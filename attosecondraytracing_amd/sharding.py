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


XHDR = 26   # doubles a rank contributes to the per-step header exchange: count, flags (int64 bit patterns), 24 statistics


class SurvivorGather:
    """The gather of SURVEY.md 8(e) as written: `(number:int32, X, Y, path)` = 28 B per SURVIVING ray from every shard to
    rank `dst` (the consumer, ART/ModuleDetector.py:254-279, sees survivors only; ReadoutGather above ships all slots, dead
    or alive).  Per step and rank, on set b of `buffers` independent buffer sets:

      1. the send buffer.  ZERO-COPY (a shard that loses nothing -- `zero_copy=True`, the caller's knowledge of its scene):
         `acquire(b)` hands out X, Y, path views of the buffer's dense sections, the step's read-out writes straight into
         them (Detector read-out targets: art_trace_chain_readout / scene read-outs take any pointers) and
         `art_survivor_finish` adds the 16-byte header -- no pack, no staging copy, 24 B/ray.  Otherwise `art_pack_survivors`
         compacts the read-out of the alive slots into it (28 B per survivor; 24 where every slot is alive and the numbers
         are the implicit first + slot * step);
      2. ONE small all-gather: every rank's header (count, flags) + its 24 read-out statistics (208 B) -- the global
         statistics of the step (`stats(b)`: the delays are relative to the GLOBAL mean path, ART/ModuleDetector.py:277)
         ride along, and a copy of the headers goes to pinned host memory behind an event NOBODY WAITS FOR in this step;
      3. the payload: every peer SENDS its records to `dst`, which posts one receive per peer -- grouped point-to-point
         operations (on the xGMI mesh every peer has its own link into the root; the root's own shard is read where it
         lies, it is not copied).  A transfer's size must be known on the host when it is issued: it is PREDICTED, per
         rank, from the newest headers the host already holds (those of step i - 2 with two sets), with a margin (the
         larger of `margin` = 1/16 of the count and `slack` = 1024 records; a shard that was dense is predicted dense: it
         cannot grow).  The count travels in the payload's own header, so the root decodes exactly what was packed.

    When the headers of a step land and show that some shard packed MORE than was shipped, or that a zero-copy shard did
    lose rays (header flag `unpacked`), that step is SHORT.  Whoever wants its records calls `settle(b)` (every rank: it
    re-issues that one step's transfers with the exact sizes, packing an unpacked shard first) before set b is acquired
    again; a short step that nobody settled is dropped when its set is reused (`dropped`), its headers still teach the
    next prediction.  Only the first step (nothing to predict from) reads its own headers synchronously.  So a
    steady-state step costs the 208-byte all-gather and the payload transfers, and no host synchronisation;
    `host_syncs` counts the exceptions (first step, settled short steps).

    `start(b, ...)` enqueues a step, `settle(b)` / `drain()` complete it (them), `result(b)` -> per-rank list of (number
    int64, X, Y, path) views on dst, `assemble(b)` -> the four arrays of the whole job in global ray order.

    TILES (`tiles=T` > 1, zero-copy shards): the caller traces its step as T launches over consecutive slot ranges
    (`tile_range(t)`; bundle.RayBundle.slots, graph.SceneProgram(outputs=...)) and calls `start_tile(b, t)` behind each: the
    records of tile t leave while tile t + 1 is traced, so the transfer of a step overlaps THAT step's trace, not only the
    next one's (what matters for a single step: trace / T + transfer instead of trace + transfer).  Which ranks tile is
    decided from the same headers as the sizes -- a rank that was dense two steps ago (`tiling(b)`), on every rank alike;
    `start` then ships only what did not travel in tiles.  A tiled shard that did lose rays is short like any other."""

    def __init__(self, backend, n, world, rank, dst=0, buffers=2, specs=None, margin=1.0 / 16, slack=1024, predict=True,
                 zero_copy=False, tiles=1):
        self.be, self.n, self.world, self.rank, self.dst = backend, int(n), int(world), int(rank), int(dst)
        # (first, step, n) of every rank's shard: for the implicit numbers of dense shards on the root
        self.specs = specs if specs is not None else [(0, 1, self.n)] * self.world
        top = max(first + (cnt - 1) * step for first, step, cnt in self.specs if cnt > 0) if any(s[2] > 0 for s in self.specs) else 0
        if top > 2 ** 31 - 1:
            raise ValueError("ray numbers up to %d do not fit the int32 of a survivor record" % top)
        dev = backend.device
        self.margin, self.slack, self.predict, self.zero_copy = float(margin), int(slack), bool(predict), bool(zero_copy)
        self.n_tiles = max(1, int(tiles))
        self._tiled = [frozenset()] * buffers       # ranks whose records of the step on set b travel tile by tile
        self._tiles_started = [0] * buffers
        # capacity of every buffer: the longest shard with every slot alive and explicit numbers (28 B per ray)
        self.cap = backend.survivor_bytes(max([self.n] + [s[2] for s in self.specs]), False)
        # (a send buffer starts 496 bytes into its allocation: the zero-copy sections begin behind the 16-byte header, and the
        # read-out's stores should meet whole cache lines -- a row that starts 16 bytes off a line costs bandwidth)
        self._send_raw = [torch.empty(self.cap + 512, dtype=torch.uint8, device=dev) for _ in range(buffers)]
        self.send = [t[496:496 + self.cap] for t in self._send_raw]
        self.alt = [None] * buffers           # where an unpacked zero-copy shard is packed when its step is settled
        self._src = list(self.send)           # the buffer this rank's records of set b lie in
        self.recv = [[None if r == self.rank else torch.empty(self.cap, dtype=torch.uint8, device=dev) for r in range(self.world)]
                     if self.rank == self.dst else None for _ in range(buffers)]
        self.xh_mine = [torch.zeros(XHDR, dtype=torch.float64, device=dev) for _ in range(buffers)]
        self.xh = [torch.zeros((self.world, XHDR), dtype=torch.float64, device=dev) for _ in range(buffers)]
        pin = torch.device(dev).type == "cuda"
        self._xh_host = [torch.zeros((self.world, XHDR), dtype=torch.float64, pin_memory=pin) for _ in range(buffers)]
        self._side = torch.cuda.Stream(device=dev) if pin else None
        self._hdr_event = [None] * buffers    # the headers of set b are on the host once this event has completed
        self._state = ["idle"] * buffers      # idle | inflight (started, headers unread) | ok | short
        self._acquired = [False] * buffers
        self._args = [None] * buffers         # (X, Y, opl, alive, number) of the step on set b: what a repair packs from
        self._stats_out = [torch.zeros(24, dtype=torch.float64, device=dev) for _ in range(buffers)]
        self.headers = [None] * buffers       # host copies: [[count, flags]] per rank (valid once set b has been checked)
        self.sizes = [[0] * self.world for _ in range(buffers)]     # bytes shipped per rank on set b
        self.nbytes = [0] * buffers           # ... by this rank
        self.work = [[] for _ in range(buffers)]
        self._known = None                    # newest checked headers: what the next sizes are predicted from
        self.host_syncs = 0                   # steps that read their own headers synchronously + settled short steps
        self.overflows = 0                    # steps whose shipped sizes did not cover what was packed
        self.dropped = 0                      # ... of which nobody asked for the records before the set was reused

    @staticmethod
    def _dist():
        return dist.is_available() and dist.is_initialized()

    # ---- the zero-copy sections ----------------------------------------------------------------------------------------
    def targets(self, b):
        """X, Y, path views (n doubles each) of send buffer b's dense sections."""
        sec = self.send[b][16:16 + 24 * self.n].view(torch.float64).view(3, self.n)
        return sec[0], sec[1], sec[2]

    # ---- tiles ---------------------------------------------------------------------------------------------------------
    @staticmethod
    def _tile_range(n, t, T):
        per = -(-n // T)
        per = -(-per // 64) * 64                  # every tile starts on a 512-byte boundary of every row
        return min(n, t * per), min(n, (t + 1) * per)

    def tile_range(self, t):
        """Slots [lo, hi) of this rank's tile t."""
        return self._tile_range(self.n, t, self.n_tiles)

    def _tiling_ranks(self):
        if self.n_tiles <= 1 or not self.predict or self._known is None:
            return frozenset()
        return frozenset(r for r, (c, f) in enumerate(self._known) if (f & 1) and c == self.specs[r][2])

    def tiling(self, b):
        """After acquire(b): does THIS rank send the step on set b tile by tile (it was dense two steps ago)?"""
        return self.zero_copy and self.rank in self._tiled[b]

    def start_tile(self, b, t):
        """Tile t of the step on set b has been traced (its read-out wrote slots tile_range(t) of the zero-copy sections on
        the caller's stream): its records leave now -- three ranges (X, Y, path) per tiling peer into the root."""
        assert t == self._tiles_started[b], "tiles are started in order, each once"
        self._tiles_started[b] = t + 1
        if not self._dist() or self.world == 1:
            return
        ops = []
        for r in (sorted(self._tiled[b]) if self.rank == self.dst else ([self.rank] if self.rank in self._tiled[b] else [])):
            if r == self.dst:
                continue
            n_r = self.specs[r][2]
            lo, hi = self._tile_range(n_r, t, self.n_tiles)
            if hi <= lo:
                continue
            buf = self.recv[b][r] if self.rank == self.dst else self.send[b]
            for k in range(3):
                piece = buf[16 + 8 * (k * n_r + lo):16 + 8 * (k * n_r + hi)]
                ops.append(dist.P2POp(dist.irecv, piece, r) if self.rank == self.dst else dist.P2POp(dist.isend, piece, self.dst))
        if ops:
            self.work[b] = list(self.work[b]) + list(dist.batch_isend_irecv(ops))

    # ---- sizes ---------------------------------------------------------------------------------------------------------
    def _exact(self, headers):
        return [self.be.survivor_bytes(c, bool(f & 1)) for c, f in headers]

    def _predicted(self, headers):
        out = []
        for (c, f), (_, _, slots) in zip(headers, self.specs):
            if f & 1:                                   # dense last time: predicted dense (a shard cannot grow)
                out.append(self.be.survivor_bytes(c, True))
            else:
                out.append(min(self.cap, self.be.survivor_bytes(min(int(slots), int(c + max(self.slack, self.margin * c))), False)))
        return out

    # ---- transfers -----------------------------------------------------------------------------------------------------
    def _issue(self, b, sizes, skip=frozenset()):
        """The payload of set b: every peer sends sizes[rank] bytes to dst, dst posts one receive per peer (one group);
        `skip`: ranks whose records have travelled already (tile by tile).
        A transfer of 16 bytes would carry the header alone, which the header exchange has delivered already: skipped,
        on both sides alike."""
        self.sizes[b], self.nbytes[b] = list(sizes), int(sizes[self.rank])
        if not skip:
            self.work[b] = []
        if not self._dist() or self.world == 1:
            return                                       # the root's own shard is read where it lies
        ops = []
        if self.rank == self.dst:
            ops = [dist.P2POp(dist.irecv, self.recv[b][r][:sizes[r]], r) for r in range(self.world)
                   if r != self.dst and sizes[r] > 16 and r not in skip]
        elif sizes[self.rank] > 16 and self.rank not in skip:
            ops = [dist.P2POp(dist.isend, self._src[b][:sizes[self.rank]], self.dst)]
        if ops:
            self.work[b] = list(self.work[b]) + list(dist.batch_isend_irecv(ops))

    def _wait(self, b):
        for w in self.work[b]:
            w.wait()          # NCCL / RCCL: the caller's STREAM waits; gloo: the host does
        self.work[b] = []

    def _read_headers(self, b):
        ev = self._hdr_event[b]
        if ev is not None:
            ev.synchronize()
        raw = self._xh_host[b][:, :2].contiguous().view(torch.int64)
        self.headers[b] = raw.tolist()

    def _check(self, b):
        """Headers of set b on the host, the shipped sizes compared with what was packed (no communication)."""
        if self._state[b] != "inflight":
            return self._state[b] in ("ok", "idle")
        self._read_headers(b)
        need = self._exact([(c, f) for c, f in self.headers[b]])
        short = any((f & 2) or nd > sz for (c, f), nd, sz in zip(self.headers[b], need, self.sizes[b]))
        # what the next prediction starts from: an unpacked shard is a sparse shard of that count
        self._known = [[c, f & 1] for c, f in self.headers[b]]
        if short:
            self.overflows += 1
        self._state[b] = "short" if short else "ok"
        return not short

    # ---- per step ------------------------------------------------------------------------------------------------------
    def acquire(self, b):
        """Before the step's trace: the caller's stream waits for the previous transfers of set b (its send buffer is about
        to be rewritten), the step before that is checked (a short one nobody settled is dropped).  -> the zero-copy
        targets (X, Y, path views) if this gather was built with zero_copy, else None."""
        self._wait(b)
        if not self._check(b):
            self.dropped += 1
        self._state[b] = "idle"
        self._src[b] = self.send[b]
        self._acquired[b] = True
        self._tiled[b] = self._tiling_ranks()
        self._tiles_started[b] = 0
        return self.targets(b) if self.zero_copy else None

    def start(self, b, X, Y, opl, alive, stats_dev=None, number=None):
        """Enqueue the step on set b: header (zero-copy: X is the view acquire(b) handed out, stats_dev the read-out's 24
        statistics) or pack, header exchange, payload transfers.  -> the bytes this rank ships."""
        if not self._acquired[b]:
            self.acquire(b)
        self._acquired[b] = False
        if self._tiled[b]:        # (a rank that does not tile itself still posts its side of the peers' tile transfers)
            for t in range(self._tiles_started[b], self.n_tiles):
                self.start_tile(b, t)
        first, step, _ = self.specs[self.rank]
        zero = self.n > 0 and X.data_ptr() == self.targets(b)[0].data_ptr()
        if zero:
            if stats_dev is None or number is not None:
                raise ValueError("zero-copy survivor records need the read-out's statistics and implicit ray numbers")
            self.be.survivor_finish(stats_dev, self.n, self.send[b], self.xh_mine[b])
        else:
            self.be.pack_survivors(alive, X, Y, opl, number, first, step, self.send[b])
            self.be.survivor_xheader(self.send[b], stats_dev, self.xh_mine[b])
        self._args[b] = (X, Y, opl, alive, number)
        if self._side is not None:
            # the headers travel on the communicator's stream, their copy to the host on a side stream behind that: the
            # caller's stream (the next trace) never waits for either
            if self._dist():
                w = dist.all_gather_into_tensor(self.xh[b].view(-1), self.xh_mine[b], async_op=True)
                with torch.cuda.stream(self._side):
                    w.wait()                  # the SIDE stream waits for the collective
            else:
                self.xh[b][0].copy_(self.xh_mine[b])
                self._side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._side):
                self._xh_host[b].copy_(self.xh[b], non_blocking=True)
                self._hdr_event[b] = torch.cuda.Event()
                self._hdr_event[b].record()
        else:                                 # CPU tensors (gloo tests): everything is synchronous
            if self._dist():
                dist.all_gather_into_tensor(self.xh[b].view(-1), self.xh_mine[b])
            else:
                self.xh[b][0].copy_(self.xh_mine[b])
            self._xh_host[b].copy_(self.xh[b])
        self._state[b] = "inflight"
        if self._known is None or not self.predict:
            self.host_syncs += 1              # nothing to predict from: this step's own headers, synchronously
            self._read_headers(b)
            self._issue(b, self._exact([(c, f) for c, f in self.headers[b]]))
            self._check(b)                    # exact by construction -- unless a zero-copy shard came out unpacked (settle repairs)
            return self.nbytes[b]
        sizes = self._predicted(self._known)
        self._issue(b, sizes, skip=self._tiled[b])
        return self.nbytes[b]

    def settle(self, b):
        """Complete the step on set b -- EVERY rank calls it (a short step's transfers are issued again with the exact sizes,
        which is communication); afterwards result(b) holds on dst.  Waits for set b only: the step on the other set may
        still be tracing."""
        if self._state[b] == "idle":
            return
        self._wait(b)
        if self._check(b):
            return
        self.host_syncs += 1
        mine = self.headers[b][self.rank]
        if mine[1] & 2:                       # a zero-copy shard that lost rays: pack its records (the sections are intact)
            X, Y, opl, alive, number = self._args[b]
            if self.alt[b] is None:
                self.alt[b] = torch.empty(self.cap, dtype=torch.uint8, device=self.be.device)
            first, step, _ = self.specs[self.rank]
            self.be.pack_survivors(alive, X, Y, opl, number, first, step, self.alt[b])
            self._src[b] = self.alt[b]
        self.headers[b] = [[c, (f & 1) if not (f & 2) else 0] for c, f in self.headers[b]]     # packed now: sparse records
        self._issue(b, self._exact([(c, f) for c, f in self.headers[b]]))
        self._wait(b)
        self._state[b] = "ok"

    def drain(self):
        for b in range(len(self.send)):
            self.settle(b)

    def stats(self, b):
        """Global read-out statistics of the step on set b (DEVICE tensor [24]): the shards' statistics rode on the header
        exchange; folded on the device (sums added, minima / maxima folded)."""
        if self._hdr_event[b] is not None:
            torch.cuda.current_stream().wait_event(self._hdr_event[b])      # (recorded behind the all-gather)
        self.be.exchange_fold(self.xh[b].view(-1)[2:], self.world, XHDR, self._stats_out[b])
        return self._stats_out[b]

    def result(self, b):
        """On dst: [(number int64 [c], X [c], Y [c], path [c])] per rank (views of the receive buffers -- the root's own
        shard: of its send buffer; the numbers of a dense shard are generated).  None elsewhere.  With more than one rank
        the step must have been settled (settle(b) / drain() on EVERY rank)."""
        if self._state[b] not in ("ok", "idle") and self.world > 1 and self._dist():
            raise RuntimeError("SurvivorGather.result: set %d is not settled yet -- call settle(%d) or drain() on every rank first" % (b, b))
        self.settle(b)
        if self.recv[b] is None:
            return None
        bufs = [self._src[b] if r == self.rank else buf for r, buf in enumerate(self.recv[b])]
        return [decode_survivors(buf, *self.headers[b][r], self.specs[r]) for r, buf in enumerate(bufs)]

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
