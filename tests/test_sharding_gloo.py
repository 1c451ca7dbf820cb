"""CPU, world_size 2 over gloo: the N > 1 path of bench.py / sharding.py -- index-range shards generated per rank,
traced independently (CPU twin of the kernels here), statistics all-reduced, read-out gathered to rank 0 --
reproduces the single-process result exactly (rays are independent; rank order = global ray order)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as tmp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    """A TCP port nobody listens on right now (the rendezvous of the worker processes)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _scene(n_total):
    import ART.ModuleMirror as mmirror
    import ART.ModuleMask as mmask
    import ART.ModuleSupport as msupp
    import ART.ModuleProcessing as mp
    SP = {"Divergence": 0.025, "SourceSize": 0, "Wavelength": 50e-6, "DeltaFT": 0.5, "NumberRays": 64}
    Mask = mmask.Mask(msupp.SupportRoundHole(30, 10.25, 0, 0))
    R, r = mmirror.ReturnOptimalToroidalRadii(600, 80)
    Tor = mmirror.MirrorToroidal(R, r, msupp.SupportRectangle(200, 30))
    return mp.OEPlacement(SP, [Mask, Tor, Tor], [500, 100, 600], [0, 80, -80], [0, 0, 40.0], "gloo")


def _run_shard(rank, world, n_total):
    """Trace shard `rank` of a n_total-ray point source; returns (readout dict, last bundle, detector)."""
    from attosecondraytracing_amd import sharding, ModuleGeometry as mgeo
    from attosecondraytracing_amd.bundle import RayBundle
    from attosecondraytracing_amd import _lib
    import ART.ModuleProcessing as mp
    import ART.ModuleDetector as mdet
    be = _lib.get_backend()
    chain = _scene(n_total)
    lo, hi = sharding.shard_range(n_total, rank, world)
    src = RayBundle.allocate(hi - lo, backend=be)
    rot = mgeo.rotation_matrix(np.array([0.0, 0.0, 1.0]), np.array([1.0, 0.0, 0.0]))
    be.make_source(0, 0.025, rot, np.zeros(3), lo, hi - lo, n_total, src.view())
    src.intensity = torch.ones(hi - lo, dtype=torch.float64)
    out = mp.RayTracingCalculation(src, chain.optical_elements)
    det = mdet.Detector(np.zeros(3), np.array([1900.0, 30.0, 0.0]), np.array([-0.9, -0.1, 0.2]))
    return det.readout(out[-1], sync=False), out[-1], det


def _worker(rank, world, port, n_total, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from twin_backend import TwinBackend
    from attosecondraytracing_amd import _lib, sharding
    _lib._BACKEND = TwinBackend()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        r, last, det = _run_shard(rank, world, n_total)
        stats = sharding.allreduce_stats(r["stats_dev"], torch.device("cpu"))
        again = sharding.allreduce_stats(r["stats_dev"], torch.device("cpu"), async_op=True).result()
        assert torch.equal(stats, again)
        XYO, alive = sharding.gather_readout(r["X"], r["Y"], r["opl"], last.alive, 0)
        # the overlapped form bench.py uses: preallocated pack, known sizes, handles waited for later
        sizes = [b - a for a, b in (sharding.shard_range(n_total, k, world) for k in range(world))]
        nmax = max(sizes)
        pack = {"send": torch.zeros((3, nmax), dtype=torch.float64), "asend": torch.zeros(nmax, dtype=torch.uint8)}
        if rank == 0:
            pack["recv"] = [torch.empty((3, nmax), dtype=torch.float64) for _ in range(world)]
            pack["arecv"] = [torch.empty(nmax, dtype=torch.uint8) for _ in range(world)]
        for w in sharding.gather_readout(r["X"], r["Y"], r["opl"], last.alive, 0, pack, sizes, async_op=True):
            w.wait()
        if rank == 0:
            again = torch.cat([t[:, :sz] for t, sz in zip(pack["recv"], sizes)], dim=1)
            assert torch.equal(again, XYO)
        # sampled gather (what bench.py does inside every step at N > 1): equal counts on every rank
        k = 100
        slots = sharding.sample_slots(min(sizes), k, torch.device("cpu"))
        S = sharding.gather_sample(r["X"], r["Y"], r["opl"], last.alive, slots, 0)
        if rank == 0:
            assert S.shape == (4, world * len(slots))
            off = 0
            for kk in range(world):
                part = S[:, kk * len(slots):(kk + 1) * len(slots)]
                assert torch.equal(part[0:3], XYO[:, off + slots]) and torch.equal(part[3], alive[off + slots].to(torch.float64))
                off += sizes[kk]
        else:
            assert S is None
        # the one-collective form bench.py uses in every step at N > 1: statistics + sample in ONE all-gather
        ex = sharding.Exchange(_lib.get_backend(), min(sizes), sample=200)
        st2, smp = ex(r["stats_dev"], r["X"], r["Y"], r["opl"], last.alive)
        assert torch.equal(st2, stats)                       # same fold as allreduce_stats, bit for bit
        assert smp.shape == (world, ex.k, 4)
        mine = torch.stack([r["X"][ex.slots], r["Y"][ex.slots], r["opl"][ex.slots], last.alive[ex.slots].to(torch.float64)], dim=1)
        assert torch.equal(smp[rank], mine)
        if rank == 0:
            off = sizes[0]
            assert torch.equal(smp[1][:, 0:3], XYO[:, off + ex.slots].T)    # the other rank's sample, global order
        # the pipelined form (bench.py: the all-gather of step i travels while step i+1 is traced): both buffer sets in
        # flight, finished in order, same results; a set cannot be started twice without a finish
        ex.start(0, r["stats_dev"], r["X"], r["Y"], r["opl"], last.alive)
        ex.start(1, r["stats_dev"], r["X"], r["Y"], r["opl"], last.alive)
        try:
            ex.start(0, r["stats_dev"], r["X"], r["Y"], r["opl"], last.alive)
            raise RuntimeError("second start on a pending buffer set was accepted")
        except AssertionError:
            pass
        for b in (0, 1):
            stb, smpb = ex.finish(b)
            assert torch.equal(stb, stats) and torch.equal(smpb[rank], mine)
        # the north_star's gather as ONE collective (what bench.py times as value_full_gather): equal shards, two
        # buffer sets used alternately
        nmin = min(sizes)
        rg = sharding.ReadoutGather(nmin, world, rank, torch.device("cpu"), dst=0, buffers=2)
        for b in (0, 1, 0):
            rg.start(b, r["X"][:nmin], r["Y"][:nmin], r["opl"][:nmin], last.alive[:nmin])
        rg.drain()
        got, galive = rg.result(0)
        if rank == 0:
            assert got.shape == (world, 3, nmin) and galive.shape == (world, nmin)
            off = 0
            for kk in range(world):
                assert torch.equal(got[kk], XYO[:, off:off + nmin]) and torch.equal(galive[kk], alive[off:off + nmin])
                off += sizes[kk]
        else:
            assert got is None and galive is None
        # the gather of SURVEY 8(e) as written: (number:int32, X, Y, path) of the SURVIVORS only -- every peer sends its
        # records to the root (sizes per rank), unequal shards, two buffer sets used alternately; then a shard with every
        # slot alive (dense: no number section)
        be = _lib.get_backend()
        specs = [sharding.shard_spec(n_total, k, world) for k in range(world)]
        sg = sharding.SurvivorGather(be, sizes[rank], world, rank, dst=0, buffers=2, specs=specs)
        nb = [sg.start(b, r["X"], r["Y"], r["opl"], last.alive, r["stats_dev"]) for b in (0, 1, 0)]
        sg.drain()
        counts = [c for c, _ in sg.headers[0]]
        exact = [be.survivor_bytes(c, bool(f)) for c, f in sg.headers[0]]
        # the first step reads its own headers (exact sizes, the one host synchronisation); the following ones are sized
        # from headers the host already holds, with a margin -- never below what was packed, never above the capacity
        assert nb[0] == exact[rank] and sg.host_syncs == 1 and sg.overflows == 0 and sg.dropped == 0
        assert exact[rank] <= nb[1] <= sg.cap and exact[rank] <= nb[2] <= sg.cap
        assert all(e <= z <= sg.cap for e, z in zip(exact, sg.sizes[0]))
        assert [f for _, f in sg.headers[0]] == [int(c == sz) for c, sz in zip(counts, sizes)]    # dense iff nothing was lost
        assert sum(counts) < n_total and counts[rank] == int(last.alive.sum())
        # the shards' statistics rode on the header exchange: the global ones, same fold as allreduce_stats
        assert torch.equal(sg.stats(0), stats)
        surv = sg.assemble(0)
        everyone = torch.ones_like(last.alive)
        sg.start(1, r["X"], r["Y"], r["opl"], everyone)
        sg.drain()
        # (whatever was predicted at start, the settled size covers the dense records)
        assert [f for _, f in sg.headers[1]] == [1] * world and sg.nbytes[1] >= be.survivor_bytes(sizes[rank], True)
        dense = sg.assemble(1)
        # ZERO-COPY: the "read-out" writes X, Y, path straight into the send buffer's dense sections; with every slot alive
        # art_survivor_finish's header completes the buffer -- same records as the packed dense step, no pack
        zc = sharding.SurvivorGather(be, sizes[rank], world, rank, dst=0, buffers=2, specs=specs, zero_copy=True)
        all_stats = r["stats_dev"].clone()
        all_stats[0] = sizes[rank]
        packs = [0]
        real_pack = be.pack_survivors
        be.pack_survivors = lambda *a, **k: (packs.__setitem__(0, packs[0] + 1), real_pack(*a, **k))[1]
        try:
            for b in (0, 1, 0):
                tx, ty, tp = zc.acquire(b)
                tx.copy_(r["X"]); ty.copy_(r["Y"]); tp.copy_(r["opl"])
                zc.start(b, tx, ty, tp, everyone, all_stats)
            zc.drain()
            assert packs[0] == 0 and zc.overflows == 0 and zc.host_syncs == 1
            assert zc.nbytes[0] == be.survivor_bytes(sizes[rank], True)
            zdense = zc.assemble(0)
            # ... and a zero-copy shard that DID lose rays: its header says `unpacked`; settle() packs it from the sections and
            # ships the exact records -- on every rank alike
            lost_stats = r["stats_dev"].clone()          # (its slot 0 is the true count of survivors of last.alive)
            tx, ty, tp = zc.acquire(1)
            tx.copy_(r["X"]); ty.copy_(r["Y"]); tp.copy_(r["opl"])
            zc.start(1, tx, ty, tp, last.alive, lost_stats)
            lost_some = counts[rank] < sizes[rank]
            zc.settle(1)
            assert packs[0] == (1 if lost_some else 0) and zc.overflows == 1       # (some rank lost rays: the step was short)
            zlost = zc.assemble(1)
            # a short step nobody settles is dropped when its set is reused; its headers still teach the next prediction
            tx, ty, tp = zc.acquire(0)
            tx.copy_(r["X"]); ty.copy_(r["Y"]); tp.copy_(r["opl"])
            zc.start(0, tx, ty, tp, last.alive, lost_stats)
            zc.acquire(0)
            assert zc.dropped == 1 and zc.overflows == 2
            zc.start(0, r["X"], r["Y"], r["opl"], last.alive, r["stats_dev"])      # (not the targets: the packing path)
            zc.drain()
            assert zc.overflows == 2
            zpacked = zc.assemble(0)
            # TILES: the step's records leave in T ranges of the zero-copy sections, each behind "its" part of the trace;
            # which ranks tile follows from the headers of two steps earlier (here: every rank was dense) -- same records
            tz = sharding.SurvivorGather(be, sizes[rank], world, rank, dst=0, buffers=2, specs=specs, zero_copy=True, tiles=3)
            used_tiles = []
            for step_, b in enumerate((0, 1, 0, 1)):
                tx, ty, tp = tz.acquire(b)
                if tz.tiling(b):
                    for t in range(3):
                        lo, hi = tz.tile_range(t)
                        tx[lo:hi].copy_(r["X"][lo:hi]); ty[lo:hi].copy_(r["Y"][lo:hi]); tp[lo:hi].copy_(r["opl"][lo:hi])
                        tz.start_tile(b, t)
                    used_tiles.append(step_)
                else:
                    tx.copy_(r["X"]); ty.copy_(r["Y"]); tp.copy_(r["opl"])
                tz.start(b, tx, ty, tp, everyone, all_stats)
            tz.drain()
            assert used_tiles == [1, 2, 3] and tz.overflows == 0 and tz.host_syncs == 1       # (the first step has nothing to go by)
            assert tz.tile_range(0)[0] == 0 and tz.tile_range(2)[1] == sizes[rank] and tz.tile_range(1)[0] % 64 == 0
            ztiled = [tz.assemble(0), tz.assemble(1)]
        finally:
            be.pack_survivors = real_pack
        # OVERFLOW: a gather sized from a step that lost most of its rays, followed by a step in which every ray survives
        # (no margin, no slack: the prediction is the previous count).  The shipped size is too small; when that step is
        # settled its transfers are issued again with the exact sizes -- on both ranks alike -- and the assembled result is
        # the exact one.  A step that packs LESS than predicted is simply decoded by its own count.
        few = torch.zeros_like(last.alive)
        few[::10] = last.alive[::10]
        tight = sharding.SurvivorGather(be, sizes[rank], world, rank, dst=0, buffers=2, specs=specs, margin=0.0, slack=0)
        n0 = tight.start(0, r["X"], r["Y"], r["opl"], few)
        n1 = tight.start(1, r["X"], r["Y"], r["opl"], few)
        assert n0 == n1 and tight.host_syncs == 1
        n2 = tight.start(0, r["X"], r["Y"], r["opl"], everyone)          # predicted from `few`: too small
        assert n2 == n0 and tight.overflows == 0
        n3 = tight.start(1, r["X"], r["Y"], r["opl"], last.alive)        # (packs more than `few` as well)
        tight.drain()
        assert tight.overflows == 2 and tight.host_syncs == 3 and tight.nbytes[0] == be.survivor_bytes(sizes[rank], True)
        over, after = tight.assemble(0), tight.assemble(1)
        n4 = tight.start(0, r["X"], r["Y"], r["opl"], few)               # predicted from the big step: shrinks again
        tight.drain()
        small = tight.assemble(0)
        assert n4 >= n0 and tight.overflows == 2
        try:            # the records of an unsettled step are refused, not guessed
            tight.start(1, r["X"], r["Y"], r["opl"], everyone)
            tight.result(1)
            raise AssertionError("result() of an unsettled short step was handed out")
        except RuntimeError as e:
            assert "not settled" in str(e)
        tight.drain()
        if rank == 0:
            assert all(torch.equal(a_, b_) for a_, b_ in zip(over, dense))
            assert all(torch.equal(a_, b_) for a_, b_ in zip(after, surv))
            assert all(torch.equal(a_, b_) for a_, b_ in zip(zdense, dense))
            assert all(torch.equal(a_, b_) for zt in ztiled for a_, b_ in zip(zt, dense))
            assert all(torch.equal(a_, b_) for a_, b_ in zip(zlost, surv))
            assert all(torch.equal(a_, b_) for a_, b_ in zip(zpacked, surv))
            # the survivors of `few`: those of the full step whose slot index within their shard is a multiple of ten
            offs = torch.tensor([sp[0] for sp in specs])
            shard_of = torch.bucketize(surv[0], offs[1:], right=True)
            mask_few = ((surv[0] - offs[shard_of]) % 10 == 0)
            assert torch.equal(small[0], surv[0][mask_few]) and torch.equal(torch.stack(small[1:]), torch.stack(surv[1:])[:, mask_few])
        if rank == 0:
            assert torch.equal(dense[0], torch.arange(n_total)) and torch.equal(torch.stack(dense[1:]), XYO)
            q.put((stats.numpy(), XYO.numpy(), alive.numpy(), [t.numpy() for t in surv]))
        else:
            assert XYO is None and alive is None and surv is None and dense is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [4001])
def test_two_rank_shards_match_single_process(n_total):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from twin_backend import TwinBackend
    from attosecondraytracing_amd import _lib
    old = _lib._BACKEND
    _lib._BACKEND = TwinBackend()
    try:
        r, last, det = _run_shard(0, 1, n_total)
        ref_stats = r["stats_dev"].numpy()
        ref = np.stack([r["X"].numpy(), r["Y"].numpy(), r["opl"].numpy()])
        ref_alive = last.alive.numpy()
    finally:
        _lib._BACKEND = old
    ctx = tmp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(rk, 2, port, n_total, q)) for rk in range(2)]
    for p in procs:
        p.start()
    stats, XYO, alive, surv = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(alive, ref_alive)
    m = ref_alive.astype(bool)
    # survivor-only gather == the single-process Detector read-out of the survivors, numbers included
    assert np.array_equal(surv[0], np.nonzero(m)[0]) and surv[0].dtype == np.int64
    assert np.array_equal(np.stack(surv[1:]), ref[:, m])
    assert 0 < m.sum() < n_total
    assert np.array_equal(XYO[:, m], ref[:, m])            # bit-exact: same per-ray code, same inputs
    assert stats[0] == ref_stats[0]
    for k in (2, 3, 4, 5, 12, 13):
        assert stats[k] == ref_stats[k]                    # min / max are exact
    for k in (1, 6, 7, 8, 11):
        assert abs(stats[k] - ref_stats[k]) <= 1e-12 * abs(ref_stats[k])   # sums: order of addition differs


def test_shard_ranges_partition():
    from attosecondraytracing_amd.sharding import shard_range
    for n in (0, 1, 7, 8, 1000, 10 ** 8 + 3):
        for w in (1, 2, 3, 8):
            edges = [shard_range(n, r, w) for r in range(w)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in edges) - min(b - a for a, b in edges) <= 1


def test_sample_slots_are_exact_integers():
    """Evenly spaced sample slots for shards above 2^24 slots: a float32 linspace rounds n-1 up to n there (reading
    one element past the arrays in art_exchange_pack); the integer form never does."""
    from attosecondraytracing_amd.sharding import sample_slots
    dev = torch.device("cpu")
    for n in (20_000_000, 40_000_000, 2 ** 25, 2 ** 25 + 3, 2 ** 24 + 1, 10 ** 8 + 7, 20001):
        for k in (2, 2500, 20000):
            s = sample_slots(n, k, dev)
            assert s.dtype == torch.int64 and s.numel() == k
            assert int(s[0]) == 0 and int(s[-1]) == n - 1 and int(s.max()) < n
            assert bool((s[1:] > s[:-1]).all())
    assert torch.equal(sample_slots(5, 10, dev), torch.arange(5))
    assert torch.equal(sample_slots(7, 1, dev), torch.zeros(1, dtype=torch.int64))


def test_exchange_refuses_out_of_range_slots():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from twin_backend import TwinBackend
    from attosecondraytracing_amd import sharding
    ex = sharding.Exchange(TwinBackend(), 2 ** 25 + 3, sample=20000)
    assert int(ex.slots.max()) == 2 ** 25 + 2


def test_strided_shards_reassemble_to_the_single_process_result():
    """Strided shards (rank r traces rays r, r + N, ...): every rank sees the whole aperture, the interleaved read-outs
    equal the single-process run bit for bit, and the shards lose (almost) the same number of rays at the mask --
    contiguous shards of the radially ordered Vogel source do not."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from twin_backend import TwinBackend
    from attosecondraytracing_amd import _lib, sharding, ModuleGeometry as mgeo
    from attosecondraytracing_amd.bundle import RayBundle
    import ART.ModuleProcessing as mp
    import ART.ModuleDetector as mdet
    old = _lib._BACKEND
    be = _lib._BACKEND = TwinBackend()
    try:
        n_total, world = 6000, 3
        chain = _scene(n_total)
        det = mdet.Detector(np.zeros(3), np.array([1900.0, 30.0, 0.0]), np.array([-0.9, -0.1, 0.2]))
        rot = mgeo.rotation_matrix(np.array([0.0, 0.0, 1.0]), np.array([1.0, 0.0, 0.0]))

        def run(first, step, n):
            src = RayBundle.allocate(n, backend=be)
            be.make_source(0, 0.025, rot, np.zeros(3), first, n, n_total, src.view(), step=step)
            out = mp.RayTracingCalculation(src, chain.optical_elements)
            r = det.readout(out[-1], sync=False)
            return torch.stack([r["X"], r["Y"], r["opl"]]), out[-1].alive.clone()

        full, full_alive = run(0, 1, n_total)
        for layout in ("blocks", "strided"):
            specs = [sharding.shard_spec(n_total, rk, world, layout) for rk in range(world)]
            assert sum(n for _, _, n in specs) == n_total
            parts = [run(*sp) for sp in specs]
            XYO = sharding.assemble(torch.stack([p[0] for p in parts]), layout)
            alive = sharding.assemble(torch.stack([p[1] for p in parts]), layout)
            assert torch.equal(alive, full_alive)
            m = full_alive.bool()
            assert torch.equal(XYO[:, m], full[:, m])
            # survivor records of every shard (art_pack_survivors' layout), merged into global ray order
            sgs = [sharding.SurvivorGather(be, sp[2], 1, 0, specs=[sp]) for sp in specs]
            recs = []
            for g, p in zip(sgs, parts):
                g.start(0, p[0][0], p[0][1], p[0][2], p[1])
                recs.append(g.result(0)[0])
            num = torch.cat([t[0] for t in recs])
            order = torch.argsort(num, stable=True)
            assert torch.equal(num[order], torch.nonzero(full_alive).reshape(-1))
            assert torch.equal(torch.stack([torch.cat([t[k] for t in recs])[order] for k in (1, 2, 3)]), full[:, m])
            counts = [int(p[1].sum()) for p in parts]
            if layout == "strided":
                assert max(counts) - min(counts) <= 2, counts          # balanced
            else:
                assert max(counts) - min(counts) > 1000, counts        # the outer block is (almost) all stopped
        assert sharding.shard_spec(10, 3, 4, "strided") == (3, 4, 2) and sharding.shard_spec(2, 3, 4, "strided")[2] == 0
    finally:
        _lib._BACKEND = old


def test_survivor_gather_refuses_ray_numbers_beyond_int32():
    """The survivor record carries the ray number as int32: a job whose numbers do not fit is refused when the
    gather is set up, not truncated on the device."""
    from attosecondraytracing_amd import sharding

    class NoBackend:            # the check comes before anything touches the device
        device = "cpu"

        def survivor_bytes(self, count, dense):
            raise AssertionError("not reached")
    with pytest.raises(ValueError, match="int32"):
        sharding.SurvivorGather(NoBackend(), 10, 2, 0, specs=[(0, 1, 10), (2 ** 31 - 5, 1, 10)])
