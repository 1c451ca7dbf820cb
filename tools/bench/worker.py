"""One rank of bench.py: builds the workload, runs the timed regions, and (rank 0) prints the JSON line."""
import json
import os
import sys
import time

from .launcher import BoxState, log, ROOT
from . import workloads, baselines, roofline as rl

SETTLE_SECONDS = 2.0    # device-busy time before the `value_sustained` region (long enough for clocks AND power management to settle)
EVENT_STEPS = 20        # passes whose launches are bracketed by HIP events for roofline.kernel_ms


def worker(args):
    import numpy as np
    import torch

    # the contract is ONE JSON line on stdout: route everything libraries print there (RCCL prints a version banner
    # on first use) to stderr until the result line is written
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # started before anything below touches the GPU; the first query describes the box before this run loads it
    boxq = BoxState() if (rank == 0 and os.environ.get("ART_BENCH_BACKEND_HOOK") is None) else None
    box_idle_pending = boxq.ask() if boxq else False
    # ART_BENCH_BACKEND_HOOK="module:function" (TEST HOOK, tests/test_bench_launcher.py): install another backend
    # before the workload starts, so that the launcher and the distributed logic of this file can be exercised by CPU
    # ranks over gloo.  Never set on a GPU box; the product itself has no such switch (attosecondraytracing_amd/_lib.py).
    hook = os.environ.get("ART_BENCH_BACKEND_HOOK")
    on_gpu = hook is None
    if on_gpu:
        torch.cuda.set_device(local)
    # ART_FORCE_DIST=1 runs the multi-rank code path (process group, header exchange, transfers) even with one rank: a way
    # to exercise the RCCL calls on a single-GPU box
    use_dist = env_world > 1 or os.environ.get("ART_FORCE_DIST") == "1"
    world = 1
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = os.environ.get("ART_DIST_BACKEND", "nccl")
        kw = {"device_id": torch.device("cuda", local)} if (on_gpu and backend == "nccl") else {}
        import datetime
        # a rank that never arrives must not hold the others forever: collectives give up after --pg-timeout seconds
        # (the launcher's own wall-clock limit, --time-limit, is the second line of defence)
        dist.init_process_group(backend, rank=rank, world_size=env_world,
                                timeout=datetime.timedelta(seconds=args.pg_timeout), **kw)
        world = dist.get_world_size()       # what RCCL actually saw
    if world != args.gpus:
        log(f"[bench] FATAL: --gpus {args.gpus} but the process group has {world} rank(s)")
        if use_dist:
            dist.destroy_process_group()
        return 3

    def sync():
        if on_gpu:
            torch.cuda.synchronize()

    def barrier():
        if use_dist:
            dist.barrier()

    if hook:
        mod, fn = hook.split(":")
        getattr(__import__(mod), fn)()
    elif rank == 0 or not use_dist:
        import __graft_entry__
        __graft_entry__.ensure_built()       # no-op when libart_hip.so is up to date
    barrier()
    from attosecondraytracing_amd import _lib, sharding
    from attosecondraytracing_amd.graph import SceneProgram
    import ART.ModuleProcessing as mp
    import ART.ModuleDetector as mdet
    be = _lib.get_backend()
    mode = args.mode or mp.DEFAULT_TRACE_MODE

    # ------------------------------------------------------------------ workload
    cfg = args.config
    W = workloads.select(cfg, args.mirrors, args.rays)
    element_lists, src_kind, det_dist, n, label, ignore_defects = (W[k] for k in ("element_lists", "src_kind", "det_dist", "n",
                                                                                  "label", "ignore_defects"))
    n_chains, n_elems = len(element_lists), len(element_lists[0])
    n_total = n * world
    # --shard blocks (default): contiguous index ranges; strided: rank r traces rays r, r + N, ... -- balanced
    # where a mask or an overfilled aperture stops the outer rays of the Vogel spiral (C2, C3)
    first, stride, n_shard = sharding.shard_spec(n_total, rank, world, args.shard)
    assert n_shard == n
    # one resident source shard shared by all chains (OEPlacement gives every chain of a loop list the same source)
    if on_gpu:
        be.count_from = n // 2          # count the full-size launches of the fused kernels from here on (see timed())
    src = workloads.device_source(n, first, n_total, be, src_kind, W["wavelength"], step=stride)
    if on_gpu:
        # one launch with an exactly known byte count (49 B per ray read, every slot alive): what tools/summarize_profile.py
        # calibrates the FETCH_SIZE counter of a profiled run on (k_make_source above does the same for WRITE_SIZE)
        be.bundle_sums(src.view(), None, n)
    batched = n_chains > 1
    # The whole step (trace + read-outs) is replayed from a HIP graph (graph.SceneProgram, the product's compiled-scene
    # path): small bundles are launch-bound without it, and at 1e7 rays it takes the host out of the measurement.
    # `--graph off` issues eager launches.
    use_graph = on_gpu and args.graph in ("on", "auto") and mode == "chain"

    # detectors: placed once (untimed) from the mean ray of each chain's last bundle, like ARTmain.setup_detector
    if batched:
        outs0 = mp.RayTracingCalculationMany([src] * n_chains, element_lists, IgnoreDefects=ignore_defects)
    else:
        outs0 = [mp.RayTracingCalculation(src, element_lists[0], IgnoreDefects=ignore_defects, mode=mode)]
    # Rank 0 places them (its shard holds the innermost rays of the Vogel spiral, so it always has survivors) and
    # broadcasts the poses: every rank reads out on the same detector planes, as a single-process run would.
    dets, entering, surv_last, live = [], 0, [], []
    for els, out in zip(element_lists, outs0):
        live.append([len(o) for o in out])         # survivors after every element of this chain (rank-local)
        det = mdet.Detector(np.asarray(els[-1].position, dtype=float))
        if rank == 0:
            det.autoplace(out[-1], det_dist)
        dets.append(det)
        entering += n + sum(len(o) for o in out[:-1])
        surv_last.append(len(out[-1]))
    if use_dist:
        poses = [[(d.centre, d.normal, d.refpoint) for d in dets]]
        dist.broadcast_object_list(poses, src=0)
        dets = [mdet.Detector(np.asarray(rp, float), np.asarray(c, float), np.asarray(nn, float)) for c, nn, rp in poses[0]]
        # index-range shards of a radially ordered source do not lose the same number of rays at a mask: the job's
        # units per step are the sum over ranks
        t = torch.tensor([entering, surv_last[-1]], dtype=torch.int64, device=be.device)
        dist.all_reduce(t)
        inter_per_step_job, surv_last_job = int(t[0].item()), int(t[1].item())
    else:
        inter_per_step_job, surv_last_job = int(entering), surv_last[-1]
    inter_per_step_rank = int(entering)
    del outs0

    lite = args.readout == "lite"      # the fused tail with 8 of its 22 statistics (ArtChainReadout.lite): a measurement option

    def readouts(outs):
        return [d.readout(o[-1], sync=False, lite=lite) for d, o in zip(dets, outs)]

    # the detectors are in place before the timed region, so their read-out rides on the tracing launch (the ray is
    # still in registers: 24 B/ray of outputs instead of a second pass that re-reads 57 B/ray); --readout separate
    # launches art_detector_readout on the last bundle instead
    fuse = mode == "chain" and args.readout in ("fused", "auto", "lite")
    # Python's cyclic collector: a full (generation-2) pass walks the ~1e6 objects that importing torch/numpy leaves
    # behind and stops the host for ~40 ms -- once per run, at an arbitrary step.  Everything alive now is moved to the
    # permanent generation; the steps themselves create no reference cycles.
    import gc
    gc.collect()
    gc.freeze()

    # ------------------------------------------------------------------ N > 1: the gather north_star names, in every step
    # `(number:int32, X, Y, path)` of every SURVIVING ray of the analysed (last) chain to rank 0.  A shard that loses
    # nothing writes its read-out straight into the gather's send buffers (zero-copy: two programs, one per buffer set,
    # whose fused read-outs target the set's dense sections); a masked shard keeps its own read-out arrays and packs.
    specs = [sharding.shard_spec(n_total, rk, world, args.shard) for rk in range(world)]
    gather = None
    zero_copy = False
    if use_dist:
        zero_copy = bool(fuse and surv_last[-1] == n and not lite and args.gather_copy == "zero")
        gather = sharding.SurvivorGather(be, n, world, rank, dst=0, buffers=2, specs=specs, zero_copy=zero_copy,
                                         tiles=args.gather_tiles)

    def make_program(targets=None, lo=0, hi=None, **kw):
        """The step's program; with a slot range [lo, hi): one TILE of it (the same output bundles, ranges of them)."""
        hi = n if hi is None else hi
        whole = lo == 0 and hi == n
        s_ = src if whole else src.slots(lo, hi)
        ro_t = None if targets is None else [None] * (n_chains - 1) + [tuple(t_[lo:hi] for t_ in targets)]
        if not whole:
            kw["outputs"] = [[b_.slots(lo, hi) for b_ in outs] for outs in kw["outputs"]]
        return SceneProgram([s_] * n_chains, element_lists, IgnoreDefects=ignore_defects, post=readouts, capture=use_graph,
                            detectors=dets if fuse else None, readout_lite=lite, readout_targets=ro_t, **kw)

    program = None
    if batched or use_graph or (zero_copy and args.gather_tiles > 1):
        program = make_program(gather.targets(0) if zero_copy else None, placement_tries=args.placement_tries)
    programs = [program, program]
    if zero_copy and program is not None:
        programs[1] = make_program(gather.targets(1))
    # --gather-tiles T > 1 (zero-copy shards): each buffer set's step also exists as T tile programs over consecutive slot
    # ranges of the SAME output bundles; the records of tile t leave while tile t + 1 is traced (sharding.SurvivorGather)
    tile_programs, tile_stats = [None, None], None
    if zero_copy and program is not None and args.gather_tiles > 1:
        for b_ in (0, 1):
            tile_programs[b_] = [make_program(gather.targets(b_), *gather.tile_range(t_), outputs=programs[b_].outputs)
                                 for t_ in range(args.gather_tiles) if gather.tile_range(t_)[1] > gather.tile_range(t_)[0]]
        tile_stats = [torch.zeros(24, dtype=torch.float64, device=be.device) for _ in (0, 1)]

    # the same step WITHOUT the intermediate bundles (what ARTmain's lazy history traces: the analysed bundle + its
    # read-out; the rest of the history only when somebody looks at it) -- reported beside `value`, never as `value`
    program_lazy = None
    if (batched or use_graph) and n_elems <= 8 and world == 1 and not use_dist:
        program_lazy = SceneProgram([src] * n_chains, element_lists, IgnoreDefects=ignore_defects, post=readouts,
                                    capture=use_graph, detectors=dets if fuse else None, history=False, readout_lite=lite)

    def trace_and_readout_lazy():
        if program_lazy is not None:
            o = program_lazy.run()
            return o, program_lazy.post_result
        o = [mp.RayTracingCalculation(src, element_lists[0], IgnoreDefects=ignore_defects, mode=mode, history=False,
                                      detector=dets[0] if fuse else None, readout_lite=lite)]
        return o, readouts(o)

    def trace_and_readout(b=0):
        if programs[b] is not None:
            o = programs[b].run()
            return o, programs[b].post_result
        o = [mp.RayTracingCalculation(src, element_lists[0], IgnoreDefects=ignore_defects, mode=mode,
                                      detector=dets[0] if fuse else None, readout_lite=lite)]
        return o, readouts(o)

    # the secondary exchange: ONE all-gather of every shard's 24 statistics + an evenly spaced 20000-ray sample
    exchange = sharding.Exchange(be, n, sample=20000) if use_dist else None
    sample_k = exchange.k if exchange else 0
    state = {"stats": None, "sample": None, "step": 0, "xstep": 0, "gather_bytes": 0}

    def exchange_drain():
        # fold the exchange that is still in flight (the last step's) and start the numbering afresh
        if exchange is not None and state["xstep"] > 0:
            state["stats"], state["sample"] = exchange.finish((state["xstep"] - 1) % 2)
            state["xstep"] = 0

    def step(kind):
        """kind: "plain" (N = 1), "gather" (N > 1 headline: + the survivor gather), "stats" (N > 1 secondary: + the
        statistics / sample all-gather).  Nothing in a step blocks the host: launches queue up like a training loop's."""
        if kind == "gather":
            b = state["step"] % 2
            gather.acquire(b)            # the stream waits for set b's previous transfers: its send buffer is rewritten now
            if tile_programs[b] is not None and gather.tiling(b):
                parts = []
                for t_, prog in enumerate(tile_programs[b]):
                    prog.run()
                    gather.start_tile(b, t_)                 # tile t's records leave while tile t + 1 is traced
                    parts.append(prog.post_result[-1]["stats_dev"])
                be.exchange_fold(torch.stack(parts).view(-1), len(parts), 24, tile_stats[b])     # the step's statistics
                o = programs[b].outputs
                o[-1][-1].touch()
                tg = gather.targets(b)
                r = [None] * (n_chains - 1) + [{"X": tg[0], "Y": tg[1], "opl": tg[2], "stats_dev": tile_stats[b]}]
                state["tiled_steps"] = state.get("tiled_steps", 0) + 1
            else:
                o, r = trace_and_readout(b)
            state["gather_bytes"] = gather.start(b, r[-1]["X"], r[-1]["Y"], r[-1]["opl"], o[-1][-1].alive, r[-1]["stats_dev"])
            state["step"] += 1
            return o, r
        o, r = trace_and_readout(0)
        if kind == "stats":
            # statistics of every shard + a sample of every shard's read-out (last chain), double-buffered: the all-gather
            # of this step travels while the next step is traced, its result is folded one step later
            b = state["xstep"] % 2
            if state["xstep"] > 0:
                state["stats"], state["sample"] = exchange.finish(1 - b)
            exchange.start(b, r[-1]["stats_dev"], r[-1]["X"], r[-1]["Y"], r[-1]["opl"], o[-1][-1].alive)
            state["xstep"] += 1
        return o, r

    def timed(kind, steps, warmup):
        for _ in range(warmup):
            step(kind)
        exchange_drain()
        if gather:
            gather.drain()
        barrier()
        sync()
        li0 = getattr(be, "counted_launches", 0)
        t0 = time.perf_counter()
        for k in range(steps):
            o, r = step(kind)
        t_enq = time.perf_counter() - t0     # host time to enqueue all steps (diagnostic: host-bound if ~ dt)
        state.setdefault("timed_launches", (li0, getattr(be, "counted_launches", 0)))    # of the FIRST timed region
        exchange_drain()                     # the last step's statistics are folded ...
        if gather:
            gather.drain()                   # ... and every record has landed on rank 0 before the clock stops
        sync()
        barrier()
        dt = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([dt], dtype=torch.float64, device=be.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, t_enq, o, r

    main_kind = "gather" if use_dist else "plain"
    if getattr(args, "preheat_ms", 0.0) > 0 and on_gpu:
        # opt-in (--preheat-ms): untimed steps until the device has been busy that long; a step COUNT from rank 0's clock,
        # the same on every rank (the steps carry communication)
        t_ph = time.perf_counter()
        step(main_kind)
        sync()
        per = max(time.perf_counter() - t_ph, 1e-5)
        cnt = torch.tensor([max(1, int(args.preheat_ms * 1e-3 / per))], dtype=torch.int64, device=be.device)
        if use_dist:
            dist.broadcast(cnt, src=0)
        for _ in range(int(cnt.item())):
            step(main_kind)
    dt, t_enq, o, r = timed(main_kind, args.steps, args.warmup)
    dt_stats = gather_only = None
    if use_dist:
        b_last = (state["step"] - 1) % 2
        if rank == 0:
            parts = gather.result(b_last)
            gathered_counts = [c for c, _ in gather.headers[b_last]]
            assert len(parts) == world and sum(gathered_counts) == surv_last_job, (gathered_counts, surv_last_job)
            # rank 0's own shard: the records of its survivors, in slot order, numbers included, bit for bit
            idx0 = o[-1][-1].index()
            mine = torch.stack([r[-1]["X"], r[-1]["Y"], r[-1]["opl"]]).index_select(1, idx0)
            assert torch.equal(torch.stack(parts[0][1:]).contiguous().view(torch.int64), mine.contiguous().view(torch.int64))
            assert torch.equal(parts[0][0], specs[0][0] + specs[0][1] * idx0)
            num_all = gather.assemble(b_last)[0]
            assert bool((num_all[1:] > num_all[:-1]).all()) and int(num_all[-1]) < n_total     # global ray order, each ray once
        gstats = gather.stats(b_last).clone()          # the global statistics rode on the header exchange
        # the transfers alone, back to back on resident buffers (no trace, no pack): what the links deliver
        reps = max(3, min(20, args.steps))
        sizes = list(gather.sizes[b_last])
        barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            gather._issue(b_last, sizes)
            gather._wait(b_last)
        sync()
        barrier()
        gather_only = (time.perf_counter() - t0) / reps
        dt_stats, _, o, r = timed("stats", args.steps, args.warmup)
        if rank == 0:
            S = state["sample"]
            assert S.shape == (world, sample_k, 4)
            own = torch.stack([r[-1]["X"], r[-1]["Y"], r[-1]["opl"]]).index_select(1, exchange.slots).T
            assert torch.equal(S[0][:, 0:3].contiguous().view(torch.int64), own.contiguous().view(torch.int64))   # own part of the sample
            # both routes to the global statistics give the same bits (same fold, rank order) -- unless the step was traced in
            # tiles: their partial sums are folded tile by tile (count, minima and maxima stay exact)
            if state.get("tiled_steps", 0) == 0:
                assert torch.equal(gstats.view(torch.int64), state["stats"].view(torch.int64))
            else:
                exact = [0, 2, 3, 4, 5, 12, 13]
                assert torch.equal(gstats[exact], state["stats"][exact])
                assert torch.allclose(gstats, state["stats"], rtol=1e-11, atol=0.0)
    # ------------------------------------------------------------------ kernel durations (HIP events on the launch stream)
    # Right after the timed region(s), on the same resident data: EVENT_STEPS more passes of trace + read-out with every
    # launch bracketed by HIP events recorded on the launch stream.  NOT inside the timed region: every timing event is a
    # barrier packet that keeps the next kernel from overlapping the previous one's tail (0.09 ms per 0.75-ms step).
    # The roofline's `frac` is the TIMED REGION's (bytes x launches / ms_per_step); these give `frac_post_region`.
    kernel_ms = readout_ms = None
    launches = 1
    if on_gpu:
        be.trace_events, be.readout_events = [], []
        for _ in range(EVENT_STEPS):
            if program is not None:
                program._launch()
            else:
                trace_and_readout()
        sync()
        tr_ev, ro_ev = be.trace_events, be.readout_events
        be.trace_events, be.readout_events = None, None
        launches = max(1, len(tr_ev) // EVENT_STEPS)
        kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in tr_ev]))       # average duration of one trace launch
        if ro_ev:
            readout_ms = float(np.mean([a.elapsed_time(b) for a, b in ro_ev]))  # one read-out (kernel + 24-slot fold)

    # The contract's region above starts W steps after an idle device.  An MI355X needs ~40 ms of load to reach its
    # sustained clocks.  `value` stays what the contract defines; the SAME K steps timed again once the device has been
    # busy for SETTLE_SECONDS are reported beside it as `value_sustained`.
    dt_sus = kernel_ms_sus = None
    if on_gpu:
        # a step COUNT, derived from the rank-reduced dt: identical on every rank (the steps carry communication)
        for _ in range(max(1, int(np.ceil(SETTLE_SECONDS / (dt / args.steps))))):
            step(main_kind)
        if gather:
            gather.drain()
        sync()
        dt_sus, _, o, r = timed(main_kind, args.steps, 0)
        be.trace_events = []
        for _ in range(EVENT_STEPS):
            if program is not None:
                program._launch()
            else:
                trace_and_readout()
        sync()
        ev_sus, be.trace_events, be.readout_events = be.trace_events, None, None
        kernel_ms_sus = float(np.mean([a.elapsed_time(b) for a, b in ev_sus]))
    stats_host = (gather.stats((state["step"] - 1) % 2) if use_dist else r[-1]["stats_dev"]).cpu().numpy()
    assert stats_host[0] == surv_last_job and np.isfinite(stats_host[1]), (stats_host[0], surv_last_job)

    dt_lazy = None
    if on_gpu and not use_dist and (program is None or program_lazy is not None):
        for _ in range(args.warmup):
            ol, rl_ = trace_and_readout_lazy()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ol, rl_ = trace_and_readout_lazy()
        sync()
        dt_lazy = time.perf_counter() - t0
        # the analysed bundle and its read-out are the full-history step's, bit for bit
        assert torch.equal(ol[-1][-1].alive, o[-1][-1].alive)
        lv_ = o[-1][-1].alive.bool()
        assert torch.equal(ol[-1][-1].data[:, lv_].view(torch.int64), o[-1][-1].data[:, lv_].view(torch.int64))
        assert torch.equal(rl_[-1]["stats_dev"].view(torch.int64), r[-1]["stats_dev"].view(torch.int64))
        del ol, rl_
    # Box state UNDER LOAD: one rocm-smi query runs while the device keeps tracing; clocks, power and partition modes go
    # on the line beside the numbers.
    box = None
    if on_gpu and rank == 0 and not use_dist and boxq is not None and boxq.p is not None:
        box = {"idle_before_run": boxq.answer(local) if box_idle_pending else None}
        if boxq.ask():
            t_end = time.perf_counter() + 10.0
            while not boxq.ready() and time.perf_counter() < t_end:
                for _ in range(50):
                    step("plain")
                sync()
            box["under_load"] = boxq.answer(local)
    if boxq is not None:
        boxq.close()
    if rank == 0:
        value = inter_per_step_job * args.steps / dt
        peers = [sz for rk, sz in enumerate(gather.sizes[(state["step"] - 1) % 2]) if rk != 0] if use_dist else []
        res = {
            "metric": "ray-surface intersections/s", "value": value, "unit": "intersections/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "preheat_ms": float(getattr(args, "preheat_ms", 0.0)),
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic" if on_gpu else f"synthetic -- TEST HOOK {hook}: CPU ranks, NOT a measurement",
            "config": {"workload": f"{label}; {n} rays/GPU x {n_elems} elements x {n_chains} chain(s) = "
                                   f"{inter_per_step_rank} intersections/step on rank 0, {inter_per_step_job} on all "
                                   f"{world} rank(s); full per-element history",
                       "name": cfg, "rays_per_gpu": n, "elements": n_elems, "chains": n_chains,
                       "trace_mode": "scene (one launch for all chains)" if program is not None else mode,
                       "hip_graph": bool(use_graph), "world_size_seen": world, "shard_layout": args.shard,
                       "output_placement": "first allocation (the placement look is opt-in: --placement-tries N)"
                       if program is None or program.placement is None else dict(
                           program.placement, note="OPT-IN (--placement-tries): the program allocated `tries` candidate blocks "
                           "for its output bundles, timed its own launch into each (launch_ms) and kept the first unless another "
                           "was 3 % faster; gain_vs_first = launch time in the first block / in the chosen one"),
                       "readout": ("fused into the tracing launch" + (" (LITE: count, sum of paths, bounding box, path range only)"
                                                                      if lite else "")) if fuse else "separate launch",
                       "step": "RayTracingCalculation + Detector.readout"
                               + (" + the gather BASELINE.json's north_star names, in EVERY step: (number:int32, X, Y, optical path) of "
                                  "every SURVIVING ray of the analysed chain to rank 0 (28 B per survivor; 24 B from shards that lose "
                                  "nothing, whose read-out writes straight into the send buffer: zero-copy) as point-to-point transfers, "
                                  "one per peer into the root (the root's own shard is not copied), behind ONE 208-byte all-gather of "
                                  "every shard's count and 24 read-out statistics; sizes predicted from the counts of two steps earlier, "
                                  "double-buffered behind the next step's tracing: no step blocks the host" if use_dist else ""),
                       "step_stats_exchange": None if not use_dist else
                       f"RayTracingCalculation + Detector.readout + ONE all-gather of every shard's 24 statistics and a "
                       f"{sample_k * world}-ray sample of the read-out, folded on the device (value_stats_exchange: what a caller "
                       "that plots a sample and prints global statistics needs; NOT the headline)",
                       "step_full_gather": None if not use_dist else "= step (the headline carries the gather since round 5)",
                       "gather_zero_copy": None if not use_dist else zero_copy,
                       "gather_tiles": None if not use_dist else args.gather_tiles,
                       "gather_tiled_steps": None if not use_dist else state.get("tiled_steps", 0),
                       "gather_host_syncs": None if not use_dist else gather.host_syncs,
                       "gather_overflows": None if not use_dist else gather.overflows,
                       "gather_dropped": None if not use_dist else gather.dropped,
                       "gather_host_syncs_note": None if not use_dist else
                       "steps of this run that read their own headers synchronously (the first of each gather: nothing to predict "
                       "from) + settled short steps (a shard packed more than was shipped)",
                       "gather_bytes_per_rank": None if not use_dist else state["gather_bytes"],
                       "gather_bytes_per_peer": None if not use_dist else peers,
                       # one xGMI link per peer into the root (the mesh is point to point): a shard's records cannot
                       # arrive faster than bytes / link rate, whatever the tracing does
                       "gather_floor_ms": None if not use_dist else state["gather_bytes"] / (rl.XGMI_LINK_GBS * 1e9) * 1e3,
                       "gather_floor_note": None if not use_dist else
                       f"gather_bytes_per_rank / {rl.XGMI_LINK_GBS:.0f} GB/s (one xGMI link per peer into rank 0; if that figure "
                       "is the link's two directions together, the one-way floor is twice this -- gather_link_gbs_measured answers it)",
                       "gather_only_ms": None if gather_only is None else gather_only * 1e3,
                       "gather_link_gbs_measured": None if (gather_only is None or not peers) else max(peers) / gather_only / 1e9,
                       "gather_root_ingest_gbs_measured": None if (gather_only is None or not peers) else sum(peers) / gather_only / 1e9,
                       "gather_link_note": None if not use_dist else
                       "the step's transfers issued alone, back to back on resident buffers (no trace, no pack): largest peer "
                       "payload / time = what ONE link delivered, sum of the peers' payloads / time = what the root ingested "
                       "(null with one rank: nothing crosses a link)",
                       "gather_survivors": None if not use_dist else surv_last_job,
                       "dist_backend": None if not use_dist else dist.get_backend()},
            "value_sustained": None if dt_sus is None else inter_per_step_job * args.steps / dt_sus,
            "ms_per_step_sustained": None if dt_sus is None else dt_sus / args.steps * 1e3,
            "sustained_note": None if dt_sus is None else
            f"the same {args.steps} steps timed again after the device had been busy for {SETTLE_SECONDS} s more "
            f"(sustained clocks); `value` is the contract's region, {args.warmup} warm-up steps after an idle device",
            "value_lazy_history": None if dt_lazy is None else inter_per_step_job * args.steps / dt_lazy,
            "ms_per_step_lazy_history": None if dt_lazy is None else dt_lazy / args.steps * 1e3,
            "lazy_history_note": None if dt_lazy is None else
            "the same intersections with only the analysed (last) bundle and its read-out written -- the product's lazy "
            "history mode (get_output_rays(history='lazy'), what ARTmain.run_ART uses); the analysed bundle and the 24 "
            "statistics are bit-identical to the full-history step's (asserted in this run); NOT the headline: `value` "
            "writes every per-element bundle",
            "value_full_gather": None if not use_dist else value,
            "ms_per_step_full_gather": None if not use_dist else dt / args.steps * 1e3,
            "value_stats_exchange": None if dt_stats is None else inter_per_step_job * args.steps / dt_stats,
            "ms_per_step_stats_exchange": None if dt_stats is None else dt_stats / args.steps * 1e3,
            "host_enqueue_ms_per_step": t_enq / args.steps * 1e3,
            "box": box,
        }
        if on_gpu:
            res["roofline"], res["roofline_readout"] = _roofline(
                args, cfg, be, program, mode, fuse, lite, src, element_lists, live, n, n_chains, n_elems, launches,
                inter_per_step_rank, dt, kernel_ms, kernel_ms_sus, readout_ms, state)
            res["trace_only_intersections_per_s"] = inter_per_step_rank / (kernel_ms * launches * 1e-3)
        if args.cpu_sample > 0 and on_gpu:
            # N = 1: the CPU baseline (the oracle timed on a bounded sample) and the parity of that sample.  N > 1: the
            # baseline is an N = 1 figure, but `parity` stays on the line -- rank 0 traces a smaller oracle sample on its
            # own device after the timed regions (no communication involved; the other ranks are done)
            n_cpu = args.cpu_sample if world == 1 else min(args.cpu_sample, 200_000)
            v, inter, secs, oracle_result = baselines.cpu_baseline(element_lists[-1], src_kind, det_dist, n_cpu, ignore_defects)
            res["parity"] = baselines.parity_against(oracle_result, element_lists[-1], be, mode, ignore_defects)
        if world == 1 and args.cpu_sample > 0 and on_gpu:
            res["cpu_baseline"] = {"value": v, "unit": "intersections/s", "cores": 1, "kind": "port",
                                   "sample": f"oracle/art_oracle.py (NumPy, batched LAPACK eigvals; single thread) on "
                                             f"{args.cpu_sample} rays x {n_elems} elements of one chain + detector = {inter} "
                                             f"intersections in {secs:.1f} s; host has {os.cpu_count()} cores",
                                   "reference_as_is": baselines.reference_as_is("relay4" if cfg == "relay4" else cfg)}
            try:
                v2, inter2, secs2, thr = baselines.cpu_twin_allcores(element_lists[-1], src_kind, min(args.cpu_sample, 4_000_000),
                                                                     ignore_defects)
                res["cpu_twin_allcores"] = {"value": v2, "unit": "intersections/s", "cores": thr,
                                            "note": f"oracle/twin: the kernels' per-ray code built by g++ -O2 -fopenmp, "
                                                    f"{inter2} intersections in {secs2:.2f} s (best of 3); for scale only"}
            except Exception as e:    # noqa: BLE001 -- an optional extra must never cost the result line
                log(f"[bench] cpu_twin_allcores skipped: {e!r}")
        if use_dist:
            link = res["config"]["gather_link_gbs_measured"]
            log(f"[bench] N = {world}: `value` ({res['value']:.4g} intersections/s, {res['ms_per_step']:.3f} ms per step) is the step "
                f"WITH the gather north_star names -- every surviving ray's (number, X, Y, path) to rank 0 in every step -- and is the "
                f"number the >= 6x scaling target is judged on; it cannot be shorter than one peer's records over one xGMI link "
                f"(gather_floor_ms {res['config']['gather_floor_ms']:.3f} ms at {rl.XGMI_LINK_GBS:.0f} GB/s; measured here: "
                + (f"{link:.1f} GB/s per link, {res['config']['gather_only_ms']:.3f} ms per gather alone" if link else "no peer link with one rank")
                + f").  `value_stats_exchange` ({res['value_stats_exchange']:.4g}) is the step with statistics + sample only.")
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(res), flush=True)
        os.dup2(2, 1)
    if use_dist:
        dist.destroy_process_group()
    return 0


def _roofline(args, cfg, be, program, mode, fuse, lite, src, element_lists, live, n, n_chains, n_elems, launches,
              inter_per_step_rank, dt, kernel_ms, kernel_ms_sus, readout_ms, state):
    """The `roofline` / `roofline_readout` objects of the line.  `frac` is the TIMED REGION's: bytes per launch x launches
    per step / ms_per_step / peak -- bytes COUNTED by rocprofv3's PMC passes of THIS build where such a profile is
    committed (matched by source hash), else the compulsory bytes computed from this run's survivor counts."""
    from tools.source_hash import source_hash
    inter_per_launch = inter_per_step_rank / launches
    defects = any(len(getattr(oe.type, "DeformationList", [])) > 0 for els in element_lists for oe in els)
    tf = "true" if defects else "false"
    # the body is chosen by the library (csrc/art_kernels.hip chain_rpl(): two rays per lane where a mask is part of a launch
    # without defects, or where the chains of a scene share their input): match either name; the label is used when no
    # profile names the kernel
    has_mask = any(oe.type.type == "Mask" for els in element_lists for oe in els)
    rpl_env = os.environ.get("ART_CHAIN_RPL", "")
    shared = program is not None and n_chains > 1          # (bench scenes trace every chain from the one source shard)
    two = (rpl_env == "2" or (rpl_env != "1" and (has_mask or shared))) and not defects
    if program is not None:
        # (a one-element chain with defects on a simple optic runs the body compiled for its kind: k_trace_scene1<kind, waves>)
        kprefix, kpat = f"k_trace_scene{'2' if two else ''}<{tf}", (rf"k_trace_scene(2?<{tf}|1<)" if defects else rf"k_trace_scene2?<{tf}")
    elif mode == "chain" and (n_elems > 1 or fuse):
        kprefix, kpat = f"k_trace_chain{'2' if two else ''}<{tf}", rf"k_trace_chain2?<{tf}"
    else:                       # per-element launches; a one-element chain without read-out is that kernel too
        kprefix, kpat = "k_trace_element<", r"k_trace_element<"
    base = f"relay{args.mirrors}" if cfg == "relay4" else cfg        # (profiles exist for the 4-mirror headline)
    pkey = (base + "_lite") if lite else (base if fuse else base + "_separate")
    tr, tr_note = rl.profiled_traffic(pkey, kpat, n)
    has_w = fuse and src.intensity is not None and not lite
    algo = comp = 0.0
    for lv in live:
        a_, c_ = rl.chain_bytes(lv, n, has_w, fuse, fused_kernels=(mode == "chain" or program is not None))
        algo, comp = algo + a_, comp + c_
    shared_credit = 0.0
    if program is not None and n_chains > 1 and launches == 1:
        # every chain of a bench scene is traced from the ONE source shard: the launch has to read it once
        shared_credit = rl.shared_source_credit(n, n_chains)
        algo, comp = algo - shared_credit, comp - shared_credit
    algo, comp = algo / launches, comp / launches
    step_s = dt / args.steps
    bytes_launch = tr[0] if tr else comp
    basis = "counted" if tr else "compulsory"
    achieved = bytes_launch * launches / step_s / 1e9
    post = bytes_launch / (kernel_ms * 1e-3) / 1e9
    roof = {
        "bound": "hbm", "kernel": tr[2] if tr else kprefix + "...>",
        # `achieved` / `frac`: HBM bytes of the step's launches over the TIMED REGION's own clock (ms_per_step: fold, gaps
        # between launches and the clock ramp of the first steps included) -- counted bytes (rocprofv3 PMC, committed
        # profile OF THIS BUILD) where they exist, else the compulsory bytes computed in this run (`frac_basis`)
        "achieved": achieved, "peak": rl.HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / rl.HBM_PEAK_GBS, "frac_basis": basis,
        "frac_note": "bytes per launch x launches per step / ms_per_step / peak: the timed region itself (round 4 called this "
                     "frac_step); frac_post_region divides by kernel_ms, measured AFTER the region (warmer clocks, no gaps)",
        "traffic": None if tr is None else tr[0],
        "traffic_source": (tr_note or None) if tr is None else tr[1] + " (rocprofv3 PMC, bytes per launch)",
        "compulsory_bytes": comp, "achieved_compulsory": comp * launches / step_s / 1e9,
        "frac_compulsory": comp * launches / step_s / 1e9 / rl.HBM_PEAK_GBS,
        "compulsory_formula": "per chain: n (57 + 8 w) read + sum_k (64 live_k + n) written + fused read-out 24 live_last + "
                              "176 B per workgroup; from this run's survivor counts" +
                              ("" if not shared_credit else f"; the {n_chains} chains read ONE source: 57 n counted once "
                                                            f"(-{shared_credit / 1e6:.0f} MB)"),
        "counted_over_compulsory": None if tr is None else tr[0] / comp,
        "shared_input_note": None if not (program is not None and n_chains > 1) else
        "all chains of this scene read the SAME source bundle: the launch is XCD-grouped (the chains' workgroups of one tile "
        "run on one XCD and share the tile in its L2), so the source crosses the fabric about once instead of once per "
        "chain; compulsory and algorithmic bytes count it once",
        # the ALGORITHMIC bytes of the fused chain (tools/bench/roofline.py): a ray is read once per chain
        "algorithmic_bytes_per_launch": algo, "achieved_algorithmic": algo * launches / step_s / 1e9,
        "frac_algorithmic": algo * launches / step_s / 1e9 / rl.HBM_PEAK_GBS,
        "algorithmic_model": "fused chain: per chain n (57 + 8 w) read once + per element (64 live + n) written + 24 B per "
                             "read-out ray; SURVEY.md 8(d)'s 128 B per intersection + 88 B per read-out ray price one kernel "
                             "per element that re-reads the ray (" +
                             f"{(rl.SURVEY_BYTES_PER_INTERSECTION * inter_per_launch + (rl.SURVEY_BYTES_PER_READOUT_RAY * n * n_chains / launches if fuse else 0.0)) / 1e6:.0f}"
                             " MB per launch by that model: more than the fused kernel moves, so no fraction is formed from it)",
        "frac_post_region": post / rl.HBM_PEAK_GBS, "achieved_post_region": post,
        "kernel_ms": kernel_ms, "launches_per_step": launches, "intersections_per_launch": inter_per_launch,
        "kernel_ms_note": f"POST-REGION: mean of {EVENT_STEPS} event-bracketed launches (trace kernel + its 9-us fold) issued "
                          "right after the timed region(s) -- not inside them, because every timing event is a barrier packet "
                          "that would slow the timed steps; *_sustained: the same after the sustained-load region",
        "timed_region_launches": list(state.get("timed_launches", (0, 0))),
        "timed_region_launches_note": "[first, last) of the full-size fused-kernel launches of this process, in issue order, "
                                      "that lie inside the timed region: cuts a rocprofv3 kernel trace of the same command to "
                                      "it (tools/summarize_profile.py)",
        "kernel_ms_sustained": kernel_ms_sus,
        "frac_sustained": None if kernel_ms_sus is None else bytes_launch / (kernel_ms_sus * 1e-3) / 1e9 / rl.HBM_PEAK_GBS,
        "frac_of_achievable_6300": achieved / 6300.0,
        "source_hash": source_hash(),
    }
    if cfg == "relay4" and fuse and n == 10_000_000 and args.mirrors == 4:
        # the kernel's OTHER roof: SQ counters of this very workload and BUILD (tools/prof_sq.sh + summarize_sq.py)
        import glob
        for sq in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_relay4_sq.json")), reverse=True):
            jsq = json.load(open(sq))
            if jsq.get("source_hash") == roof["source_hash"]:
                jsq.pop("counters", None)
                roof["second_bound"] = dict(jsq, source=os.path.relpath(sq, ROOT))
                break
    if readout_ms is None:
        ro = {"fused": True, "kernel": roof["kernel"],
              "note": "the read-out rides on the tracing launch (art_trace_chain_readout / scene read-outs): its 24 B/ray of "
                      "outputs and the per-workgroup partial statistics are part of that kernel's traffic and time; "
                      "`--readout separate` launches k_detector_readout instead"}
    else:
        tro, _ = rl.profiled_traffic(pkey, r"k_detector_readout", n)
        ro_bytes = 65.0 * n + 24.0 * n
        counted_ro = None if tro is None else tro[0] / (readout_ms * 1e-3) / 1e9
        ro = {"fused": False, "bound": "hbm", "kernel": "k_detector_readout (+ k_readout_final)", "achieved": counted_ro,
              "peak": rl.HBM_PEAK_GBS, "unit": "GB/s", "frac": None if counted_ro is None else counted_ro / rl.HBM_PEAK_GBS,
              "traffic": None if tro is None else tro[0],
              "traffic_source": None if tro is None else tro[1] + " (rocprofv3 PMC, bytes per launch)",
              "achieved_algorithmic": ro_bytes / (readout_ms * 1e-3) / 1e9,
              "frac_algorithmic": ro_bytes / (readout_ms * 1e-3) / 1e9 / rl.HBM_PEAK_GBS,
              "algorithmic_bytes_per_ray": 89.0, "kernel_ms": readout_ms, "launches_per_step": n_chains}
    return roof, ro
