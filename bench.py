#!/usr/bin/env python3
"""bench.py -- headline benchmark: ray-surface intersections/s on MI355X (BASELINE.json metric).

Workload at N=1 ("relay4"): a point source (half-angle 20 mrad) of 1e7 rays through 4 toroidal mirrors
(two f-x-f relays with the C3 scene's toroid: f = 600 mm, 80 deg incidence, 200x30 mm aperture), then the
detector read-out -- 1e7 rays x 4 mirrors = 4e7 ray-surface intersections per step, every ray surviving.
A step = one pass of the hot path over one resident bundle:
    RayTracingCalculation(source, elements)  -> all 4 per-element bundles written (full history, as the API returns)
    Detector.readout(last)                   -> X, Y, optical path per ray + the 16 global statistics
    (N > 1) all-reduce of the 24 read-out statistics over RCCL (the delays need the GLOBAL mean path)
Inputs are resident in HBM before the timed region and so are the results after it: at N = 1 nothing is copied to
the host inside a step, and at N > 1 the per-ray read-out likewise stays in the HBM of the rank that owns the shard.
Collecting it on rank 0 -- the single RCCL gather of (X, Y, optical path, alive), 25 B/ray -- is an on-demand
operation like the D2H copy; `--gather full` puts it into every step (overlapped with the next step's tracing on
RCCL's own stream).  Whatever the mode, the gather is executed and timed after the timed region and reported
(`gather_to_rank0_ms`): it is per-link bound (one xGMI link per peer into the root), i.e. ~3 ms per 1e7-ray shard
against ~1 ms of compute, which is why it is not the default step.
N > 1 is weak scaling: every rank traces its own 1e7-ray shard (index range of a N*1e7-ray source), no collective
on the tracing path.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_INTERSECTION = 128.0   # read 48+8+4, write 48+8+8+4
HBM_PEAK_GBS = 8000.0                 # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_scene(n_mirrors, small_n=1000):
    """Element poses through the product's own OEPlacement (1-ray alignment traces on the GPU)."""
    import ART.ModuleMirror as mmirror
    import ART.ModuleSupport as msupp
    import ART.ModuleProcessing as mp
    R, r = mmirror.ReturnOptimalToroidalRadii(600, 80)
    Tor = mmirror.MirrorToroidal(R, r, msupp.SupportRectangle(200, 30))
    optics = [Tor] * n_mirrors
    dist_ = [600 if (k % 2 == 1 or k == 0) else 1200 for k in range(n_mirrors)]
    inc = [80 if k % 2 == 0 else -80 for k in range(n_mirrors)]
    SP = {"Divergence": 0.02, "SourceSize": 0, "Wavelength": 50e-6, "DeltaFT": 0.5, "NumberRays": small_n}
    chain = mp.OEPlacement(SP, optics, dist_, inc, [0] * n_mirrors, "relay%d" % n_mirrors)
    return chain, (R, r)


def device_source(n, first, n_total, be):
    """Shard [first, first+n) of an n_total-ray point source, generated on the device."""
    from attosecondraytracing_amd.bundle import RayBundle
    b = RayBundle.allocate(n, backend=be)
    b.wavelength = 50e-6
    rot = np.array([[0.0, 0.0, 1.0], [0.0, 1.0, 0.0], [-1.0, 0.0, 0.0]])  # ez -> ex
    from attosecondraytracing_amd import ModuleGeometry as mgeo
    rot = mgeo.rotation_matrix(np.array([0.0, 0.0, 1.0]), np.array([1.0, 0.0, 0.0]))
    be.make_source(0, 0.02, rot, np.zeros(3), first, n, n_total, b.view())
    b.intensity = torch.ones(n, dtype=torch.float64, device=be.device)
    return b


def cpu_baseline(chain, Rr, n_sample):
    """The CPU oracle (NumPy port of the reference algorithm) on a bounded sample of the same workload."""
    from oracle import art_oracle as orc
    R, r = Rr
    B = orc.point_source([0.0, 0.0, 0.0], [1.0, 0.0, 0.0], 0.02, n_sample, 50e-6)
    els = [orc.Element(orc.Optic("torus", orc.Support("rect", [200, 30]), {"R": R, "r": r}, [], "Toroidal Mirror"),
                       np.asarray(oe.position, float), oe.normal, oe.majoraxis) for oe in chain.optical_elements]
    t0 = time.perf_counter()
    out = orc.ray_tracing_calculation(B, els)
    D = orc.detector_autoplace(out[-1], 600.0)
    delays = orc.detector_delays(D, out[-1])
    dt = time.perf_counter() - t0
    inter = n_sample + sum(len(o) for o in out[:-1])
    return inter / dt, inter, dt, {"last": out[-1], "detector": D, "delays": delays}


def parity_against(oracle_result, chain, n_sample, be, mode):
    """The second half of BASELINE.json's metric ("fp64 delay max-rel-err"): the same n_sample-ray workload traced on
    the GPU and compared with what the oracle just computed for the CPU baseline."""
    import ART.ModuleProcessing as mp
    import ART.ModuleDetector as mdet
    from oracle import art_oracle as orc
    ref, Do = oracle_result["last"], oracle_result["detector"]
    src = device_source(n_sample, 0, n_sample, be)
    last = mp.RayTracingCalculation(src, chain.optical_elements, mode=mode)[-1]
    same = bool(np.array_equal(last.numbers(), ref.number))
    det = mdet.Detector(np.asarray(Do.refpoint, float), np.asarray(Do.centre, float), np.asarray(Do.normal, float))
    res = {"rays": n_sample, "survivor_indices_equal": same}
    if same and len(ref) > 0:
        mean_path = float(np.mean(orc.optical_paths(Do, ref)))
        d = np.asarray(det.get_Delays(last))
        res["delay_max_rel_err"] = float(np.abs(d - oracle_result["delays"]).max() / (mean_path / orc.LightSpeed * 1e15))
        res["position_max_rel_err"] = float(np.abs(last.points() - ref.point).max() / max(1.0, np.abs(ref.point).max()))
        res["path_max_rel_err"] = float(np.abs(last.paths_total() - ref.path.sum(axis=1)).max() / mean_path)
        res["note"] = "GPU vs oracle on the cpu_baseline sample; delays and paths relative to the mean optical path"
    return res


def cpu_twin_allcores(chain, n_sample):
    """Second CPU figure, for scale: the kernels' own per-ray code compiled by g++ (oracle/twin, the test suite's CPU
    twin) with OpenMP over rays on all host cores, on a sample of the same workload.  Not the reference's algorithm
    (that is cpu_baseline, the oracle): it shows what the same arithmetic does on the host CPU."""
    import ctypes as C
    import subprocess
    from attosecondraytracing_amd import _abi
    import ART.ModuleProcessing as mp
    from oracle import art_oracle as orc
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    # threads: the cores this process may use, at most 16 (a one-GPU box's share of its host)
    threads = min(len(os.sched_getaffinity(0)), 16)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_twin", "libart_twin.so"))
    try:        # libgomp is usually initialised already (torch links it): set the team size through its API as well
        C.CDLL("libgomp.so.1").omp_set_num_threads(threads)
    except OSError:
        pass
    lib.art_cpu_trace_chain.restype = C.c_int
    lib.art_cpu_trace_chain.argtypes = [C.POINTER(_abi.ArtElementDesc), C.c_int32, C.POINTER(_abi.ArtBundleView),
                                        C.POINTER(_abi.ArtBundleView), C.c_int64]
    B = orc.point_source([0.0, 0.0, 0.0], [1.0, 0.0, 0.0], 0.02, n_sample, 50e-6)
    m = len(chain.optical_elements)

    def block():
        d = np.zeros((8, n_sample))
        a = np.ones(n_sample, dtype=np.uint8)
        v = _abi.ArtBundleView()
        p = d.ctypes.data
        v.ox, v.oy, v.oz, v.dx, v.dy, v.dz, v.path, v.incidence = (p + k * n_sample * 8 for k in range(8))
        v.alive = a.ctypes.data
        return d, a, v
    sd, sa, sv = block()
    sd[0:3], sd[3:6] = B.point.T, B.vector.T
    outs = [block() for _ in range(m)]
    descs = (_abi.ArtElementDesc * m)(*[mp.element_descriptor(oe)[0] for oe in chain.optical_elements])
    views = (_abi.ArtBundleView * m)(*[o[2] for o in outs])
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        rc = lib.art_cpu_trace_chain(descs, m, C.byref(sv), views, n_sample)
        dt = time.perf_counter() - t0
        assert rc == 0
        best = dt if best is None else min(best, dt)
    inter = n_sample + sum(int(o[1].sum()) for o in outs[:-1])
    return inter / best, inter, best, threads


def profiled_traffic(kernel, n, mirrors, mode):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summary of this same workload
    (profiles/rNN_relay<M>_<mode>.json, written by tools/summarize_profile.py from separate --pmc FETCH_SIZE /
    WRITE_SIZE passes with the gfx950 x2 read correction).  None when no matching profile is committed."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_relay{mirrors}_{mode}.json"))):
        try:
            j = json.load(open(f))
        except Exception:
            continue
        stem = "k_trace_chain<false" if mode == "chain" else "k_trace_element<3, false"
        hits = [k for k in j.get("per_launch", {}) if k.startswith(stem)]
        if j.get("rays_per_gpu") == n and hits:
            best = (j["per_launch"][hits[0]]["total_bytes"], os.path.relpath(f, ROOT))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rays", type=int, default=10_000_000, help="rays per GPU")
    ap.add_argument("--mirrors", type=int, default=4)
    ap.add_argument("--mode", default=None, choices=[None, "chain", "element"])
    ap.add_argument("--gather", default="sample", choices=["sample", "ondemand", "full"],
                    help="N > 1, what is exchanged inside every step: 'sample' (default) = the 24 statistics + an evenly "
                         "spaced 20000-ray sample of the read-out (what the plots consume) in ONE all-gather; "
                         "'ondemand' = the statistics only; 'full' = statistics + every ray's read-out gathered to "
                         "rank 0, double-buffered behind the next step.  The full gather is always executed and timed "
                         "once after the steps (gather_to_rank0_ms).")
    ap.add_argument("--cpu-sample", type=int, default=2_000_000, help="rays of the CPU-baseline sample (0 = skip)")
    args = ap.parse_args()

    # the contract is ONE JSON line on stdout: route everything libraries print there (RCCL prints a version banner
    # on first use) to stderr until the result line is written
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    # ART_FORCE_DIST=1 runs the multi-rank code path (process group, all-reduce, gather) even with one rank: a way to
    # exercise the RCCL calls on a single-GPU box
    use_dist = world > 1 or os.environ.get("ART_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(os.environ.get("ART_DIST_BACKEND", "nccl"), rank=rank, world_size=world,
                                device_id=torch.device("cuda", local))
    if world != args.gpus and rank == 0:
        log(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE")

    if rank == 0 or not use_dist:
        import __graft_entry__
        __graft_entry__.ensure_built()       # no-op when libart_hip.so is up to date
    if use_dist:
        dist.barrier()
    from attosecondraytracing_amd import _lib, sharding
    import ART.ModuleProcessing as mp
    import ART.ModuleDetector as mdet
    be = _lib.get_backend()
    mode = args.mode or mp.DEFAULT_TRACE_MODE

    chain, Rr = build_scene(args.mirrors)
    els = chain.optical_elements
    n = args.rays
    n_total = n * world
    lo, hi = sharding.shard_range(n_total, rank, world)
    src = device_source(hi - lo, lo, n_total, be)

    # detector: placed once (untimed) from the mean ray of the last bundle, like ARTmain.setup_detector
    out = mp.RayTracingCalculation(src, els, mode=mode)
    det = mdet.Detector(np.asarray(els[-1].position, dtype=float))
    det.autoplace(out[-1], 600.0)
    entering = [n] + [len(o) for o in out[:-1]]
    surv_last = len(out[-1])
    inter_per_step_rank = int(sum(entering))
    del out

    packs, works = [], [None, None]
    gather_each_step = use_dist and args.gather == "full"
    sample_each_step = use_dist and args.gather == "sample"
    exchange = sharding.Exchange(be, n, sample=20000 if sample_each_step else 0) if use_dist else None
    sample_k = exchange.k if exchange else 0
    last_sample = [None]
    if use_dist:
        import torch.distributed as dist
        # double-buffered gather buffers: the gather of step i (RCCL, its own stream) overlaps the tracing of
        # step i+1; a buffer is reused only after the gather that read it has been waited for
        for _ in range(2):
            pk = {"send": torch.empty((3, n), dtype=torch.float64, device=be.device),
                  "asend": torch.empty(n, dtype=torch.uint8, device=be.device)}
            if rank == 0:
                pk["recv"] = [torch.empty((3, n), dtype=torch.float64, device=be.device) for _ in range(world)]
                pk["arecv"] = [torch.empty(n, dtype=torch.uint8, device=be.device) for _ in range(world)]
            packs.append(pk)
    step_no = [0]
    last_stats = [None]

    def step():
        # nothing in a step blocks the host: launches queue up like the steps of a training loop
        o = mp.RayTracingCalculation(src, els, mode=mode)
        r = det.readout(o[-1], sync=False)
        if use_dist:
            # ONE collective per step: statistics of every shard (+ a sample of every shard's read-out)
            last_stats[0], last_sample[0] = exchange(r["stats_dev"], r["X"], r["Y"], r["opl"], o[-1].alive)
        if gather_each_step:
            b = step_no[0] % 2
            step_no[0] += 1
            if works[b] is not None:
                for w in works[b]:
                    w.wait()
            works[b] = sharding.gather_readout(r["X"], r["Y"], r["opl"], o[-1].alive, 0, packs[b],
                                               sizes=[n] * world, async_op=True)
        return o, r

    def drain():
        for b in range(2):
            if works[b] is not None:
                for w in works[b]:
                    w.wait()
                works[b] = None

    for _ in range(args.warmup):
        step()
    drain()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    be.trace_events = []          # HIP events bracketing every trace launch, on the launch stream
    t0 = time.perf_counter()
    for i in range(args.steps):
        o, r = step()
    t_enq = time.perf_counter() - t0   # host time to enqueue all steps (diagnostic: host-bound if ~ dt)
    drain()                       # every gather has landed on rank 0 before the clock stops
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=be.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    evs, be.trace_events = be.trace_events, None
    gather_ms = None
    if use_dist:
        # the on-demand collection of the last step's read-out on rank 0, timed (second of two runs)
        for rep in range(2):
            torch.cuda.synchronize()
            dist.barrier()
            tg = time.perf_counter()
            XYO, alv = sharding.gather_readout(r["X"], r["Y"], r["opl"], o[-1].alive, 0, packs[0], sizes=[n] * world)
            torch.cuda.synchronize()
            dist.barrier()
            gather_ms = (time.perf_counter() - tg) * 1e3
        if rank == 0 and sample_each_step:
            S = last_sample[0]
            assert S.shape == (world, sample_k, 4)
            own = torch.stack([r["X"], r["Y"], r["opl"]]).index_select(1, exchange.slots).T
            assert torch.equal(S[0][:, 0:3], own)                        # rank 0's own part of the last step's sample
        if rank == 0:
            assert XYO.shape == (3, n * world) and int(alv.sum().item()) > 0
            assert torch.equal(XYO[:, :n], torch.stack([r["X"], r["Y"], r["opl"]]))   # rank 0's own shard, in place
    stats_host = (last_stats[0] if use_dist else r["stats_dev"]).cpu().numpy()
    assert stats_host[0] == surv_last * (world if use_dist else 1) and np.isfinite(stats_host[1])
    launches = 1 if mode == "chain" else args.mirrors
    assert len(evs) == launches * args.steps
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))   # average duration of one trace launch
    trace_ms = kernel_ms * launches
    inter_per_launch = inter_per_step_rank / launches
    achieved = ALGO_BYTES_PER_INTERSECTION * inter_per_launch / (kernel_ms * 1e-3) / 1e9

    if rank == 0:
        kname = "k_trace_chain<false, 5>" if mode == "chain" else "k_trace_element<ART_TORUS, false>"
        tr = profiled_traffic(kname, n, args.mirrors, mode)
        value = inter_per_step_rank * world * args.steps / dt
        res = {
            "metric": "ray-surface intersections/s", "value": value, "unit": "intersections/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"relay{args.mirrors}: point source 20 mrad -> {args.mirrors} toroidal mirrors "
                                   f"(f=600 mm, 80 deg, 200x30 mm) -> detector; {n} rays/GPU x {args.mirrors} mirrors "
                                   f"= {inter_per_step_rank} intersections/GPU/step; full per-element history",
                       "rays_per_gpu": n, "mirrors": args.mirrors, "trace_mode": mode,
                       "step": "RayTracingCalculation + Detector.readout"
                               + (" + ONE RCCL all-gather of the 24 statistics of every shard, folded on the device"
                                  if use_dist else "")
                               + (f" (it also carries a {sample_k * world}-ray sample of the read-out)" if sample_each_step else "")
                               + (" + RCCL gather of the per-ray read-out to rank 0 (overlapped)" if gather_each_step else "")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None if tr is None else tr[0],
                         "traffic_source": None if tr is None else tr[1] + " (rocprofv3 PMC, bytes per launch)",
                         "kernel": kname,
                         "kernel_ms": kernel_ms, "intersections_per_launch": inter_per_launch,
                         "algorithmic_bytes_per_intersection": ALGO_BYTES_PER_INTERSECTION,
                         # what the memory system really delivers: counted bytes / live kernel time.  `frac` above is
                         # on the ALGORITHMIC 128 B per intersection and can exceed 1 for the fused
                         # kernel, which reads a ray once per chain instead of once per element.
                         "hbm_real_GBps": None if tr is None else tr[0] / (kernel_ms * 1e-3) / 1e9,
                         "hbm_real_frac": None if tr is None else tr[0] / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "trace_only_intersections_per_s": inter_per_step_rank / (trace_ms * 1e-3),
            "host_enqueue_ms_per_step": t_enq / args.steps * 1e3,
            "gather_to_rank0_ms": gather_ms,
            "gather_note": None if gather_ms is None else
            f"one gather of the {n * world}-ray read-out (25 B/ray) to rank 0, run after the timed steps; "
            f"{'inside' if gather_each_step else 'not inside'} the timed step"
            + (f"; every timed step all-gathers the statistics and a {sample_k * world}-ray sample (32 B/ray) in one "
               f"collective" if sample_each_step else ""),
        }
        if world == 1 and args.cpu_sample > 0:
            v, inter, secs, oracle_result = cpu_baseline(chain, Rr, args.cpu_sample)
            res["parity"] = parity_against(oracle_result, chain, args.cpu_sample, be, mode)
            res["cpu_baseline"] = {"value": v, "unit": "intersections/s", "cores": 1, "kind": "port",
                                   "sample": f"oracle/art_oracle.py (NumPy, batched LAPACK eigvals; single thread) on "
                                             f"{args.cpu_sample} rays x {args.mirrors} mirrors + detector = {inter} "
                                             f"intersections in {secs:.1f} s; host has {os.cpu_count()} cores"}
            try:
                v2, inter2, secs2, thr = cpu_twin_allcores(chain, min(args.cpu_sample, 4_000_000))
                res["cpu_twin_allcores"] = {"value": v2, "unit": "intersections/s", "cores": thr,
                                            "note": f"oracle/twin: the kernels' per-ray code built by g++ -O2 -fopenmp, "
                                                    f"{inter2} intersections in {secs2:.2f} s (best of 3); for scale only"}
            except Exception as e:    # noqa: BLE001 -- an optional extra must never cost the result line
                log(f"[bench] cpu_twin_allcores skipped: {e!r}")
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(res), flush=True)
        os.dup2(2, 1)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
