#!/usr/bin/env python3
"""bench.py -- ray-surface intersections/s on MI355X (BASELINE.json metric), one JSON line on stdout.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config relay4|C2|C3|C4|C5]

Default workload ("relay4", the headline BASELINE.json's metric is quoted on): a point source (half-angle 20 mrad) of
1e7 rays per GPU through 4 toroidal mirrors (two f-x-f relays with the C3 scene's toroid: f = 600 mm, 80 deg, 200 x 30 mm),
then the detector read-out: 4e7 ray-surface intersections per GPU and step, every ray surviving, full per-element
history written as the API returns it; the step is replayed from a HIP graph (graph.SceneProgram; `--graph off`: eager
launches).  `--config` selects the other BASELINE.json configurations (same JSON contract):
  C2  CONFIG_2toroidals_f-x-f: 11 chains (loop list over the toroid distance) x 1e6 rays x (mask + 2 toroids), traced by
      ONE launch from a device-resident scene table, + 11 read-outs
  C3  CONFIG_2toroidals_twisted: 10 chains (incidence-plane twist) x 1e7 rays x (mask + 2 toroids) + read-outs, one launch
  C4  8-element mixed chain (OAP, plane, 2 toroids, 2 planes, OAP, plane), 1.25e7 rays per GPU (1e8 over 8 GPUs)
  C5  CONFIG_deformed geometry with a 6th-order Zernike defect, IgnoreDefects=False (perturbed normals), 1e7 rays

A step = one pass of the hot path over resident bundles:
    RayTracingCalculation(source, elements)   every per-element bundle written
    Detector.readout(last)                    X, Y, optical path per ray + 24 global statistics (fused reductions)
    N > 1:  + ONE RCCL all-gather per step carrying every shard's 24 statistics and an evenly spaced 20000-ray sample of
            the read-out (the delays are relative to the GLOBAL mean path, ART/ModuleDetector.py:277; the plots draw a
            sample) -- this is `value`;
            and, measured in a second timed region of the same K steps, the same step + ONE RCCL gather of every
            SURVIVING ray's read-out (number:int32, X, Y, optical path: 28 B per survivor, SURVEY.md 8e -- the gather
            BASELINE.json's north_star names; 24 B where a shard lost nothing and its numbers are implicit) to rank 0 in
            every step, double-buffered behind the next step's tracing -- this is `value_full_gather`.
Beside `value` (every per-element bundle written) the line carries `value_lazy_history`: the same step with only the
analysed bundle written, the product's lazy-history mode (what ARTmain uses) -- never the headline.  `roofline.frac` is
counted HBM bytes (committed rocprofv3 PMC profile of THIS build, matched by source hash) over the launch's duration, or
the compulsory bytes computed in the run when no such profile exists (`frac_basis`); `box` holds the clocks / power /
partition mode of the device before the run and under load.
Inputs are resident in HBM before the timed region; nothing is copied to the host inside a step.  N > 1 is weak scaling:
every rank traces its own shard (index range of an N x rays source), no collective on the tracing path.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process is only a LAUNCHER -- it starts N worker
processes of itself (one per GPU, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set, free rendezvous port on 127.0.0.1) before
anything touches the GPU, relays rank 0's JSON line and exits non-zero if a worker fails.  Under torchrun (WORLD_SIZE set)
it is a worker.  A worker fails if the process group's size differs from --gpus.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_INTERSECTION = 128.0   # read 48+8+4, write 48+8+8+4
ALGO_BYTES_READOUT = 88.0             # read 48+8+4, write 8+8+8+4
XGMI_LINK_GBS = 153.0                 # one xGMI link (7 per GPU, point to point)
HBM_PEAK_GBS = 8000.0                 # MI355X HBM3E spec peak (MI355X_MICROARCH.md; ~6300 GB/s is what a copy achieves)
CONFIGS = ("relay4", "C2", "C3", "C4", "C5")
SETTLE_SECONDS = 0.25   # device-busy time before the `value_sustained` region (see worker())
EVENT_STEPS = 20    # passes whose launches are bracketed by HIP events for roofline.kernel_ms (see worker())


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# =========================================================================================== launcher (no GPU, no torch)
def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def multi_process_env(env):
    """What every process of a multi-process GPU job needs in its environment on this pool.
    HSA_ENABLE_IPC_MODE_LEGACY=0: the hosts' kernel driver supports only dmabuf-based IPC.  RCCL opens its peers' buffers
    through hipIpcGetMemHandle / hipIpcOpenMemHandle (and so does any CUDA-tensor sharing between processes); with the
    runtime's LEGACY IPC mode (the default of some ROCr builds) those calls fail with `hipIpcGetMemHandle: invalid
    argument` as soon as two ranks on one node set up their xGMI / P2P transport -- a one-rank group never gets there.
    The image exports the variable already; it is set here too (setdefault: an explicit choice of the caller wins) so that
    a worker started from a scrubbed environment behaves the same.  examples/sharded_trace.py does the same."""
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def launch_workers(n, argv, time_limit=1500.0):
    """Start n workers of this script, one per GPU; relay rank 0's stdout (the JSON line); fail if any worker fails or
    the job exceeds `time_limit` seconds of wall clock (all workers are killed, exit code 4).  The workers are fresh
    child processes: nothing that has touched the GPU is ever re-exec'ed.
    Runs before any torch.cuda / HIP call of this process: nothing here initialises the GPU."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = multi_process_env(dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                                     MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)))
        out = subprocess.PIPE if r == 0 else sys.stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=out))
    # rank 0's stdout is drained while the workers run (a reader thread): a rank 0 that printed more than the pipe holds
    # would otherwise block in write() while this loop waits for it to exit
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    # watch all workers: if one dies, the others would sit in a collective until RCCL's own timeout -- end them at once
    failed = timed_out = False
    t_start = time.monotonic()
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            failed = True
            break
        if time.monotonic() - t_start > time_limit:
            failed = timed_out = True
            log(f"[bench] the {n}-rank job exceeded its wall-clock limit of {time_limit:.0f} s: killing all workers")
            break
        time.sleep(0.2)
    if failed:
        time.sleep(1.0)                      # let the failing rank's traceback reach stderr first
        for p in procs:
            if p.poll() is None:
                p.kill()
    rcs = [p.wait() for p in procs]
    reader.join(timeout=10.0)
    line = b"".join(chunks)
    sys.stdout.write(line.decode(errors="replace"))
    sys.stdout.flush()
    if any(rc != 0 for rc in rcs):
        log(f"[bench] worker exit codes {rcs}: failing")
        return 4 if timed_out else 1
    return 0


# =========================================================================================== box state (read-only queries)
SMI_ARGS = ["rocm-smi", "--showclocks", "--showperflevel", "--showpower", "--showmaxpower", "--showmemorypartition",
            "--showcomputepartition", "--showtemp", "--json"]
_SMI_HELPER = r"""
import subprocess, sys
for line in sys.stdin:
    try:
        out = subprocess.run(%r, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=30).stdout.decode(errors="replace")
    except Exception as e:
        out = "{}"
    sys.stdout.write(out.replace("\n", " ") + "\n")
    sys.stdout.flush()
""" % (SMI_ARGS,)


class BoxState:
    """rocm-smi queries (clocks, power, power cap, partition modes: sysfs reads, no queue touched) through a helper
    process that is started BEFORE this process initialises the GPU: nothing is ever exec'ed from a GPU-initialised
    process (a rule of the pool).  Under rocprofv3 the profiler's preloaded library has initialised the GPU before main()
    runs, so no helper is started there: tools/prof.sh records the box state beside the passes itself."""

    def __init__(self):
        self.p = None
        if "rocprof" in os.environ.get("LD_PRELOAD", "") or "ROCP_TOOL_LIBRARIES" in os.environ:
            return
        try:
            self.p = subprocess.Popen([sys.executable, "-c", _SMI_HELPER], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                      stderr=subprocess.DEVNULL)
        except OSError:
            self.p = None

    def ask(self):
        """Start one query; returns immediately (read it with `answer`)."""
        if self.p is None:
            return False
        try:
            self.p.stdin.write(b"q\n")
            self.p.stdin.flush()
            return True
        except OSError:
            self.p = None
            return False

    def ready(self):
        import select
        return self.p is None or bool(select.select([self.p.stdout], [], [], 0)[0])

    def answer(self, card):
        """Compact dict of the pending query's fields for device `card` (clock levels, power, partitions), or None."""
        if self.p is None:
            return None
        try:
            out = self.p.stdout.readline().decode(errors="replace")
            j = json.loads(out[out.index("{"):])
            c = j.get(f"card{card}", next(iter(j.values())))
            return {k: v for k, v in c.items()
                    if any(t in k.lower() for t in ("clock", "power", "partition", "performance", "temperature (sensor junction)",
                                                    "temperature (sensor memory)"))}
        except Exception as e:    # noqa: BLE001 -- a diagnostic must never cost the result line
            return {"error": repr(e)[:200]}

    def close(self):
        if self.p is not None:
            try:
                self.p.stdin.close()
                self.p.wait(timeout=5)
            except Exception:     # noqa: BLE001
                pass
            self.p = None


# =========================================================================================== scenes
def build_scene(n_mirrors, small_n=1000):
    """relay<M>: element poses through the product's own OEPlacement (1-ray alignment traces on the GPU)."""
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


def scene_c2():
    """examples/CONFIG_2toroidals_f-x-f.py:19-68: mask -> toroid -> toroid at 11 distances (loop list)."""
    import numpy as np
    import ART.ModuleMirror as mmirror, ART.ModuleMask as mmask, ART.ModuleSupport as msupp, ART.ModuleProcessing as mp
    SP = {"Divergence": 50e-3 / 2, "SourceSize": 0, "Wavelength": 50e-6, "DeltaFT": 0.5, "NumberRays": 1000}
    Mask = mmask.Mask(msupp.SupportRoundHole(20, 14e-3 * 500, 0, 0))
    R, r = mmirror.ReturnOptimalToroidalRadii(500, 80)
    Tor = mmirror.MirrorToroidal(R, r, msupp.SupportRectangle(150, 32))
    chains = mp.OEPlacement(SP, [Mask, Tor, Tor], [400, 100, np.linspace(300, 700, 11).tolist()], [0, 80, -80], [0, 0, 0], "C2")
    return [c.optical_elements for c in chains], ("point", 0.025), 500.0


def scene_c3():
    """examples/CONFIG_2toroidals_twisted.py:19-67: mask -> toroid -> toroid, incidence plane twisted in 10 steps."""
    import numpy as np
    import ART.ModuleMirror as mmirror, ART.ModuleMask as mmask, ART.ModuleSupport as msupp, ART.ModuleProcessing as mp
    SP = {"Divergence": 50e-3 / 2, "SourceSize": 0, "Wavelength": 50e-6, "DeltaFT": 0.5, "NumberRays": 1000}
    Mask = mmask.Mask(msupp.SupportRoundHole(30, 41e-3 / 2 * 500, 0, 0))
    R, r = mmirror.ReturnOptimalToroidalRadii(600, 80)
    Tor = mmirror.MirrorToroidal(R, r, msupp.SupportRectangle(200, 30))
    chains = mp.OEPlacement(SP, [Mask, Tor, Tor], [500, 100, 600], [0, 80, -80], [0, 0, np.linspace(-90, 90, 10).tolist()], "C3")
    return [c.optical_elements for c in chains], ("point", 0.025), 600.0


def scene_c4():
    """8 elements mixing OAP, plane and toroidal mirrors (not in the reference; >= 90 % survive)."""
    import ART.ModuleMirror as mmirror, ART.ModuleSupport as msupp, ART.ModuleProcessing as mp
    SP = {"Divergence": 0.03, "SourceSize": 0, "Wavelength": 50e-6, "DeltaFT": 0.5, "NumberRays": 1000}
    oap = mmirror.MirrorParabolic(200, 60, msupp.SupportRound(20))
    plane = mmirror.MirrorPlane(msupp.SupportRound(30))
    R, r = mmirror.ReturnOptimalToroidalRadii(400, 78)
    tor = mmirror.MirrorToroidal(R, r, msupp.SupportRectangle(180, 30))
    oap2 = mmirror.MirrorParabolic(150, 45, msupp.SupportRound(25))
    ch = mp.OEPlacement(SP, [oap, plane, tor, tor, plane, plane, oap2, plane], [200, 150, 250, 800, 650, 120, 140, 60],
                        [0, 45, 78, -78, 30, -30, 0, 20], [0, 0, 0, 0, 90, 0, 0, 45], "C4")
    return [ch.optical_elements], ("point", 0.03), 100.0


def scene_c5():
    """examples/CONFIG_deformed.py:19-57 geometry with a Zernike defect, perturbed normals."""
    import ART.ModuleMirror as mmirror, ART.ModuleSupport as msupp, ART.ModuleProcessing as mp, ART.ModuleDefects as mdef
    S = msupp.SupportRectangle(40, 40)
    M = mmirror.MirrorParabolic(25.4, 0, S)
    Z = mdef.Zernike(S, {(2, 1): 1e-4, (3, 1): 5e-5, (4, 2): 2e-5, (3, 3): -3e-5, (5, 2): 1e-5, (6, 3): -4e-6, (2, 0): 2.5e-5})
    SP = {"Divergence": 0, "SourceSize": 40, "Wavelength": 800e-6, "DeltaFT": 0, "NumberRays": 1000}
    ch = mp.OEPlacement(SP, [mmirror.DeformedMirror(M, [Z])], [15], [0], Description="C5")
    return [ch.optical_elements], ("plane", 20.0), 25.4


def device_source(n, first, n_total, be, kind=("point", 0.02), wavelength=50e-6, step=1):
    """Rays first, first + step, ... (n of them) of an n_total-ray source (point: half-angle; plane: disk radius),
    generated on the device."""
    import numpy as np
    import torch
    from attosecondraytracing_amd.bundle import RayBundle
    from attosecondraytracing_amd import ModuleGeometry as mgeo
    b = RayBundle.allocate(n, backend=be)
    b.wavelength = wavelength
    rot = mgeo.rotation_matrix(np.array([0.0, 0.0, 1.0]), np.array([1.0, 0.0, 0.0]))
    be.make_source(0 if kind[0] == "point" else 1, kind[1], rot, np.zeros(3), first, n, n_total, b.view(), step=step)
    b.intensity = torch.ones(n, dtype=torch.float64, device=be.device)
    return b


# =========================================================================================== CPU baseline (the oracle)
def oracle_elements(elements):
    """The product's OpticalElements as oracle elements (checker side: the oracle is test infrastructure)."""
    import numpy as np
    from oracle import art_oracle as orc
    kinds = {0: "plane", 1: "sphere", 2: "parabola", 3: "torus", 4: "ellipsoid", 5: "cylinder", 6: "mask"}
    sups = {0: "round", 1: "roundhole", 2: "rect", 3: "recthole", 4: "rectrecthole"}
    els = []
    for oe in elements:
        o = oe.type
        base = getattr(o, "Mirror", o)
        kind = kinds[o._abi_kind]
        params = {}
        if kind == "torus":
            params = {"R": base.majorradius, "r": base.minorradius}
        elif kind in ("sphere", "cylinder"):
            params = {"R": base.radius}
        elif kind == "parabola":
            params = {"feff": base.feff, "offaxis_rad": base.offaxisangle, "p": base.p}
        elif kind == "ellipsoid":
            params = {"a": base.a, "b": base.b, "offaxis_rad": base._offaxisangle}
        defects = [orc.ZernikeDefect(dict(d.coefficients), d.R) for d in getattr(o, "DeformationList", [])]
        els.append(orc.Element(orc.Optic(kind, orc.Support(sups[o.support._abi_kind], o.support._abi_params()), params,
                                         defects, o.type), np.asarray(oe.position, float), oe.normal, oe.majoraxis))
    return els


def cpu_baseline(elements, src_kind, det_dist, n_sample, ignore_defects):
    """The CPU oracle (NumPy port of the reference algorithm) on a bounded sample of the same workload (one chain)."""
    import numpy as np
    from oracle import art_oracle as orc
    if src_kind[0] == "point":
        B = orc.point_source([0.0, 0.0, 0.0], [1.0, 0.0, 0.0], src_kind[1], n_sample, 50e-6)
    else:
        B = orc.plane_wave_disk([0.0, 0.0, 0.0], [1.0, 0.0, 0.0], src_kind[1], n_sample, 50e-6)
    els = oracle_elements(elements)
    t0 = time.perf_counter()
    out = orc.ray_tracing_calculation(B, els, IgnoreDefects=ignore_defects)
    D = orc.detector_autoplace(out[-1], det_dist)
    delays = orc.detector_delays(D, out[-1])
    dt = time.perf_counter() - t0
    inter = len(B) + sum(len(o) for o in out[:-1])
    return inter / dt, inter, dt, {"source": B, "last": out[-1], "detector": D, "delays": delays}


def parity_against(oracle_result, elements, be, mode, ignore_defects):
    """Second half of BASELINE.json's metric ("fp64 delay max-rel-err"): the cpu_baseline sample traced on the GPU and
    compared with what the oracle computed for it."""
    import numpy as np
    import ART.ModuleProcessing as mp
    import ART.ModuleDetector as mdet
    from oracle import art_oracle as orc
    from attosecondraytracing_amd.bundle import RayBundle
    ref, Do, B = oracle_result["last"], oracle_result["detector"], oracle_result["source"]
    src = RayBundle.from_arrays(B.point, B.vector, B.number, np.ones(len(B)), 50e-6, backend=be)
    last = mp.RayTracingCalculation(src, elements, IgnoreDefects=ignore_defects, mode=mode)[-1]
    same = bool(np.array_equal(last.numbers(), ref.number))
    det = mdet.Detector(np.asarray(Do.refpoint, float), np.asarray(Do.centre, float), np.asarray(Do.normal, float))
    res = {"rays": len(B), "survivors": int(len(ref)), "survivor_indices_equal": same}
    if same and len(ref) > 0:
        mean_path = float(np.mean(orc.optical_paths(Do, ref)))
        d = np.asarray(det.get_Delays(last))
        res["delay_max_rel_err"] = float(np.abs(d - oracle_result["delays"]).max() / (mean_path / orc.LightSpeed * 1e15))
        res["position_max_rel_err"] = float(np.abs(last.points() - ref.point).max() / max(1.0, np.abs(ref.point).max()))
        res["path_max_rel_err"] = float(np.abs(last.paths_total() - ref.path.sum(axis=1)).max() / mean_path)
        res["note"] = ("GPU vs oracle on the cpu_baseline sample; delays and paths relative to the mean optical path, "
                       "positions to max|ref|; bar 1e-10")
    return res


def cpu_twin_allcores(elements, src_kind, n_sample, ignore_defects):
    """Second CPU figure, for scale: the kernels' own per-ray code compiled by g++ (oracle/twin, the test suite's CPU
    twin) with OpenMP over rays on the host cores, on a sample of the same workload.  Not the reference's algorithm
    (that is cpu_baseline, the oracle): it shows what the same arithmetic does on the host CPU."""
    import ctypes as C
    import numpy as np
    from attosecondraytracing_amd import _abi
    import ART.ModuleProcessing as mp
    from oracle import art_oracle as orc
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    threads = min(len(os.sched_getaffinity(0)), 16)     # a one-GPU box's share of its host
    os.environ["OMP_NUM_THREADS"] = str(threads)
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_twin", "libart_twin.so"))
    try:        # libgomp is usually initialised already (torch links it): set the team size through its API as well
        C.CDLL("libgomp.so.1").omp_set_num_threads(threads)
    except OSError:
        pass
    lib.art_cpu_trace_chain.restype = C.c_int
    lib.art_cpu_trace_chain.argtypes = [C.POINTER(_abi.ArtElementDesc), C.c_int32, C.POINTER(_abi.ArtBundleView),
                                        C.POINTER(_abi.ArtBundleView), C.c_int64]
    if src_kind[0] == "point":
        B = orc.point_source([0.0, 0.0, 0.0], [1.0, 0.0, 0.0], src_kind[1], n_sample, 50e-6)
    else:
        B = orc.plane_wave_disk([0.0, 0.0, 0.0], [1.0, 0.0, 0.0], src_kind[1], n_sample, 50e-6)
    n_sample = len(B)
    m = len(elements)

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
    # descriptors built afresh (not the cached ones, whose defect tables are DEVICE pointers): tables in host memory
    keep = [mp._build_descriptor(oe, ignore_defects, _HostTables()) for oe in elements]
    descs = (_abi.ArtElementDesc * m)(*[k[0] for k in keep])
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


class _HostTables:
    """Stand-in backend for element_descriptor in cpu_twin_allcores: defect tables stay in host memory."""
    device = "cpu"

    @staticmethod
    def from_numpy(a, dtype=None):
        import numpy as np
        import torch
        return torch.from_numpy(np.array(a, copy=True))


def profiled_traffic(config, kernel_pattern, rays):
    """HBM bytes per launch of the kernel whose name matches the regular expression `kernel_pattern` (anchored at the
    start; masked chains launch the two-rays-per-lane bodies k_trace_scene2 / k_trace_chain2), from the newest committed rocprofv3
    PMC summary of this workload (profiles/rNN_<config>*.json, written by tools/summarize_profile.py from separate
    --pmc FETCH_SIZE / WRITE_SIZE passes with the gfx950 x2 read correction) THAT WAS TAKEN ON THIS BUILD: a profile whose
    `source_hash` (csrc/* + include/art_hip.h at profiling time) differs from the tree's is dropped, and the line says so.
    -> (bytes, file, kernel) or None, and a note."""
    import glob
    from tools.source_hash import source_hash
    here = source_hash()
    best, dropped = None, []
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{config}.json")))    # newest round last
    for f in files:
        try:
            j = json.load(open(f))
        except Exception:
            continue
        hits = [k for k in j.get("per_launch", {}) if re.match(kernel_pattern, k)]
        if j.get("rays_per_gpu") == rays and hits:
            if j.get("source_hash") != here:
                dropped.append(f"{os.path.relpath(f, ROOT)} (sources {j.get('source_hash', 'unrecorded')} != {here})")
                continue
            best = (j["per_launch"][hits[0]]["total_bytes"], os.path.relpath(f, ROOT), hits[0])
    note = None
    if best is None and dropped:
        note = "no PMC profile of THIS build: dropped " + "; ".join(dropped)
    return best, note


# =========================================================================================== worker
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
    # ART_FORCE_DIST=1 runs the multi-rank code path (process group, all-gather, gather) even with one rank: a way to
    # exercise the RCCL calls on a single-GPU box
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
    ignore_defects = True
    if cfg == "relay4":
        chain, _ = build_scene(args.mirrors)
        element_lists, src_kind, det_dist = [chain.optical_elements], ("point", 0.02), 600.0
        n = args.rays or 10_000_000
        label = (f"relay{args.mirrors}: point source 20 mrad -> {args.mirrors} toroidal mirrors (f=600 mm, 80 deg, "
                 f"200x30 mm) -> detector")
    elif cfg == "C2":
        element_lists, src_kind, det_dist = scene_c2()
        n = args.rays or 1_000_000
        label = "C2 CONFIG_2toroidals_f-x-f: 11 chains (toroid distance 300..700 mm) x (mask + 2 toroids) -> detector"
    elif cfg == "C3":
        element_lists, src_kind, det_dist = scene_c3()
        n = args.rays or 10_000_000
        label = "C3 CONFIG_2toroidals_twisted: 10 chains (incidence-plane twist -90..90 deg) x (mask + 2 toroids) -> detector"
    elif cfg == "C4":
        element_lists, src_kind, det_dist = scene_c4()
        n = args.rays or 12_500_000
        label = "C4 8-element mixed chain (OAP, plane, 2 toroids, 2 planes, OAP, plane) -> detector; 1e8 rays over 8 GPUs"
    else:
        element_lists, src_kind, det_dist = scene_c5()
        n = args.rays or 10_000_000
        ignore_defects = False
        label = "C5 CONFIG_deformed geometry: parabola f=25.4 mm + 6th-order Zernike defect, perturbed normals -> detector"
    n_chains, n_elems = len(element_lists), len(element_lists[0])
    n_total = n * world
    # --shard blocks (default): contiguous index ranges; strided: rank r traces rays r, r + N, ... -- balanced
    # where a mask or an overfilled aperture stops the outer rays of the Vogel spiral (C2, C3)
    first, stride, n_shard = sharding.shard_spec(n_total, rank, world, args.shard)
    assert n_shard == n
    wl = 800e-6 if cfg == "C5" else 50e-6
    # one resident source shard shared by all chains (OEPlacement gives every chain of a loop list the same source)
    if on_gpu:
        be.count_from = n // 2          # count the full-size launches of the fused kernels from here on (see timed())
    src = device_source(n, first, n_total, be, src_kind, wl, step=stride)
    if on_gpu:
        # one launch with an exactly known byte count (49 B per ray read, every slot alive): what tools/summarize_profile.py
        # calibrates the FETCH_SIZE counter of a profiled run on (k_make_source above does the same for WRITE_SIZE)
        be.bundle_sums(src.view(), None, n)
    batched = n_chains > 1
    # The whole step (trace + read-outs) is replayed from a HIP graph (graph.SceneProgram, the product's compiled-scene
    # path): small bundles are launch-bound without it, and at 1e7 rays -- where the eager step is GPU-bound on a quiet
    # host (0.09 ms of Python per 0.65-ms step) -- it takes the host out of the measurement: on a box whose host was busy
    # the eager relay4 step took 1.25 ms for a 0.72-ms kernel (host_enqueue_ms_per_step 0.67; profiles/r03_experiments.md).
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
    # auto = fused.  Measured per step, fused vs separate (tools/r02_exp19.sh, one box): C2 0.44 vs 0.69 ms,
    # C3 3.97 vs 5.14 ms (many chains, a third to a half of the rays stopped by the mask: the fused tail skips them and
    # replaces 10-11 small read-out launches), relay4 0.77 vs 0.80 ms, C4 1.56-1.58 vs 1.64-1.66 ms (behind eight
    # elements the tail used to cost more than the saved re-read; since its instruction count went down it wins there too).
    fuse = mode == "chain" and args.readout in ("fused", "auto", "lite")
    # Python's cyclic collector: a full (generation-2) pass walks the ~1e6 objects that importing torch/numpy leaves
    # behind and stops the host for ~40 ms -- once per run, at an arbitrary step; in a 20-step timed region of 0.8-ms
    # steps that is the difference between 1.6 and 2.1 ms per step (profiles/HISTORY.md, round 2).  Everything alive now is
    # moved to the permanent generation; the steps themselves create no reference cycles.  Done HERE, before the step's
    # programs are built, so that the device is not left idle for those ~50 ms right in front of the warm-up steps.
    import gc
    gc.collect()
    gc.freeze()
    program = None
    if batched or use_graph:
        program = SceneProgram([src] * n_chains, element_lists, IgnoreDefects=ignore_defects,
                               post=readouts, capture=use_graph, detectors=dets if fuse else None,
                               placement_tries=args.placement_tries, readout_lite=lite)

    # the same step WITHOUT the intermediate bundles (what ARTmain's lazy history traces: the analysed bundle + its
    # read-out; the rest of the history only when somebody looks at it) -- reported beside `value`, never as `value`
    program_lazy = None
    if (batched or use_graph) and n_elems <= 8 and world == 1:
        program_lazy = SceneProgram([src] * n_chains, element_lists, IgnoreDefects=ignore_defects, post=readouts,
                                    capture=use_graph, detectors=dets if fuse else None, history=False, readout_lite=lite)

    def trace_and_readout_lazy():
        if program_lazy is not None:
            o = program_lazy.run()
            return o, program_lazy.post_result
        o = [mp.RayTracingCalculation(src, element_lists[0], IgnoreDefects=ignore_defects, mode=mode, history=False,
                                      detector=dets[0] if fuse else None, readout_lite=lite)]
        return o, readouts(o)

    def trace_and_readout():
        if program is not None:
            o = program.run()
            return o, program.post_result
        o = [mp.RayTracingCalculation(src, element_lists[0], IgnoreDefects=ignore_defects, mode=mode,
                                      detector=dets[0] if fuse else None, readout_lite=lite)]
        return o, readouts(o)

    # ------------------------------------------------------------------ N > 1 exchanges
    exchange = sharding.Exchange(be, n, sample=20000) if use_dist else None
    sample_k = exchange.k if exchange else 0
    specs = [sharding.shard_spec(n_total, rk, world, args.shard) for rk in range(world)]
    gather = sharding.SurvivorGather(be, n, world, rank, dst=0, buffers=2, specs=specs) if use_dist else None
    state = {"stats": None, "sample": None, "step": 0, "xstep": 0, "gather_bytes": 0}

    def exchange_drain():
        # fold the exchange that is still in flight (the last step's) and start the numbering afresh
        if exchange is not None and state["xstep"] > 0:
            state["stats"], state["sample"] = exchange.finish((state["xstep"] - 1) % 2)
            state["xstep"] = 0

    def step(full_gather):
        # nothing in a step blocks the host: launches queue up like the steps of a training loop
        o, r = trace_and_readout()
        if use_dist:
            # ONE collective per step: statistics of every shard + a sample of every shard's read-out (last chain).
            # Double-buffered like the full gather below: the all-gather of this step travels while the next step is
            # traced, its result is folded one step later (exchange_drain() picks up the last one).
            b = state["xstep"] % 2
            if state["xstep"] > 0:
                state["stats"], state["sample"] = exchange.finish(1 - b)
            exchange.start(b, r[-1]["stats_dev"], r[-1]["X"], r[-1]["Y"], r[-1]["opl"], o[-1][-1].alive)
            state["xstep"] += 1
            if full_gather:
                # + ONE gather of every ray's read-out to rank 0, overlapped with the next step's tracing
                state["gather_bytes"] = gather.start(state["step"] % 2, r[-1]["X"], r[-1]["Y"], r[-1]["opl"], o[-1][-1].alive)
                state["step"] += 1
        return o, r

    def timed(full_gather, steps):
        for _ in range(args.warmup):
            step(full_gather)
        exchange_drain()
        if gather:
            gather.drain()
        barrier()
        sync()
        li0 = getattr(be, "counted_launches", 0)
        t0 = time.perf_counter()
        for k in range(steps):
            o, r = step(full_gather)
        t_enq = time.perf_counter() - t0     # host time to enqueue all steps (diagnostic: host-bound if ~ dt)
        state.setdefault("timed_launches", (li0, getattr(be, "counted_launches", 0)))    # of the FIRST timed region
        exchange_drain()                     # the last step's statistics are folded ...
        if gather:
            gather.drain()                   # ... and every gather has landed on rank 0 before the clock stops
        sync()
        barrier()
        dt = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([dt], dtype=torch.float64, device=be.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, t_enq, o, r

    dt, t_enq, o, r = timed(False, args.steps)
    dt_full = None
    if use_dist:
        dt_full, _, o, r = timed(True, args.steps)
        if rank == 0:
            b_last = (state["step"] - 1) % 2
            parts = gather.result(b_last)
            gathered_counts = [c for c, _ in gather.headers[b_last]]
            assert len(parts) == world and sum(gathered_counts) == surv_last_job, (gathered_counts, surv_last_job)
            # rank 0's own shard arrived bit for bit: the records of its survivors, in slot order, numbers included
            idx0 = o[-1][-1].index()
            mine = torch.stack([r[-1]["X"], r[-1]["Y"], r[-1]["opl"]]).index_select(1, idx0)
            assert torch.equal(torch.stack(parts[0][1:]).view(torch.int64), mine.view(torch.int64))
            assert torch.equal(parts[0][0], specs[0][0] + specs[0][1] * idx0)
            num_all = gather.assemble(b_last)[0]
            assert bool((num_all[1:] > num_all[:-1]).all()) and int(num_all[-1]) < n_total     # global ray order, each ray once
            S = state["sample"]
            assert S.shape == (world, sample_k, 4)
            own = torch.stack([r[-1]["X"], r[-1]["Y"], r[-1]["opl"]]).index_select(1, exchange.slots).T
            assert torch.equal(S[0][:, 0:3].contiguous().view(torch.int64), own.contiguous().view(torch.int64))   # own part of the sample
    # ------------------------------------------------------------------ kernel durations (HIP events on the launch stream)
    # Right after the timed region(s), on the same resident data and in the same clock state (before the sustained-load
    # region below: on some boxes 0.25 s of continuous load already ends in power throttling -- one run measured 1.00 ms per
    # step there after 0.73 in the timed region): EVENT_STEPS more passes of trace + read-out with every launch bracketed
    # by HIP events recorded on the launch stream.  They are NOT inside the timed region because every
    # timing event is a barrier packet that keeps the next kernel from overlapping the previous one's tail: bracketing
    # the timed steps themselves costs 0.09 ms per 0.75-ms step (measured).
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
    # sustained clocks: 20 timed steps after 5 warm-up steps run at 0.80 ms, after 50 at 0.72, after 200 at 0.69
    # (round-2 batch 11, profiles/HISTORY.md).  `value` stays what the contract defines; the SAME K steps timed again once the device has
    # been busy for SETTLE_SECONDS are reported beside it as `value_sustained`.
    dt_sus = kernel_ms_sus = None
    if on_gpu:
        # a step COUNT, derived from the rank-reduced dt: identical on every rank (the steps carry a collective)
        for _ in range(max(1, int(np.ceil(SETTLE_SECONDS / (dt / args.steps))))):
            step(False)
        exchange_drain()
        sync()
        saved_w, args.warmup = args.warmup, 0
        dt_sus, _, o, r = timed(False, args.steps)
        args.warmup = saved_w
        # ... and the kernel's duration in THAT clock state (round 2 took `kernel_ms` here), reported beside the one above
        be.trace_events = []
        for _ in range(EVENT_STEPS):
            if program is not None:
                program._launch()
            else:
                trace_and_readout()
        sync()
        ev_sus, be.trace_events, be.readout_events = be.trace_events, None, None
        kernel_ms_sus = float(np.mean([a.elapsed_time(b) for a, b in ev_sus]))
    stats_host = (state["stats"] if use_dist else r[-1]["stats_dev"]).cpu().numpy()
    assert stats_host[0] == surv_last_job and np.isfinite(stats_host[1]), (stats_host[0], surv_last_job)

    # (Everything below runs AFTER the timed regions and the event-bracketed passes: the load loop of the box-state query
    # keeps the device at its 1400-W power cap for about a second, after which the clocks are throttled -- kernel times
    # measured behind it read 10 % high.)
    dt_lazy = None
    if on_gpu and not use_dist and (program is None or program_lazy is not None):
        for _ in range(args.warmup):
            ol, rl = trace_and_readout_lazy()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ol, rl = trace_and_readout_lazy()
        sync()
        dt_lazy = time.perf_counter() - t0
        # the analysed bundle and its read-out are the full-history step's, bit for bit
        assert torch.equal(ol[-1][-1].alive, o[-1][-1].alive)
        lv_ = o[-1][-1].alive.bool()
        assert torch.equal(ol[-1][-1].data[:, lv_].view(torch.int64), o[-1][-1].data[:, lv_].view(torch.int64))
        assert torch.equal(rl[-1]["stats_dev"].view(torch.int64), r[-1]["stats_dev"].view(torch.int64))
        del ol, rl
    # Box state UNDER LOAD (VERDICT r2 #5b: boxes of the pool differ by up to 20 % on this access pattern): one rocm-smi
    # query runs while the device keeps tracing; clocks, power and partition modes go on the line beside the numbers.
    box = None
    if on_gpu and rank == 0 and not use_dist and boxq is not None and boxq.p is not None:
        box = {"idle_before_run": boxq.answer(local) if box_idle_pending else None}
        if boxq.ask():
            t_end = time.perf_counter() + 10.0
            while not boxq.ready() and time.perf_counter() < t_end:
                for _ in range(50):
                    step(False)
                sync()
            box["under_load"] = boxq.answer(local)
    if boxq is not None:
        boxq.close()
    if rank == 0:
        value = inter_per_step_job * args.steps / dt
        res = {
            "metric": "ray-surface intersections/s", "value": value, "unit": "intersections/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
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
                               + (f" + ONE RCCL all-gather of every shard's 24 statistics and a {sample_k * world}-ray sample "
                                  f"of the read-out, folded on the device" if use_dist else ""),
                       "step_full_gather": None if not use_dist else
                       "the same + ONE RCCL gather of every SURVIVING ray's read-out (number:int32, X, Y, optical path; 28 B "
                       "per survivor, 24 B in shards that lost nothing) to rank 0 in every step, double-buffered behind the "
                       "next step's tracing (value_full_gather); the gather's size is predicted from the counts of two steps "
                       "earlier (a 16-byte header all-gather per step that nobody waits for), so no step blocks the host: "
                       "three collectives per step in all (statistics all-gather, header all-gather, payload gather)",
                       "gather_host_syncs": None if not use_dist else gather.host_syncs,
                       "gather_overflows": None if not use_dist else gather.overflows,
                       "gather_host_syncs_note": None if not use_dist else
                       "steps of this run whose gather read its own headers synchronously (the very first one: nothing to "
                       "predict from) + gathers re-issued because a shard packed more than predicted",
                       "gather_bytes_per_rank": None if not use_dist else state["gather_bytes"],
                       # one xGMI link per peer into the root (the mesh is point to point): a shard's records cannot
                       # arrive faster than bytes / link rate, whatever the tracing does
                       "gather_floor_ms": None if not use_dist else state["gather_bytes"] / (XGMI_LINK_GBS * 1e9) * 1e3,
                       "gather_floor_note": None if not use_dist else
                       f"gather_bytes_per_rank / {XGMI_LINK_GBS:.0f} GB/s (one xGMI link per peer into rank 0; if that figure "
                       "is the link's two directions together, the one-way floor is twice this); a step of value_full_gather "
                       "cannot be shorter than max(trace, this)",
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
            "value_full_gather": None if dt_full is None else inter_per_step_job * args.steps / dt_full,
            "ms_per_step_full_gather": None if dt_full is None else dt_full / args.steps * 1e3,
            "host_enqueue_ms_per_step": t_enq / args.steps * 1e3,
            "box": box,
        }
        if on_gpu:
            inter_per_launch = inter_per_step_rank / launches
            defects = any(len(getattr(oe.type, "DeformationList", [])) > 0 for els in element_lists for oe in els)
            # the body is chosen by the library (two rays per lane for chains with a mask): match either name
            tf = "true" if defects else "false"
            # (the library's rule, csrc/art_kernels.hip chain_rpl(): ART_CHAIN_RPL=1|2, else two rays per lane exactly where a
            # mask is part of a launch without defects -- used for the LABEL when no profile names the kernel)
            has_mask = any(oe.type.type == "Mask" for els in element_lists for oe in els)
            rpl_env = os.environ.get("ART_CHAIN_RPL", "")
            two = (rpl_env == "2" or (rpl_env != "1" and has_mask)) and not defects
            if program is not None:
                kprefix, kpat = f"k_trace_scene{'2' if two else ''}<{tf}", rf"k_trace_scene2?<{tf}"
            elif mode == "chain" and (n_elems > 1 or fuse):
                kprefix, kpat = f"k_trace_chain{'2' if two else ''}<{tf}", rf"k_trace_chain2?<{tf}"
            else:                       # per-element launches; a one-element chain without read-out is that kernel too
                kprefix, kpat = "k_trace_element<", r"k_trace_element<"
            # profiles/r0N_<config>.json: the configuration as bench runs it by default; the other read-out mode is
            # profiled as r0N_<config>_fused.json / _separate.json
            auto_fuse = True
            base = f"relay{args.mirrors}" if cfg == "relay4" else cfg        # (profiles exist for the 4-mirror headline)
            pkey = (base + "_lite") if lite else (base if fuse == auto_fuse else base + ("_fused" if fuse else "_separate"))
            tr, tr_note = profiled_traffic(pkey, kpat, n)
            # 128 B per intersection, + 88 B per ray of read-out when that rides on the same launch
            algo_bytes = ALGO_BYTES_PER_INTERSECTION * inter_per_launch + (ALGO_BYTES_READOUT * n * n_chains / launches if fuse else 0.0)
            algo = algo_bytes / (kernel_ms * 1e-3) / 1e9
            # What the launch(es) of one step MUST move, computed here from the run's own survivor counts (no profile): every
            # chain reads its source once -- 7 fp64 streams + the alive byte = 57 B per slot, + 8 B of weight with a fused
            # read-out -- and writes, per element, 64 B per LIVE slot (8 fp64 streams; pairs of dead slots are dropped by
            # the range check) + the alive byte of every slot; a fused read-out adds 24 B per surviving ray and 22 doubles
            # per workgroup of partial statistics.  Per-element launches (--mode element) re-read every bundle.
            has_w = fuse and src.intensity is not None and not lite
            comp = 0.0
            for lv in live:
                if mode == "chain" or program is not None:
                    comp += n * (57.0 + (8.0 if has_w else 0.0)) + sum(64.0 * x + n for x in lv)
                else:
                    comp += sum(57.0 * n + 64.0 * x + n for x in lv)
                if fuse:
                    comp += 24.0 * lv[-1] + 176.0 * ((n + 255) // 256)
            comp /= launches
            compulsory = comp / (kernel_ms * 1e-3) / 1e9
            # what the kernel moves by construction: every chain reads its source once (57 B/slot) and writes 65 B per
            # slot and element (dead slots: only the alive byte) -- the PMC counters agree with it to 0.1 % on relay4
            counted = None if tr is None else tr[0] / (kernel_ms * 1e-3) / 1e9
            basis = "counted" if counted is not None else "compulsory"
            res["roofline"] = {
                "bound": "hbm", "kernel": tr[2] if tr else kprefix + "...>",
                # `achieved`/`frac`: COUNTED HBM bytes (rocprofv3 PMC, committed profile OF THIS BUILD) / live kernel time /
                # peak -- what the memory system really delivers; without such a profile, the compulsory bytes computed in
                # this run (`frac_basis` says which).  The fused kernel reads a ray once per chain, so it moves fewer
                # bytes than the 128 B/intersection of SURVEY 8(d): that algorithmic figure is kept beside it.
                "achieved": counted if counted is not None else compulsory, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (counted if counted is not None else compulsory) / HBM_PEAK_GBS, "frac_basis": basis,
                "traffic": None if tr is None else tr[0],
                "traffic_source": (tr_note or None) if tr is None else tr[1] + " (rocprofv3 PMC, bytes per launch)",
                "compulsory_bytes": comp, "achieved_compulsory": compulsory, "frac_compulsory": compulsory / HBM_PEAK_GBS,
                "compulsory_formula": "per chain: n (57 + 8 w) read + sum_k (64 live_k + n) written + fused read-out 24 live_last + "
                                      "176 B per workgroup; from this run's survivor counts",
                "counted_over_compulsory": None if tr is None else tr[0] / comp,
                "shared_input_note": None if not (program is not None and n_chains > 1) else
                "all chains of this scene read the SAME source bundle: the launch is chain-interleaved (grid (chains, tiles)) and, "
                "while 57 B x rays <= 256 MiB, loads the source with the default cache policy, so it comes from HBM about once "
                "instead of once per chain -- counted bytes may lie below compulsory_bytes, which charges every chain its own read",
                "achieved_algorithmic": algo, "frac_algorithmic": algo / HBM_PEAK_GBS,
                "algorithmic_bytes_per_intersection": ALGO_BYTES_PER_INTERSECTION,
                "algorithmic_bytes_per_read_out_ray": ALGO_BYTES_READOUT if fuse else None,
                "algorithmic_bytes_per_launch": algo_bytes,
                # the same bytes over the DRIVER-TIMED step (ms_per_step: fold, gaps between launches and the clock ramp of
                # the first steps included) -- the fraction the contract's own clock supports
                "frac_step": (tr[0] if tr else comp) * launches / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
                "frac_step_note": "bytes per launch x launches per step / ms_per_step / peak: the timed region itself; `frac` "
                                  "divides by kernel_ms, which is measured AFTER the timed region (warmer clocks, no gaps)",
                "kernel_ms": kernel_ms, "launches_per_step": launches, "intersections_per_launch": inter_per_launch,
                "kernel_ms_note": f"POST-REGION: mean of {EVENT_STEPS} event-bracketed launches (trace kernel + its 9-us fold) "
                                  "issued right after the timed region(s) -- not inside them, because every timing event is a "
                                  "barrier packet that would slow the timed steps; the clocks are warmer there than in the first "
                                  "timed steps, so `frac` reads a few percent above `frac_step`; *_sustained: the same after the "
                                  "sustained-load region",
                "timed_region_launches": list(state.get("timed_launches", (0, 0))),
                "timed_region_launches_note": "[first, last) of the full-size fused-kernel launches of this process, in issue "
                                              "order, that lie inside the timed region: cuts a rocprofv3 kernel trace of the "
                                              "same command to it (tools/summarize_profile.py)",
                "kernel_ms_sustained": kernel_ms_sus,
                "frac_sustained": None if kernel_ms_sus is None else (tr[0] if tr else comp) / (kernel_ms_sus * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "frac_of_achievable_6300": (counted if counted is not None else compulsory) / 6300.0,
                "source_hash": __import__("tools.source_hash", fromlist=["source_hash"]).source_hash(),
            }
            if cfg == "relay4" and fuse and n == 10_000_000 and args.mirrors == 4:
                # the kernel's OTHER roof: SQ counters of this very workload and BUILD (tools/prof_sq.sh + summarize_sq.py)
                import glob
                for sq in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_relay4_sq.json")), reverse=True):
                    jsq = json.load(open(sq))
                    if jsq.get("source_hash") == res["roofline"]["source_hash"]:
                        jsq.pop("counters", None)
                        res["roofline"]["second_bound"] = dict(jsq, source=os.path.relpath(sq, ROOT))
                        break
            if readout_ms is None:
                res["roofline_readout"] = {
                    "fused": True, "kernel": res["roofline"]["kernel"],
                    "note": "the read-out rides on the tracing launch (art_trace_chain_readout / scene read-outs): its "
                            "24 B/ray of outputs and the per-workgroup partial statistics are part of that kernel's "
                            "traffic and time; `--readout separate` launches k_detector_readout instead"}
            else:
                tro, _ = profiled_traffic(pkey, r"k_detector_readout", n)
                algo_ro = ALGO_BYTES_READOUT * n / (readout_ms * 1e-3) / 1e9
                counted_ro = None if tro is None else tro[0] / (readout_ms * 1e-3) / 1e9
                res["roofline_readout"] = {
                    "fused": False, "bound": "hbm", "kernel": "k_detector_readout (+ k_readout_final)", "achieved": counted_ro,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None if counted_ro is None else counted_ro / HBM_PEAK_GBS,
                    "traffic": None if tro is None else tro[0],
                    "traffic_source": None if tro is None else tro[1] + " (rocprofv3 PMC, bytes per launch)",
                    "achieved_algorithmic": algo_ro, "frac_algorithmic": algo_ro / HBM_PEAK_GBS,
                    "algorithmic_bytes_per_ray": ALGO_BYTES_READOUT, "kernel_ms": readout_ms, "launches_per_step": n_chains}
            res["trace_only_intersections_per_s"] = inter_per_step_rank / (kernel_ms * launches * 1e-3)
        if args.cpu_sample > 0 and on_gpu:
            # N = 1: the CPU baseline (the oracle timed on a bounded sample) and the parity of that sample.  N > 1: the
            # baseline is an N = 1 figure, but `parity` stays on the line -- rank 0 traces a smaller oracle sample on its
            # own device after the timed regions (no collective involved; the other ranks are done)
            n_cpu = args.cpu_sample if world == 1 else min(args.cpu_sample, 200_000)
            v, inter, secs, oracle_result = cpu_baseline(element_lists[-1], src_kind, det_dist, n_cpu, ignore_defects)
            res["parity"] = parity_against(oracle_result, element_lists[-1], be, mode, ignore_defects)
        if world == 1 and args.cpu_sample > 0 and on_gpu:
            res["cpu_baseline"] = {"value": v, "unit": "intersections/s", "cores": 1, "kind": "port",
                                   "sample": f"oracle/art_oracle.py (NumPy, batched LAPACK eigvals; single thread) on "
                                             f"{args.cpu_sample} rays x {n_elems} elements of one chain + detector = {inter} "
                                             f"intersections in {secs:.1f} s; host has {os.cpu_count()} cores"}
            try:
                v2, inter2, secs2, thr = cpu_twin_allcores(element_lists[-1], src_kind, min(args.cpu_sample, 4_000_000),
                                                           ignore_defects)
                res["cpu_twin_allcores"] = {"value": v2, "unit": "intersections/s", "cores": thr,
                                            "note": f"oracle/twin: the kernels' per-ray code built by g++ -O2 -fopenmp, "
                                                    f"{inter2} intersections in {secs2:.2f} s (best of 3); for scale only"}
            except Exception as e:    # noqa: BLE001 -- an optional extra must never cost the result line
                log(f"[bench] cpu_twin_allcores skipped: {e!r}")
        if use_dist:
            log(f"[bench] N = {world}: `value` ({res['value']:.4g} intersections/s) is the step with the per-step exchange of "
                f"statistics + sample (ONE all-gather) -- the number the >= 6x scaling target is judged on; "
                f"`value_full_gather` ({res['value_full_gather']:.4g}) ships every surviving ray's record to rank 0 in every "
                f"step and is bounded by one xGMI link per peer (gather_floor_ms "
                f"{res['config']['gather_floor_ms']:.3f} ms per step vs {res['ms_per_step']:.3f} ms traced)")
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(res), flush=True)
        os.dup2(2, 1)
    if use_dist:
        dist.destroy_process_group()
    return 0


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="relay4", choices=CONFIGS, help="BASELINE.json configuration (default: the headline)")
    ap.add_argument("--rays", type=int, default=0, help="rays per GPU (0 = the configuration's own size)")
    ap.add_argument("--mirrors", type=int, default=4, help="relay4 only: number of toroidal mirrors")
    ap.add_argument("--mode", default=None, choices=[None, "chain", "element"])
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="replay the step from a HIP graph (auto = on); off: eager launches")
    ap.add_argument("--shard", default="blocks", choices=["blocks", "strided"],
                    help="N > 1: contiguous index ranges per rank (default) or rank r traces rays r, r + N, ...")
    ap.add_argument("--readout", default="auto", choices=["auto", "fused", "separate", "lite"],
                    help="fused: the detector read-out rides on the tracing launch; separate: its own kernel afterwards; "
                         "auto (default) = fused; lite: fused, but only 8 of the 22 statistics are reduced (what a plain "
                         "get_Delays / get_PointList2DCentre caller consumes) -- a measurement option, never the default")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="rays of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--placement-tries", type=int, default=1,
                    help="opt-in: let the step's program time its launch into N candidate output allocations and keep the "
                         "fastest (graph.SceneProgram; default 1 = take the first allocation)")
    ap.add_argument("--time-limit", type=float, default=float(os.environ.get("ART_BENCH_TIME_LIMIT", "1500")),
                    help="N > 1 launcher: wall-clock limit in seconds after which all workers are killed (exit code 4)")
    ap.add_argument("--pg-timeout", type=float, default=float(os.environ.get("ART_PG_TIMEOUT", "300")),
                    help="N > 1 worker: timeout in seconds of the process group's collectives")
    args = ap.parse_args(argv)
    if args.cpu_sample < 0:
        args.cpu_sample = {"relay4": 2_000_000, "C2": 1_000_000, "C3": 1_000_000, "C4": 400_000, "C5": 150_000}[args.config]
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_workers(args.gpus, argv, args.time_limit)      # nothing above or in there touches the GPU
    return worker(args)


if __name__ == "__main__":
    sys.exit(main())
