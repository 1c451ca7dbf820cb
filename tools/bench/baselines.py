"""The CPU figures beside the GPU number: the oracle (the NumPy restatement of the reference's algorithm) timed on a bounded
sample, the parity of that sample traced on the GPU, the kernels' own per-ray code on all host cores, and the reference
AS IT IS (timed in the build container, where alone it exists: tools/time_reference.py -> profiles/rNN_reference_cpu.json).
The oracle is test infrastructure: it is called here as the checker and the baseline, never by the product."""
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def log(*a):
    print(*a, file=sys.stderr, flush=True)

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




def reference_as_is(cfg):
    """The reference ITSELF on this configuration's scene (its own per-ray Python loops, ART/ModuleProcessing.py:250-313),
    as timed in the BUILD CONTAINER by tools/time_reference.py: the reference cannot travel to the GPU box, so this is a
    quoted figure of another host, from the newest committed profiles/rNN_reference_cpu.json.  -> dict or None."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_reference_cpu.json")), reverse=True):
        try:
            j = json.load(open(f))
        except Exception:      # noqa: BLE001
            continue
        e = j.get("configs", {}).get(cfg)
        if e:
            return {"value": e["value"], "unit": "intersections/s", "cores": 1, "where": j.get("where"),
                    "rays": e["rays"], "intersections": e["intersections"], "seconds": e["seconds"], "scene": e.get("scene"),
                    "source": os.path.relpath(f, ROOT) + " (" + j.get("tool", "tools/time_reference.py") + ")",
                    "note": "the reference as shipped, one Python thread; measured where the reference tree exists, NOT on this box"}
    return None
