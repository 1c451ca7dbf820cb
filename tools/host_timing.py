"""Which host-side call of a step blocks?  Wraps the step's building blocks with perf_counter timers (no cProfile:
the stall this hunts for disappears when the host is slowed down).  python tools/host_timing.py C4 fused 60"""
import os
import sys
import time
import collections

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench

torch.cuda.set_device(0)
from attosecondraytracing_amd import _lib
from attosecondraytracing_amd.bundle import RayBundle
import ART.ModuleProcessing as mp
import ART.ModuleDetector as mdet

cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
fuse = (sys.argv[2] if len(sys.argv) > 2 else "fused") == "fused"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
be = _lib.get_backend()
lists, kind, dist_ = {"relay4": lambda: ([bench.build_scene(4)[0].optical_elements], ("point", 0.02), 600.0),
                      "C4": bench.scene_c4, "C5": bench.scene_c5}[cfg]()
n = {"relay4": 10_000_000, "C4": 12_500_000, "C5": 10_000_000}[cfg]
src = bench.device_source(n, 0, n, be, kind)
els = lists[0]
out = mp.RayTracingCalculation(src, els)
det = mdet.Detector(np.asarray(els[-1].position, dtype=float))
det.autoplace(out[-1], dist_)
del out

T = collections.defaultdict(float)
worst = collections.defaultdict(float)


def wrap(obj, name, label=None):
    f = getattr(obj, name)
    label = label or name

    def g(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        dt = time.perf_counter() - t0
        T[label] += dt
        worst[label] = max(worst[label], dt)
        return r
    setattr(obj, name, g)


wrap(RayBundle, "allocate_many")
wrap(be, "new_chain_readout")
wrap(be, "trace_chain")
wrap(be, "detector_readout")
wrap(be, "empty")
for name in list(be.fn):
    if name in ("art_trace_chain", "art_trace_chain_readout", "art_detector_readout"):
        f = be.fn[name]

        def mk(f, name):
            def g(*a):
                t0 = time.perf_counter()
                r = f(*a)
                dt = time.perf_counter() - t0
                T["C:" + name] += dt
                worst["C:" + name] = max(worst["C:" + name], dt)
                return r
            return g
        be.fn[name] = mk(f, name)


def step():
    o = mp.RayTracingCalculation(src, els, detector=det if fuse else None)
    r = det.readout(o[-1], sync=False)
    return o, r


for _ in range(5):
    o, r = step()
torch.cuda.synchronize()
T.clear(); worst.clear()
t0 = time.perf_counter()
per = []
for _ in range(steps):
    t1 = time.perf_counter()
    o, r = step()
    per.append(time.perf_counter() - t1)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f"{cfg} {'fused' if fuse else 'separate'}: host enqueue {1e3 * t_enq / steps:.3f} ms/step, wall {1e3 * wall / steps:.3f} ms/step")
print("per-step host ms:", " ".join(f"{1e3 * p:.2f}" for p in per))
for k in sorted(T, key=lambda k: -T[k]):
    print(f"  {k:32s} total {1e3 * T[k]:8.2f} ms   worst call {1e3 * worst[k]:7.3f} ms")
