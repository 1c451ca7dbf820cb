"""Where does the HOST spend its time in a bench step?  cProfile over N eager steps of one configuration
(python tools/host_profile.py C4 [fused|separate] [steps])."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench

torch.cuda.set_device(0)
from attosecondraytracing_amd import _lib
import ART.ModuleProcessing as mp
import ART.ModuleDetector as mdet

cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
fuse = (sys.argv[2] if len(sys.argv) > 2 else "fused") == "fused"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
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


def step():
    o = mp.RayTracingCalculation(src, els, detector=det if fuse else None)
    r = det.readout(o[-1], sync=False)
    return o, r


for _ in range(5):
    o, r = step()
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(steps):
    o, r = step()
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{cfg} {'fused' if fuse else 'separate'}: host enqueue {1e3 * (t1 - t0) / steps:.3f} ms/step, wall {1e3 * (t2 - t0) / steps:.3f} ms/step")
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
