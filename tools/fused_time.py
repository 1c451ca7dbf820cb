"""Time the fused trace + read-out launch against trace and read-out in separate launches on relay4 (1e7 rays); with a
diagnostic build (ART_HIP_LIB=..., -DART_DIAG_RO_*) the fused results are wrong by design, only the time counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
torch.cuda.set_device(0)
from attosecondraytracing_amd import _lib
import ART.ModuleProcessing as mp
import ART.ModuleDetector as mdet
be = _lib.get_backend()
cfg = os.environ.get("ART_DIAG_CFG", "relay4")
if cfg == "relay4":
    els, kind, n = bench.build_scene(4)[0].optical_elements, ("point", 0.02), 10_000_000
else:
    lists, kind, _ = bench.scene_c4()
    els, n = lists[0], 12_500_000
src = bench.device_source(n, 0, n, be, kind)
det = mdet.Detector(np.asarray(els[-1].position, dtype=float), np.asarray(els[-1].position, dtype=float) + np.array([600.0, 0, 0]),
                    np.array([-1.0, 0.0, 0.0]))
import gc
gc.collect()
gc.freeze()       # keep Python's 40-ms full collections out of the 30-step timings
tag = os.environ.get("ART_DIAG_TAG", os.environ.get("ART_HIP_LIB", "default").split("/")[-1])
for name, fused in (("separate", False), ("fused", True), ("separate", False), ("fused", True)):
    def step():
        o = mp.RayTracingCalculation(src, els, detector=det if fused else None)
        r = det.readout(o[-1], sync=False)
        return o, r
    for _ in range(5):
        o, r = step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        o, r = step()
    e1.record()
    torch.cuda.synchronize()
    print(f"{tag:26s} {cfg} {name:9s} {e0.elapsed_time(e1) / 30:.4f} ms per step (trace + read-out)", flush=True)
