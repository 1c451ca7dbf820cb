"""Time the trace kernels of a diagnostic build (ART_HIP_LIB) on the relay4 workload; results may be wrong by design."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tools import sweep
torch.cuda.set_device(0)
from attosecondraytracing_amd import _lib
be = _lib.get_backend()
chain, _ = bench.build_scene(4)
src = sweep.point_source(10_000_000, 0.02, be)
import ART.ModuleProcessing as mp
for mode in ("chain", "element"):
    o = mp.RayTracingCalculation(src, chain.optical_elements, mode=mode); del o
    torch.cuda.synchronize()
    be.trace_events = []
    for _ in range(10):
        o = mp.RayTracingCalculation(src, chain.optical_elements, mode=mode); del o
    torch.cuda.synchronize()
    ev, be.trace_events = be.trace_events, None
    ms = sum(a.elapsed_time(b) for a, b in ev) / 10
    print(f"{os.environ.get('ART_HIP_LIB','default').split('/')[-1]:24s} {mode:8s} {ms:.3f} ms per 4e7 intersections")
