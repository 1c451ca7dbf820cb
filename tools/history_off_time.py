"""relay4 trace with and without the per-element history (history=False keeps only the last bundle)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tools import sweep
torch.cuda.set_device(0)
from attosecondraytracing_amd import _lib
import ART.ModuleProcessing as mp
be = _lib.get_backend()
chain, _ = bench.build_scene(4)
src = sweep.point_source(10_000_000, 0.02, be)
for hist in (True, False):
    for mode in ("chain", "element"):
        o = mp.RayTracingCalculation(src, chain.optical_elements, mode=mode, history=hist); del o
        torch.cuda.synchronize()
        be.trace_events = []
        for _ in range(10):
            o = mp.RayTracingCalculation(src, chain.optical_elements, mode=mode, history=hist); del o
        torch.cuda.synchronize()
        ev, be.trace_events = be.trace_events, None
        ms = sum(a.elapsed_time(b) for a, b in ev) / 10
        print(f"history={hist!s:5s} {mode:8s} {ms:.3f} ms per 4e7 intersections  ({4e7 / ms * 1e3:.3e}/s)", flush=True)
