"""Time the trace kernels of a diagnostic build (ART_HIP_LIB) on the relay4 workload; with ART_DIAG_CHECK=1 the fused
chain launch is also compared with the per-element launches on the alive slots (variants that keep results intact)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tools import sweep
torch.cuda.set_device(0)
from attosecondraytracing_amd import _lib
be = _lib.get_backend()
chain, _ = bench.build_scene(4)
n = int(os.environ.get("ART_DIAG_RAYS", "10000000"))
src = sweep.point_source(n, 0.02, be)
import ART.ModuleProcessing as mp
tag = os.environ.get("ART_DIAG_TAG", os.environ.get("ART_HIP_LIB", "default").split("/")[-1])
if os.environ.get("ART_DIAG_CHECK") == "1":
    a = mp.RayTracingCalculation(src, chain.optical_elements, mode="chain")
    b = mp.RayTracingCalculation(src, chain.optical_elements, mode="element")
    for x, y in zip(a, b):
        assert torch.equal(x.alive, y.alive)
        m = x.alive.bool()
        assert int(m.sum()) > 0.9 * n
        assert float((x.data[:, m] - y.data[:, m]).abs().max()) <= 1e-9, float((x.data[:, m] - y.data[:, m]).abs().max())
    print(f"{tag:28s} chain == element on alive slots", flush=True)
    del a, b
for mode in ("chain", "element"):
    for rep in range(2):
        o = mp.RayTracingCalculation(src, chain.optical_elements, mode=mode); del o
    torch.cuda.synchronize()
    be.trace_events = []
    for _ in range(20):
        o = mp.RayTracingCalculation(src, chain.optical_elements, mode=mode); del o
    torch.cuda.synchronize()
    ev, be.trace_events = be.trace_events, None
    ms = sum(a.elapsed_time(b) for a, b in ev) / 20
    print(f"{tag:28s} {mode:8s} {ms:.4f} ms per {4*n:.0e} intersections", flush=True)
