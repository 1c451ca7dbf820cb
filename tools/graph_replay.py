"""Small bundles are launch-bound (a 1e5-ray relay4 step is ~40 us of GPU work behind ~120 us of host work).  The
library's launches are plain asynchronous kernel launches on the caller's stream with descriptors passed by value, so
a whole step -- trace + read-out -- can be captured once into a HIP graph (torch.cuda.graph) and replayed.
Prints the per-step time of eager launches and of graph replays and checks that both give identical results."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
torch.cuda.set_device(0)
from attosecondraytracing_amd import _lib
import ART.ModuleProcessing as mp
import ART.ModuleDetector as mdet
be = _lib.get_backend()
for n in (10_000, 100_000, 1_000_000):
    chain, _ = bench.build_scene(4)
    src = bench.device_source(n, 0, n, be)
    els = chain.optical_elements
    out = mp.RayTracingCalculation(src, els)
    det = mdet.Detector(np.asarray(els[-1].position, dtype=float))
    det.autoplace(out[-1], 600.0)

    def step():
        o = mp.RayTracingCalculation(src, els)
        r = det.readout(o[-1], sync=False)
        return o, r

    for _ in range(3):
        o, r = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        o, r = step()
    torch.cuda.synchronize()
    eager_us = (time.perf_counter() - t0) / 200 * 1e6
    ref_stats = r["stats_dev"].clone()
    ref_X = r["X"].clone()

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        go, gr = step()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(gr["stats_dev"], ref_stats) and torch.equal(gr["X"], ref_X)
    t0 = time.perf_counter()
    for _ in range(200):
        g.replay()
    torch.cuda.synchronize()
    graph_us = (time.perf_counter() - t0) / 200 * 1e6
    print(f"n={n:>8}: eager {eager_us:7.1f} us/step, graph replay {graph_us:7.1f} us/step  ({4 * n / graph_us * 1e6:.3e} intersections/s)", flush=True)
