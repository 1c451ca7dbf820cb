#!/usr/bin/env python3
"""PCIe-inclusive rate of the hot path: host NumPy arrays in, host arrays of the detector read-out
back, everything in between on the GPU.  Reported for information; bench.py's `value` is the HBM-resident rate."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    torch.cuda.set_device(0)
    from attosecondraytracing_amd import _lib
    from attosecondraytracing_amd.bundle import RayBundle
    import ART.ModuleProcessing as mp
    import ART.ModuleDetector as mdet
    be = _lib.get_backend()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    chain, _ = bench.build_scene(4)
    els = chain.optical_elements
    src = bench.device_source(n, 0, n, be)
    host = src.data.cpu().numpy()
    pts, vec = np.ascontiguousarray(host[0:3].T), np.ascontiguousarray(host[3:6].T)
    out = mp.RayTracingCalculation(src, els)
    det = mdet.Detector(np.asarray(els[-1].position, dtype=float))
    det.autoplace(out[-1], 600.0)
    del out
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        b = RayBundle.from_arrays(pts, vec, None, None, 50e-6, backend=be)           # host -> device (57 B/ray)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        o = mp.RayTracingCalculation(b, els)
        r = det.readout(o[-1], sync=False)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        X, Y, opl = r["X"].cpu().numpy(), r["Y"].cpu().numpy(), r["opl"].cpu().numpy()   # device -> host (24 B/ray)
        alive = o[-1].alive.cpu().numpy()
        t3 = time.perf_counter()
        inter = 4 * n
        print(f"n={n}: host prep+H2D {1e3*(t1-t0):.1f} ms, trace+readout {1e3*(t2-t1):.2f} ms, D2H {1e3*(t3-t2):.1f} ms "
              f"-> PCIe-inclusive {inter/(t3-t0):.3e} intersections/s (resident: {inter/(t2-t1):.3e})", flush=True)


if __name__ == "__main__":
    main()
