"""Micro-benchmark of the fused detector read-out (art_detector_readout) in its variants: per-ray outputs stored or
not, weights present or not, 3-D points or not.  Prints microseconds per launch and the algorithmic bytes moved."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench


def main(n=10_000_000, reps=30):
    from attosecondraytracing_amd import _lib
    import ART.ModuleProcessing as mp
    import ART.ModuleDetector as mdet
    be = _lib.get_backend()
    chain, _ = bench.build_scene(4)
    src = bench.device_source(n, 0, n, be)
    rays = mp.RayTracingCalculation(src, chain.optical_elements, history=False)[-1]
    det = mdet.Detector(np.asarray(chain.optical_elements[-1].position, dtype=float))
    det.autoplace(rays, 600.0)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    for store, p3, wts in [(True, False, True), (False, False, True), (True, True, True), (True, False, False),
                           (False, False, False)]:
        w = rays.intensity
        if not wts:
            rays.intensity = None
        for _ in range(3):
            det.readout(rays, points3d=p3, sync=False, store=store)
        a, b = ev(), ev()
        a.record()
        for _ in range(reps):
            det.readout(rays, points3d=p3, sync=False, store=store)
        b.record(); torch.cuda.synchronize()
        rays.intensity = w
        us = a.elapsed_time(b) / reps * 1e3
        byts = n * (7 * 8 + 1 + (8 if wts else 0) + (24 if store else 0) + (24 if p3 and store else 0))
        print(f"store={store} p3={p3} weights={wts}: {us:7.1f} us/launch  {byts/1e6:6.0f} MB  {byts/us/1e6:5.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
