#!/usr/bin/env python3
"""Per-step durations of the driver's protocol (W warm-up steps after an idle device, then K timed steps): where inside
the timed region is the time?  Replays the relay4 step from a HIP graph like bench.py and records a HIP event after every
step (the events cost a few microseconds per step; the shape of the curve is what matters).
    python tools/ramp_probe.py [--warmup 5] [--steps 20] [--idle 0.5]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--idle", type=float, default=0.5, help="seconds of idle device before the warm-up steps")
    ap.add_argument("--repeat", type=int, default=3)
    args = ap.parse_args()
    import torch
    import bench
    import __graft_entry__
    __graft_entry__.ensure_built()
    from attosecondraytracing_amd import _lib
    from attosecondraytracing_amd.graph import SceneProgram
    import ART.ModuleProcessing as mp
    import ART.ModuleDetector as mdet
    be = _lib.get_backend()
    chain, _ = bench.build_scene(4)
    els = chain.optical_elements
    n = 10_000_000
    src = bench.device_source(n, 0, n, be)
    out = mp.RayTracingCalculation(src, els)
    det = mdet.Detector(np.asarray(els[-1].position, dtype=float))
    det.autoplace(out[-1], 600.0)
    del out
    prog = SceneProgram([src], [els], detectors=[det], post=lambda outs: [det.readout(outs[0][-1], sync=False)])
    for rep in range(args.repeat):
        torch.cuda.synchronize()
        time.sleep(args.idle)
        for _ in range(args.warmup):
            prog.run()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        t0 = time.perf_counter()
        ev[0].record()
        for k in range(args.steps):
            prog.run()
            ev[k + 1].record()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        per = [ev[k].elapsed_time(ev[k + 1]) for k in range(args.steps)]
        print(f"rep {rep}: {dt / args.steps * 1e3:.4f} ms/step; per step: " + " ".join(f"{v:.3f}" for v in per), flush=True)


if __name__ == "__main__":
    main()
