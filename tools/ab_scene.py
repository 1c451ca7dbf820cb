#!/usr/bin/env python3
"""In-process A/B of the grid shape and cache policy of a scene launch whose chains share their input (the library reads
ART_SCENE_ORDER / ART_SCENE_KEEP at every launch): tile-major grid, chain-interleaved 2-D grid, XCD-grouped 1-D grid (the C
workgroups of a tile on ONE XCD), the shared input loaded non-temporally or through the caches -- alternating round by round through
graph.SceneProgram._launch() (eager, fused read-outs) on the same resident data, every launch bracketed by HIP events.

    python tools/ab_scene.py C2|C3 RAYS full|last        (full = every per-element bundle written, last = lazy history)
How profiles/r04_experiments.md batch 8 was taken."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from attosecondraytracing_amd import _lib
from attosecondraytracing_amd.graph import SceneProgram
import ART.ModuleProcessing as mp
import ART.ModuleDetector as mdet

be = _lib.get_backend()
cfg, n, hist = sys.argv[1], int(float(sys.argv[2])), sys.argv[3] == "full"
lists, kind, dist = getattr(bench, "scene_" + cfg.lower())()
src = bench.device_source(n, 0, n, be, kind)
dets = []
for els in lists:
    out = mp.RayTracingCalculation(src, els, history=False)
    d = mdet.Detector(np.asarray(els[-1].position, dtype=float))
    d.autoplace(out[-1], dist)
    dets.append(d)
    del out
prog = SceneProgram([src] * len(lists), lists, capture=False, detectors=dets, history=hist)
variants = [("tile", "0"), ("chain", "0"), ("chain", "1"), ("xcd", "0"), ("xcd", "1")]
times = {v: [] for v in variants}
for rnd in range(9):                       # round 0 warms every variant up
    for v in variants:
        os.environ["ART_SCENE_ORDER"], os.environ["ART_SCENE_KEEP"] = v
        be.trace_events = []
        for _ in range(6):
            prog._launch()
        torch.cuda.synchronize()
        ev, be.trace_events = be.trace_events, None
        if rnd:
            times[v].append(float(np.mean([a.elapsed_time(b) for a, b in ev])))
base = np.median(times[variants[0]])
print(f"# {cfg} {n} rays x {len(lists)} chains, history {'full' if hist else 'last bundle only'}, SceneProgram launches, in-process A/B")
for v in variants:
    t = np.array(times[v])
    print(f"grid {v[0]:5s} cached input loads {v[1]}: median {np.median(t):.4f} ms  min {t.min():.4f}  max {t.max():.4f}  ratio {np.median(t) / base:.3f}")
