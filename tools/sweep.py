#!/usr/bin/env python3
"""Throughput sweep on one MI355X: rays N in {1e5..1e8} x mirrors M in {2,4,8} (relay of toroids) plus the BASELINE
scenes C2/C3 (mask + 2 toroids), C1/C5 (parabola, Zernike-deformed parabola) and the 8-element mixed chain.
Trace-only timing (HIP events around the launches), full per-element history, fp64.  Writes a markdown table."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def time_trace(src, els, mode, reps, **kw):
    import ART.ModuleProcessing as mp
    be = src.backend
    out = mp.RayTracingCalculation(src, els, mode=mode, **kw)
    entering = [src.n_slots] + [int(o.alive.sum().item()) for o in out[:-1]]
    surv = [int(o.alive.sum().item()) for o in out]
    del out
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        o = mp.RayTracingCalculation(src, els, mode=mode, **kw)
        del o
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return ms, sum(entering), surv


def scene_c3(n):
    import ART.ModuleMirror as mmirror, ART.ModuleMask as mmask, ART.ModuleSupport as msupp, ART.ModuleProcessing as mp
    SP = {"Divergence": 50e-3 / 2, "SourceSize": 0, "Wavelength": 50e-6, "DeltaFT": 0.5, "NumberRays": 1000}
    Mask = mmask.Mask(msupp.SupportRoundHole(30, 41e-3 / 2 * 500, 0, 0))
    R, r = mmirror.ReturnOptimalToroidalRadii(600, 80)
    Tor = mmirror.MirrorToroidal(R, r, msupp.SupportRectangle(200, 30))
    return mp.OEPlacement(SP, [Mask, Tor, Tor], [500, 100, 600], [0, 80, -80], [0, 0, 30.0], "C3").optical_elements, 0.025


def scene_c2(n):
    import ART.ModuleMirror as mmirror, ART.ModuleMask as mmask, ART.ModuleSupport as msupp, ART.ModuleProcessing as mp
    SP = {"Divergence": 50e-3 / 2, "SourceSize": 0, "Wavelength": 50e-6, "DeltaFT": 0.5, "NumberRays": 1000}
    Mask = mmask.Mask(msupp.SupportRoundHole(20, 14e-3 * 500, 0, 0))
    R, r = mmirror.ReturnOptimalToroidalRadii(500, 80)
    Tor = mmirror.MirrorToroidal(R, r, msupp.SupportRectangle(150, 32))
    return mp.OEPlacement(SP, [Mask, Tor, Tor], [400, 100, 500], [0, 80, -80], [0, 0, 0], "C2").optical_elements, 0.025


def scene_mixed8(n):
    import ART.ModuleMirror as mmirror, ART.ModuleSupport as msupp, ART.ModuleProcessing as mp
    SP = {"Divergence": 0.03, "SourceSize": 0, "Wavelength": 50e-6, "DeltaFT": 0.5, "NumberRays": 1000}
    oap = mmirror.MirrorParabolic(200, 60, msupp.SupportRound(20))
    plane = mmirror.MirrorPlane(msupp.SupportRound(30))
    R, r = mmirror.ReturnOptimalToroidalRadii(400, 78)
    tor = mmirror.MirrorToroidal(R, r, msupp.SupportRectangle(180, 30))
    oap2 = mmirror.MirrorParabolic(150, 45, msupp.SupportRound(25))
    ch = mp.OEPlacement(SP, [oap, plane, tor, tor, plane, plane, oap2, plane], [200, 150, 250, 800, 650, 120, 140, 60],
                        [0, 45, 78, -78, 30, -30, 0, 20], [0, 0, 0, 0, 90, 0, 0, 45], "mixed8")
    return ch.optical_elements, 0.03


def scene_c5(n, perturbed):
    import ART.ModuleMirror as mmirror, ART.ModuleSupport as msupp, ART.ModuleProcessing as mp, ART.ModuleDefects as mdef
    S = msupp.SupportRectangle(40, 40)
    M = mmirror.MirrorParabolic(25.4, 0, S)
    Z = mdef.Zernike(S, {(2, 1): 1e-4, (3, 1): 5e-5, (4, 2): 2e-5, (3, 3): -3e-5, (5, 2): 1e-5, (6, 3): -4e-6, (2, 0): 2.5e-5})
    SP = {"Divergence": 0, "SourceSize": 40, "Wavelength": 800e-6, "DeltaFT": 0, "NumberRays": 1000}
    ch = mp.OEPlacement(SP, [mmirror.DeformedMirror(M, [Z])], [15], [0], Description="C5")
    return ch.optical_elements


def point_source(n, div, be):
    from attosecondraytracing_amd.bundle import RayBundle
    from attosecondraytracing_amd import ModuleGeometry as mgeo
    b = RayBundle.allocate(n, backend=be)
    rot = mgeo.rotation_matrix(np.array([0.0, 0.0, 1.0]), np.array([1.0, 0.0, 0.0]))
    be.make_source(0, div, rot, np.zeros(3), 0, n, n, b.view())
    return b


def plane_source(n, radius, be):
    from attosecondraytracing_amd.bundle import RayBundle
    from attosecondraytracing_amd import ModuleGeometry as mgeo
    b = RayBundle.allocate(n - 1, backend=be)
    rot = mgeo.rotation_matrix(np.array([0.0, 0.0, 1.0]), np.array([1.0, 0.0, 0.0]))
    be.make_source(1, radius, rot, np.zeros(3), 0, n - 1, n, b.view())
    return b


def main():
    torch.cuda.set_device(0)
    from attosecondraytracing_amd import _lib
    be = _lib.get_backend()
    rows = []

    def add(name, src, els, mode, reps, **kw):
        ms, inter, surv = time_trace(src, els, mode, reps, **kw)
        rows.append((name, src.n_slots, len(els), mode, ms, inter / ms * 1e3, 128 * inter / ms * 1e3 / 1e9 / 8000, surv[-1] / src.n_slots))
        print(rows[-1], flush=True)

    for M in (2, 4, 8):
        chain, _ = bench.build_scene(M)
        for n in (100_000, 1_000_000, 10_000_000, 100_000_000):
            if n * M * 65 > 120e9:
                continue
            src = point_source(n, 0.02, be)
            reps = max(3, min(50, int(2e8 / (n * M))))
            for mode in ("chain", "element"):
                add(f"relay{M} (toroids)", src, chain.optical_elements, mode, reps)
            del src
            torch.cuda.empty_cache()
    for name, fn in (("C2 f-x-f: mask + 2 toroids", scene_c2), ("C3 twisted: mask + 2 toroids", scene_c3),
                     ("C4 mixed8: OAP, plane, 2 toroids, 2 planes, OAP, plane", scene_mixed8)):
        els, div = fn(0)
        for n in (1_000_000, 10_000_000):
            src = point_source(n, div, be)
            for mode in ("chain", "element"):
                add(name, src, els, mode, 10)
            del src
    els = scene_c5(0, False)
    for n in (10_000_000,):
        src = plane_source(n, 20.0, be)
        add("C5 parabola + Zernike(order 6), IgnoreDefects=True", src, els, "chain", 10, IgnoreDefects=True)
        add("C5 parabola + Zernike(order 6), IgnoreDefects=False", src, els, "chain", 10, IgnoreDefects=False)
    out = ["| scene | rays | elements | mode | ms / trace | intersections/s | frac of 8 TB/s (128 B alg.) | survivors |",
           "|---|---:|---:|---|---:|---:|---:|---:|"]
    for r in rows:
        out.append(f"| {r[0]} | {r[1]:.0e} | {r[2]} | {r[3]} | {r[4]:.3f} | {r[5]:.3e} | {r[6]:.2f} | {r[7]:.3f} |")
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "sweep.md")
    open(path, "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
