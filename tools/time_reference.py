#!/usr/bin/env python3
"""Time the REFERENCE ITSELF (read from /root/reference, never copied) on its own scenes in this container:
`OpticalChain.get_output_rays()` + detector read-out, single Python thread.  SURVEY.md 8(d) "CPU baseline (1)".
Only runs where the reference tree exists (the build container); writes a markdown table.

usage: python tools/time_reference.py [out.md] [rays]      (also writes out.json: what bench.py quotes as
cpu_baseline.reference_as_is -- a figure of the BUILD CONTAINER, the reference cannot travel to the GPU box)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path[:0] = [REF, ROOT + "/tests/golden/_standin"]     # `ART` = the reference; quaternion stand-in
import matplotlib  # noqa: E402
matplotlib.use("Agg")
import numpy as np  # noqa: E402
import ART.ModuleProcessing as mp  # noqa: E402
import ART.ModuleMirror as mmirror  # noqa: E402
import ART.ModuleMask as mmask  # noqa: E402
import ART.ModuleSupport as msupp  # noqa: E402
import ART.ModuleDetector as mdet  # noqa: E402
import ART.ModuleDefects as mdef  # noqa: E402

assert mp.__file__.startswith(REF)


def scenes(n):
    SPp = {"Divergence": 50e-3 / 2, "SourceSize": 0, "Wavelength": 50e-6, "DeltaFT": 0.5, "NumberRays": n}
    R, r = mmirror.ReturnOptimalToroidalRadii(600, 80)
    tor = mmirror.MirrorToroidal(R, r, msupp.SupportRectangle(200, 30))
    yield "C1 singleparabola: OAP(f=100, 90 deg), plane wave", mp.OEPlacement(
        {"Divergence": 0, "SourceSize": 50, "Wavelength": 800e-6, "DeltaFT": 0.5, "NumberRays": n},
        [mmirror.MirrorParabolic(100, 90, msupp.SupportRoundHole(30, 5, 10, 5))], [200], [0], [0], "C1"), 100
    R5, r5 = mmirror.ReturnOptimalToroidalRadii(500, 80)
    tor5 = mmirror.MirrorToroidal(R5, r5, msupp.SupportRectangle(150, 32))
    yield "C2 f-x-f: mask + 2 toroids (chain 5 of 11: d = 500 mm)", mp.OEPlacement(
        dict(SPp), [mmask.Mask(msupp.SupportRoundHole(20, 14e-3 * 500, 0, 0)), tor5, tor5], [400, 100, 500],
        [0, 80, -80], [0, 0, 0], "C2"), 500
    yield "C3 twisted: mask + 2 toroids", mp.OEPlacement(
        dict(SPp), [mmask.Mask(msupp.SupportRoundHole(30, 41e-3 / 2 * 500, 0, 0)), tor, tor], [500, 100, 600],
        [0, 80, -80], [0, 0, 30.0], "C3"), 600
    SP4 = dict(SPp, Divergence=0.02)
    yield "relay4: 4 toroids (the bench.py workload)", mp.OEPlacement(
        SP4, [tor] * 4, [600, 600, 1200, 600], [80, -80, 80, -80], [0, 0, 0, 0], "relay4"), 600
    oap = mmirror.MirrorParabolic(200, 60, msupp.SupportRound(20))
    plane = mmirror.MirrorPlane(msupp.SupportRound(30))
    R4, r4 = mmirror.ReturnOptimalToroidalRadii(400, 78)
    tor4 = mmirror.MirrorToroidal(R4, r4, msupp.SupportRectangle(180, 30))
    oap2 = mmirror.MirrorParabolic(150, 45, msupp.SupportRound(25))
    yield "C4 8-element mixed chain (OAP, plane, 2 toroids, 2 planes, OAP, plane)", mp.OEPlacement(
        dict(SPp, Divergence=0.03), [oap, plane, tor4, tor4, plane, plane, oap2, plane], [200, 150, 250, 800, 650, 120, 140, 60],
        [0, 45, 78, -78, 30, -30, 0, 20], [0, 0, 0, 0, 90, 0, 0, 45], "C4"), 100
    S = msupp.SupportRectangle(40, 40)
    Z = mdef.Zernike(S, {(2, 1): 1e-4, (3, 1): 5e-5, (4, 2): 2e-5, (3, 3): -3e-5, (5, 2): 1e-5, (6, 3): -4e-6, (2, 0): 2.5e-5})
    yield "C5 parabola + Zernike(order 6), IgnoreDefects=False", mp.OEPlacement(
        {"Divergence": 0, "SourceSize": 40, "Wavelength": 800e-6, "DeltaFT": 0, "NumberRays": n},
        [mmirror.DeformedMirror(mmirror.MirrorParabolic(25.4, 0, S), [Z])], [15], [0], Description="C5"), 25.4


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "reference_cpu.md")
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    rows = []
    for name, chain, ddist in scenes(n):
        kw = {"IgnoreDefects": False} if "Zernike" in name else {}
        t0 = time.perf_counter()
        out = chain.get_output_rays(**kw)
        t_trace = time.perf_counter() - t0
        inter = len(chain.source_rays) + sum(len(o) for o in out[:-1])
        D = mdet.Detector(chain.optical_elements[-1].position)
        D.autoplace(out[-1], ddist)
        t0 = time.perf_counter()
        D.get_PointList2DCentre(out[-1])
        D.get_Delays(out[-1])
        t_det = time.perf_counter() - t0
        rows.append((name, len(chain.source_rays), len(chain.optical_elements), inter, t_trace, inter / t_trace, len(out[-1]), t_det))
        print(rows[-1], flush=True)
    lines = ["# The reference as shipped, timed in the build container (tools/time_reference.py)", "",
             f"One Python thread of an {os.cpu_count()}-core x86 host, Python {sys.version.split()[0]}, NumPy {np.__version__}; "
             "`quaternion` stand-in of tests/golden/_standin (SURVEY 8c).  The loops are per-ray and independent, so the "
             "rate is independent of the ray count; 4e7 intersections (bench.py's step) extrapolate linearly.", "",
             "| scene | rays | elements | intersections | trace s | intersections/s | 4e7 intersections would take | detector read-out of the survivors |",
             "|---|---:|---:|---:|---:|---:|---:|---:|"]
    for r in rows:
        lines.append(f"| {r[0]} | {r[1]} | {r[2]} | {r[3]} | {r[4]:.2f} | {r[5]:.3g} | {4e7 / r[5] / 3600:.1f} h | {r[6]} rays in {r[7]:.2f} s |")
    open(out_path, "w").write("\n".join(lines) + "\n")
    keys = {"C1": "C1", "C2": "C2", "C3": "C3", "relay4": "relay4", "C4": "C4", "C5": "C5"}
    js = {"where": f"build container, one Python thread of {os.cpu_count()} cores, Python {sys.version.split()[0]}, NumPy {np.__version__}",
          "tool": "tools/time_reference.py", "configs": {}}
    for r in rows:
        k = next(v for kk, v in keys.items() if r[0].startswith(kk))
        js["configs"][k] = {"value": r[5], "unit": "intersections/s", "rays": r[1], "elements": r[2], "intersections": r[3],
                            "seconds": r[4], "scene": r[0]}
    open(os.path.splitext(out_path)[0] + ".json", "w").write(json.dumps(js, indent=1) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
