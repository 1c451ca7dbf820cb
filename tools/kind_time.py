"""Per-kind timing of the single-element trace kernel (1e7 rays, HIP events around the launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tools import sweep
torch.cuda.set_device(0)
from attosecondraytracing_amd import _lib
import ART.ModuleMirror as mmirror, ART.ModuleSupport as msupp, ART.ModuleProcessing as mp, ART.ModuleMask as mmask
be = _lib.get_backend()
n = 10_000_000


def kernel_ms(src, els, mode, reps=10, **kw):
    o = mp.RayTracingCalculation(src, els, mode=mode, **kw); del o
    torch.cuda.synchronize()
    be.trace_events = []
    for _ in range(reps):
        o = mp.RayTracingCalculation(src, els, mode=mode, **kw); del o
    torch.cuda.synchronize()
    ev, be.trace_events = be.trace_events, None
    return sum(a.elapsed_time(b) for a, b in ev) / reps


S = msupp.SupportRound(60)
R, r = mmirror.ReturnOptimalToroidalRadii(600, 45)
optics = [("plane", mmirror.MirrorPlane(S), 45), ("sphere", mmirror.MirrorSpherical(1200, S), 45),
          ("parabola 0deg", mmirror.MirrorParabolic(300, 0, S), 0), ("parabola 90deg", mmirror.MirrorParabolic(300, 90, S), 0),
          ("torus", mmirror.MirrorToroidal(R, r, S), 45), ("ellipsoid", mmirror.MirrorEllipsoidal(S, f_object=600, f_image=600, OffAxisAngle=90), 0),
          ("cylinder", mmirror.MirrorCylindrical(1200, S), 45), ("mask", mmask.Mask(msupp.SupportRoundHole(60, 10, 0, 0)), 0)]
for srcname in ("point", "plane"):
    SP = {"Divergence": 0.02 if srcname == "point" else 0, "SourceSize": 0 if srcname == "point" else 20, "Wavelength": 50e-6,
          "DeltaFT": 0.5, "NumberRays": 1000}
    src = sweep.point_source(n, 0.02, be) if srcname == "point" else sweep.plane_source(n, 10.0, be)
    for name, M, inc in optics:
        els = mp.OEPlacement(SP, [M], [600], [inc], Description=name).optical_elements
        out = mp.RayTracingCalculation(src, els)
        frac = len(out[0]) / n
        del out
        for mode in ("element", "chain"):
            ms = kernel_ms(src, els, mode)
            print(f"{srcname:6s} {name:15s} {mode:8s} {ms:.3f} ms  survivors {frac:.2f}", flush=True)
