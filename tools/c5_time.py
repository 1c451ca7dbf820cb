"""C5 (deformed parabola) timing breakdown on one GPU: plain parabola vs Zernike orders, both trace modes, point and
plane-wave sources.  Kernel time from HIP events around the launches (be.trace_events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tools import sweep
torch.cuda.set_device(0)
from attosecondraytracing_amd import _lib
import ART.ModuleMirror as mmirror, ART.ModuleSupport as msupp, ART.ModuleProcessing as mp, ART.ModuleDefects as mdef
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


S = msupp.SupportRectangle(40, 40)
SP = {"Divergence": 0, "SourceSize": 40, "Wavelength": 800e-6, "DeltaFT": 0, "NumberRays": 1000}
src = sweep.plane_source(n, 20.0, be)
cases = [("plain parabola", None)]
for order in (2, 4, 6, 10, 16):
    coeffs = {(k, k // 2): 1e-5 for k in range(2, order + 1)}
    cases.append((f"Zernike order {order}", coeffs))
for name, coeffs in cases:
    M = mmirror.MirrorParabolic(25.4, 0, S)
    if coeffs:
        M = mmirror.DeformedMirror(M, [mdef.Zernike(S, coeffs)])
    els = mp.OEPlacement(SP, [M], [15], [0], Description="C5").optical_elements
    for ign in ((True,) if coeffs is None else (True, False)):
        for mode in ("chain", "element"):
            ms = kernel_ms(src, els, mode, IgnoreDefects=ign)
            print(f"{name:18s} IgnoreDefects={ign!s:5s} {mode:8s} {ms:.3f} ms  {n/ms*1e3:.3e} int/s  frac {128*n/ms*1e3/8e12:.2f}", flush=True)
