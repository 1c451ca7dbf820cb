"""C5 as shipped (examples/CONFIG_deformed.py geometry): parabola with a `Fourrier` height map, 1e7 rays, timing of
the trace (offset look-ups = 4 random 8-byte taps per ray in the map) for several map sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tools import sweep
torch.cuda.set_device(0)
from attosecondraytracing_amd import _lib
import ART.ModuleMirror as mmirror, ART.ModuleSupport as msupp, ART.ModuleProcessing as mp, ART.ModuleDefects as mdef
be = _lib.get_backend()
n = 10_000_000
S = msupp.SupportRectangle(40, 40)
SP = {"Divergence": 0, "SourceSize": 40, "Wavelength": 800e-6, "DeltaFT": 0, "NumberRays": 1000}
src = sweep.plane_source(n, 20.0, be)
for smallest in (1.0, 0.1, 0.02, 0.01):
    np.random.seed(1)
    t0 = time.perf_counter()
    D = mdef.Fourrier(S, RMS=1e-4, smallest=smallest)
    t_map = time.perf_counter() - t0
    M = mmirror.DeformedMirror(mmirror.MirrorParabolic(25.4, 0, S), [D])
    els = mp.OEPlacement(SP, [M], [15], [0], Description="C5").optical_elements
    o = mp.RayTracingCalculation(src, els); del o
    torch.cuda.synchronize()
    be.trace_events = []
    for _ in range(10):
        o = mp.RayTracingCalculation(src, els); del o
    torch.cuda.synchronize()
    ev, be.trace_events = be.trace_events, None
    ms = sum(a.elapsed_time(b) for a, b in ev) / 10
    print(f"smallest {smallest:5.2f} mm: map {D.deformation.shape[1]} x {D.deformation.shape[0]} ({D.deformation.nbytes / 1e6:.0f} MB, "
          f"host synthesis {t_map:.2f} s): trace {ms:.3f} ms per 1e7 rays = {n / ms * 1e3:.3e} intersections/s", flush=True)
