import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools import sweep
torch.cuda.set_device(0)
from attosecondraytracing_amd import _lib
be = _lib.get_backend()
els = sweep.scene_c5(0, False)
src = sweep.plane_source(10_000_000, 20.0, be)
for ign in (True, False):
    for mode in ("chain", "element"):
        ms, inter, surv = sweep.time_trace(src, els, mode, 10, IgnoreDefects=ign)
        print(f"C5 Zernike order 6, IgnoreDefects={ign}, {mode}: {ms:.3f} ms, {inter/ms*1e3:.3e} int/s")
