"""Time art_make_source (1e7 rays, 20 launches) and art_transform_bundle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.cuda.set_device(0)
from attosecondraytracing_amd import _lib, ModuleGeometry as mgeo
from attosecondraytracing_amd.bundle import RayBundle
be = _lib.get_backend()
n = 10_000_000
b = RayBundle.allocate(n, backend=be)
rot = mgeo.rotation_matrix(np.array([0.0, 0.0, 1.0]), np.array([1.0, 0.0, 0.0]))
for kind, size in ((0, 0.02), (1, 10.0)):
    be.make_source(kind, size, rot, np.zeros(3), 0, n, n, b.view())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        be.make_source(kind, size, rot, np.zeros(3), 0, n, n, b.view())
    e1.record(); torch.cuda.synchronize()
    print(f"make_source kind {kind}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per 1e7 rays ({650e6 / (e0.elapsed_time(e1) / 20 * 1e-3) / 1e12:.2f} TB/s)")
t = b.transformed(rot, np.array([1.0, 2.0, 3.0]))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    t = b.transformed(rot, np.array([1.0, 2.0, 3.0]))
e1.record(); torch.cuda.synchronize()
print(f"transform_bundle: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per 1e7 rays")
