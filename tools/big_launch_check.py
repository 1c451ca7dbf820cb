"""One-off check at a REAL size beyond one launch's limit (2^28 rays): a bundle of 2^28 + 100000 rays through a
toroid + detector read-out, sampled rays (around the chunk boundary and at both ends) compared with the CPU oracle,
global statistics compared with a float64 torch reduction.  Needs ~60 GB of HBM."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch
import bench
from oracle import art_oracle as orc

torch.cuda.set_device(0)
from attosecondraytracing_amd import _lib
import ART.ModuleProcessing as mp
import ART.ModuleDetector as mdet
be = _lib.get_backend()
n = (1 << 28) + 100_000
chain, Rr = bench.build_scene(1)
src = bench.device_source(n, 0, n, be)
out = mp.RayTracingCalculation(src, chain.optical_elements)[-1]
det = mdet.Detector(np.asarray(chain.optical_elements[-1].position, dtype=float))
det.autoplace(out, 600.0)
ro = det.readout(out, sync=True)
s = ro["stats"]
print("rays", n, "alive", int(s[0]))
assert int(s[0]) == n == int(out.alive.sum(dtype=torch.int64).item())
# sampled slots: both ends and around the 2^28 boundary
slots = np.r_[0:50, (1 << 28) - 50:(1 << 28) + 50, n - 50:n]
st = torch.as_tensor(slots, device=be.device)
P = out.data[0:3].index_select(1, st).cpu().numpy().T
V = out.data[3:6].index_select(1, st).cpu().numpy().T
path = out.data[6].index_select(0, st).cpu().numpy()
X = ro["X"].index_select(0, st).cpu().numpy()
# oracle on the same source rays
sp = src.data[0:3].index_select(1, st).cpu().numpy().T
sv = src.data[3:6].index_select(1, st).cpu().numpy().T
R, r = Rr
oe = chain.optical_elements[0]
els = [orc.Element(orc.Optic("torus", orc.Support("rect", [200, 30]), {"R": R, "r": r}, [], "Toroidal Mirror"),
                   np.asarray(oe.position, float), oe.normal, oe.majoraxis)]
ref = orc.ray_tracing_calculation(orc.make_bundle(sp, sv, np.arange(len(slots)), None, None), els)[0]
assert len(ref) == len(slots)
print("max |dP|", np.abs(P - ref.point).max(), "max |dV|", np.abs(V - ref.vector).max(), "max |dpath|", np.abs(path - ref.path.sum(axis=1)).max())
assert np.abs(P - ref.point).max() <= 1e-10 * 1000 and np.abs(V - ref.vector).max() <= 1e-10
assert np.abs(path - ref.path.sum(axis=1)).max() <= 1e-10 * 1000
# statistics over all rays vs torch
mx = float(ro["X"].sum(dtype=torch.float64).item())
print("sum X fused", s[6], "torch", mx)
assert abs(s[6] - mx) <= 1e-9 * max(1.0, abs(mx)) + 1e-6
print("BIG_LAUNCH_OK")
