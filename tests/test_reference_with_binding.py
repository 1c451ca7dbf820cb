"""CPU: the REFERENCE ITSELF (read from /root/reference, its own OpticalChain / OEPlacement / Detector classes) with
its `RayTracingCalculation` replaced by the INTEGRATION.md binding stub calling our C ABI (CPU twin of the kernels
here).  Run in a subprocess because the reference's package is also called `ART`.  The patched reference must
reproduce what the unpatched reference computes."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")

SCRIPT = textwrap.dedent('''
    import sys, ctypes as C
    sys.dont_write_bytecode = True
    ROOT, REF = sys.argv[1], sys.argv[2]
    sys.path[:0] = [REF, ROOT + "/tests/golden/_standin", ROOT + "/tests", ROOT]     # `ART` = the reference
    import matplotlib; matplotlib.use("Agg")
    import numpy as np, torch
    import ART.ModuleProcessing as mp, ART.ModuleMirror as mmirror, ART.ModuleMask as mmask
    import ART.ModuleSupport as msupp, ART.ModuleDetector as mdet, ART.ModuleOpticalRay as mray
    assert mp.__file__.startswith(REF)
    from twin_backend import build_twin
    import binding_stub
    lib = C.CDLL(build_twin())
    SP = {"Divergence": 50e-3 / 2, "SourceSize": 0, "Wavelength": 50e-6, "DeltaFT": 0.5, "NumberRays": 400}
    Mask = mmask.Mask(msupp.SupportRoundHole(30, 41e-3 / 2 * 500, 0, 0))
    R, r = mmirror.ReturnOptimalToroidalRadii(600, 80)
    Tor = mmirror.MirrorToroidal(R, r, msupp.SupportRectangle(200, 30))
    oap = mmirror.MirrorParabolic(150, 45, msupp.SupportRound(40))
    original = mp.RayTracingCalculation
    def build():
        return mp.OEPlacement(SP, [Mask, Tor, Tor, oap], [500, 100, 600, 400], [0, 80, -80, 0], [0, 0, 30.0, 0], "stub")
    ref_chain = build()
    ref = ref_chain.get_output_rays()
    mp.RayTracingCalculation = binding_stub.make_binding(lib, "art_cpu_", torch.device("cpu"), mray.Ray)
    chain = build()                      # OEPlacement's alignment traces already go through the binding
    out = chain.get_output_rays()
    for a, b in zip(ref_chain.optical_elements, chain.optical_elements):
        assert np.abs(a.position - b.position).max() <= 1e-9 and np.abs(a.normal - b.normal).max() <= 1e-12
    assert [len(o) for o in out] == [len(o) for o in ref] and len(out[-1]) > 200
    for o, q in zip(out, ref):
        assert [x.number for x in o] == [x.number for x in q]
        assert max(np.abs(x.point - y.point).max() for x, y in zip(o, q)) <= 1e-10 * 2000
        assert max(np.abs(x.vector - y.vector).max() for x, y in zip(o, q)) <= 1e-10
        assert max(abs(sum(x.path) - sum(y.path)) for x, y in zip(o, q)) <= 1e-10 * 2000
        assert max(abs(x.incidence - y.incidence) for x, y in zip(o, q)) <= 1e-9
    det = mdet.Detector(chain.optical_elements[-1].position); det.autoplace(out[-1], 150)
    det0 = mdet.Detector(ref_chain.optical_elements[-1].position); det0.autoplace(ref[-1], 150)
    d1, d0 = np.array(det.get_Delays(out[-1])), np.array(det0.get_Delays(ref[-1]))
    assert np.abs(d1 - d0).max() <= 1e-10 * 1e15 * 2000 / 299792458000
    print("BINDING_OK", [len(o) for o in out])
''')


def test_reference_runs_on_our_abi_through_the_stub():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", SCRIPT, ROOT, REF], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "BINDING_OK" in r.stdout
