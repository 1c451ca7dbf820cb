"""CPU: the C-ABI library loads and exports every symbol include/art_hip.h declares; the ctypes mirror matches
the C struct sizes; the product refuses to trace without a GPU (no silent CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import torch  # noqa: F401  (before the library: one HIP runtime per process, see __graft_entry__.build)
    import __graft_entry__ as g
    if g._stale(g.HIP_LIB, g.HIP_DEPS):
        g.build()
    return C.CDLL(g.HIP_LIB)


def test_every_declared_symbol_is_exported(lib):
    from attosecondraytracing_amd import _abi
    hdr = open(os.path.join(ROOT, "include", "art_hip.h")).read()
    declared = set(re.findall(r"\b(art_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_abi.PROTOTYPES), declared ^ set(_abi.PROTOTYPES)
    for name in declared:
        assert hasattr(lib, name), name
    _abi.bind(lib)
    assert lib.art_abi_version() == _abi.ART_ABI_VERSION


def test_struct_layout_matches_header():
    """Compile a tiny C program against the header and compare sizeof/offsetof with the ctypes mirror."""
    import subprocess
    import tempfile
    from attosecondraytracing_amd import _abi
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "art_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(ArtElementDesc), offsetof(ArtElementDesc, fwd),
         offsetof(ArtElementDesc, sp), offsetof(ArtElementDesc, zern), sizeof(ArtBundleView),
         sizeof(ArtDetectorDesc), (size_t)ART_ZERN_STRIDE, sizeof(ArtChainReadout), offsetof(ArtChainReadout, w),
         offsetof(ArtChainReadout, X), offsetof(ArtChainReadout, out24));
  printf("%zu %zu\n", offsetof(ArtChainReadout, sums), offsetof(ArtAnalysisJob, sums));
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %d %d %d\n", offsetof(ArtChainReadout, lite), sizeof(ArtAnalysisJob),
         offsetof(ArtAnalysisJob, w), offsetof(ArtAnalysisJob, distance), offsetof(ArtAnalysisJob, mode),
         offsetof(ArtAnalysisJob, centre), offsetof(ArtAnalysisJob, normal), offsetof(ArtAnalysisJob, refpoint),
         ART_ANALYSIS_DOUBLES, ART_GUIDES_MAX, ART_MAX_DEFECTS);
  return 0;
}'''
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(td, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", exe, c])
        vals = [int(v) for v in subprocess.check_output([exe]).split()]
    E = _abi.ArtElementDesc
    R = _abi.ArtChainReadout
    J = _abi.ArtAnalysisJob
    assert vals == [C.sizeof(E), E.fwd.offset, E.sp.offset, E.zern.offset, C.sizeof(_abi.ArtBundleView),
                    C.sizeof(_abi.ArtDetectorDesc), _abi.ART_ZERN_STRIDE, C.sizeof(R), R.w.offset, R.X.offset,
                    R.out24.offset, R.sums.offset, J.sums.offset,
                    R.lite.offset, C.sizeof(J), J.w.offset, J.distance.offset, J.mode.offset, J.centre.offset,
                    J.normal.offset, J.refpoint.offset, _abi.ART_ANALYSIS_DOUBLES, _abi.ART_GUIDES_MAX, _abi.ART_MAX_DEFECTS]


def test_no_cpu_fallback():
    """Without a GPU the product path must fail loudly."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from attosecondraytracing_amd import _lib
    old = _lib._BACKEND
    _lib._BACKEND = None
    try:
        with pytest.raises(RuntimeError, match="no CPU fallback|No HIP device"):
            _lib.get_backend()
        import numpy as np
        import ART.ModuleOpticalRay as mray
        import ART.ModuleProcessing as mp
        with pytest.raises(RuntimeError):
            mp.RayTracingCalculation([mray.Ray(np.zeros(3), np.array([1.0, 0, 0]))], [])
    finally:
        _lib._BACKEND = old


def test_loader_takes_no_library_from_the_environment(monkeypatch):
    """The process-wide backend loads the in-tree product library and nothing else: no environment variable can put
    another .so (a CPU stand-in, say) in its place.  Diagnostic builds are loaded explicitly: HipBackend(path=...)."""
    import importlib
    from attosecondraytracing_amd import _lib
    monkeypatch.setenv("ART_HIP_LIB", "/tmp/some_other_library.so")
    old = _lib._BACKEND
    try:
        lib2 = importlib.reload(_lib)
        assert lib2.LIB_PATH == os.path.join(ROOT, "attosecondraytracing_amd", "libart_hip.so")
        src = open(os.path.join(ROOT, "attosecondraytracing_amd", "_lib.py")).read()
        assert "os.environ" not in src and "getenv" not in src
    finally:
        _lib._BACKEND = old


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "attosecondraytracing_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
                assert "_twin" not in txt and "art_cpu_" not in txt, f
