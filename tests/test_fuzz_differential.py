"""CPU (-m "not gpu"): random scenes, product host shell + CPU twin of the kernels against the pinned oracle
(tests/fuzz_common.py).  The GPU run of the same scenes is tests/test_gpu_parity.py::test_gpu_fuzz_differential."""
import pytest

import fuzz_common as fz


@pytest.fixture(scope="module")
def twin():
    from twin_backend import TwinBackend
    from attosecondraytracing_amd import _lib
    old = _lib._BACKEND
    _lib._BACKEND = TwinBackend()
    yield _lib._BACKEND
    _lib._BACKEND = old


@pytest.mark.parametrize("block", range(6))
def test_fuzz_twin_vs_oracle(twin, block):
    st = {}
    res = fz.run_differential(range(block * 20, block * 20 + 20), stats=st)
    assert res["scenes_with_hits"] >= 12, res   # the generator must actually exercise the optics
    # every element of every scene was checked against the long-double truth; product vs oracle beyond 1e-10 only where
    # the truth says the product is the closer one (seed 65 in block 3)
    assert st["scenes"] == 20 and max(st["local_worst"].values()) > 0.0 and st["adjudicated"] <= 4, st


def test_fuzz_regressions(twin):
    """Seeds that once failed: 20797 / 23917 = self-intersecting torus with the ray origin inside the inner "lemon"
    next to its tip, outside the sphere of radius r - R that was wrongly used to bound it."""
    fz.run_differential([20797, 23917])


def test_fuzz_detector_readout(twin):
    assert fz.run_detector_fuzz(range(60)) >= 80


def test_fuzz_sources(twin):
    fz.run_source_fuzz(range(40))
