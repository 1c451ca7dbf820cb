"""CPU: the reference's shipped CONFIG_*.py scripts, read from /root/reference/examples (never copied), run
UNMODIFIED on this package through `ARTmain` (CPU twin of the kernels as backend).  Skipped where the reference
tree is absent (GPU box).  Numbers are compared with the golden fixtures / survey anchors of the same scenes."""
import importlib.util
import os
import sys

import numpy as np
import pytest

from conftest import load_golden

EXAMPLES = "/root/reference/examples"
pytestmark = pytest.mark.skipif(not os.path.isdir(EXAMPLES), reason="reference tree not present")


@pytest.fixture()
def twin(monkeypatch, tmp_path):
    from twin_backend import TwinBackend
    from attosecondraytracing_amd import _lib
    monkeypatch.setattr(_lib, "_BACKEND", TwinBackend())
    monkeypatch.chdir(tmp_path)            # save_results writes archives into the cwd
    # the reference's complete_defaults mutates module-level dicts: restore them after the test
    from attosecondraytracing_amd import DefaultOptions as D
    saved = {k: dict(getattr(D, k)) for k in ("DefaultAnalysisOptions", "DefaultSourceProperties", "DefaultDetectorOptions")}
    yield
    for k, v in saved.items():
        getattr(D, k).clear()
        getattr(D, k).update(v)


def _load(name):
    path = os.path.join(EXAMPLES, name)
    spec = importlib.util.spec_from_file_location(name[:-3], path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _run(mod):
    import ARTmain
    chains, src, det, ana = ARTmain.load_config(mod)
    ana = dict(ana)
    ana["save_results"] = False
    return ARTmain.main(chains, src, det, ana)


def test_config_singleparabola(twin):
    kept = _run(_load("CONFIG_singleparabola.py"))
    scene, a = load_golden("c1_singleparabola")
    chain = kept["OpticalChain"][0]
    out = chain.get_output_rays()[-1]
    assert len(chain.source_rays) == 999 and len(out) == scene["n_out"][-1] == 957
    assert np.array_equal(out.numbers(), a["out0_number"])
    assert np.abs(out.points() - a["out0_point"]).max() <= 1e-10 * 200
    assert abs(kept["SpotSizeSD"][0] - scene["SpotSizeSD"]) <= 1e-9
    assert abs(kept["DurationSD"][0] - scene["DurationSD"]) <= 1e-7      # 14.355 fs (survey anchor)
    assert abs(kept["DurationSD"][0] - 14.355) < 1e-3
    assert abs(kept["ETransmission"][0] - scene["ETransmission"]) <= 1e-9


def test_config_2toroidals_fxf(twin):
    """This script passes a 7th positional `render` argument to OEPlacement (TypeError in the reference itself)."""
    kept = _run(_load("CONFIG_2toroidals_f-x-f.py"))
    assert len(kept["OpticalChain"]) == 11
    for i, name in ((0, "c2_fxf_chain00"), (5, "c2_fxf_chain05"), (10, "c2_fxf_chain10")):
        scene, a = load_golden(name)
        out = kept["OpticalChain"][i].get_output_rays()
        assert [len(o) for o in out] == scene["n_out"] == [490, 490, 490]
        assert np.abs(out[-1].points() - a["out2_point"]).max() <= 1e-10 * 2000
        assert abs(kept["ETransmission"][i] - scene["ETransmission"]) <= 1e-9
        if i == 5:   # detector at 500 mm = the focus for d = 500: summary comparable with the fixture's detector
            pass


def test_config_2toroidals_twisted(twin):
    kept = _run(_load("CONFIG_2toroidals_twisted.py"))
    assert len(kept["OpticalChain"]) == 10
    for i, name in ((0, "c3_twisted_chain00"), (4, "c3_twisted_chain04"), (9, "c3_twisted_chain09")):
        scene, a = load_golden(name)
        out = kept["OpticalChain"][i].get_output_rays()
        assert [len(o) for o in out] == [673, 673, 673]
        for k in range(3):
            assert np.array_equal(out[k].numbers(), a[f"out{k}_number"])
            assert np.abs(out[k].points() - a[f"out{k}_point"]).max() <= 1e-10 * 2000
        assert abs(kept["ETransmission"][i] - 85.58) < 0.01
    # autofocus ran (AutoDetectorDistance=True): finite optimum near the nominal 600 mm
    for d, s, t in zip(kept["Detector"], kept["SpotSizeSD"], kept["DurationSD"]):
        assert 500 < d.get_distance() < 700 and np.isfinite(s) and np.isfinite(t)


@pytest.mark.parametrize("name", ["CONFIG_toroidal2f-2f.py", "CONFIG_toroidal2f-2f_byhand.py",
                                  "CONFIG_CollimatingTelescope.py"])
def test_other_configs_run(twin, name):
    kept = _run(_load(name))
    assert len(kept["OpticalChain"]) >= 1
    for ch, et in zip(kept["OpticalChain"], kept["ETransmission"]):
        assert len(ch.get_output_rays()[-1]) > 0
        assert 0 < et <= 100.0 + 1e-9


def test_config_deformed(twin):
    """CONFIG_deformed.py: a Fourrier height map with smallest = 0.01 mm on a 40 x 40 mm support = 8000 x 8000
    samples (0.5 GB as fp64, synthesised with a 4001 x 8000 complex FFT).  Runs unmodified; slow-ish on CPU."""
    import numpy as np
    np.random.seed(1)
    mod = _load("CONFIG_deformed.py")
    D = mod.DeformedMirror.DeformationList[0]
    assert D.deformation.shape == (8000, 8000) and abs(D.rms - 0.1) < 1e-12
    kept = _run(mod)
    out = kept["OpticalChain"][0].get_output_rays()[-1]
    assert len(kept["OpticalChain"][0].source_rays) == 999 and len(out) == 202      # same count as the Zernike fixtures
    assert np.isfinite(kept["SpotSizeSD"][0]) and kept["SpotSizeSD"][0] > 0
