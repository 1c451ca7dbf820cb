"""CPU (-m "not gpu"): the product's host shell + the kernels' per-ray device functions (compiled by g++ as the
CPU twin) against the reference's golden vectors.  Exercises descriptor packing, frame maps, the closed-form /
convex-Newton solvers and the Zernike recurrences exactly as the GPU runs them."""
import numpy as np
import pytest

from conftest import chain_golden_names, load_golden
import parity_common as pc


@pytest.fixture(scope="module")
def twin():
    from twin_backend import TwinBackend
    from attosecondraytracing_amd import _lib
    old = _lib._BACKEND
    _lib._BACKEND = TwinBackend()
    yield _lib._BACKEND
    _lib._BACKEND = old


@pytest.mark.parametrize("mode", ["element", "chain"])
@pytest.mark.parametrize("name", chain_golden_names())
def test_twin_chain_matches_reference(twin, name, mode):
    import ART.ModuleProcessing as mp
    scene, a = load_golden(name)
    els = pc.build_elements(scene, a)
    src = pc.source_bundle(a, scene)
    out = mp.RayTracingCalculation(src, els, IgnoreDefects=scene.get("IgnoreDefects", True), mode=mode)
    pc.check_outputs(out, a, scene)


@pytest.mark.parametrize("name", [n for n in chain_golden_names() if not n.startswith("frame_")])
def test_twin_detector_matches_reference(twin, name):
    import ART.ModuleProcessing as mp
    import ART.ModuleDetector as mdet
    import ART.ModuleAnalysisAndPlots as mplots
    scene, a = load_golden(name)
    if "detector" not in scene:
        pytest.skip("no detector in fixture")
    els = pc.build_elements(scene, a)
    src = pc.source_bundle(a, scene)
    last = mp.RayTracingCalculation(src, els, IgnoreDefects=scene.get("IgnoreDefects", True))[-1]
    d = scene["detector"]
    scale = max(1.0, np.abs(a["det_points3d"]).max())
    if name != "autofocus_c3":
        D = mdet.Detector(np.array(els[-1].position, float))
        D.autoplace(last, d["distance"])
        assert np.abs(D.centre - d["centre"]).max() <= 1e-10 * scale
        assert np.abs(D.normal - d["normal"]).max() <= 1e-10
        assert abs(D.get_distance() - d["distance"]) <= 1e-10 * max(1.0, d["distance"])
    D = mdet.Detector(np.array(d["refpoint"]), np.array(d["centre"]), np.array(d["normal"]))
    assert np.abs(D.get_PointList3D(last) - a["det_points3d"]).max() <= 1e-10 * scale
    assert np.abs(D.get_PointList2D(last) - a["det_points2d"]).max() <= 1e-10 * scale
    assert np.abs(D.get_PointList2DCentre(last) - a["det_points2dcentre"]).max() <= 1e-10 * scale
    mean_t_fs = np.mean(D.get_OpticalPaths(last)) / mdet.LightSpeed * 1e15
    assert np.abs(D.get_Delays(last) - a["det_delays"]).max() <= 1e-10 * mean_t_fs
    spot, dur = mplots.GetResultSummary(D, last, False)
    assert abs(spot - scene["SpotSizeSD"]) <= 1e-9 * scale
    assert abs(dur - scene["DurationSD"]) <= 1e-10 * mean_t_fs
    assert abs(mplots.getETransmission(src, last) - scene["ETransmission"]) <= 1e-9


def test_twin_edge_cases(twin):
    import edge_cases
    edge_cases.run_edge_cases()


def test_twin_zernike_high_order_vs_oracle(twin):
    """Zernike defects at order 16 (unrolled Horner tables) and at orders 20, 30, 48 (the recurrences per ray), any number of
    defects per mirror: against the oracle's recurrences and the long-double truth."""
    import zernike_cases
    from conftest import report
    for name, w in zernike_cases.run_high_order().items():
        report(f"[zernike {name}, local error vs long-double truth] " + "  ".join(f"{k} {v:.1e}" for k, v in w.items()))


def test_twin_batched_loop_lists_match_reference(twin):
    """Many chains in one launch (scene table): the reference's loop-list chains C2 (chains 0/5/10 of 11) and C3
    (chains 0/4/9 of 10) from ONE batched call each, against their fixtures and the single-chain launches."""
    import scene_cases
    scene_cases.run_batched_goldens(("c2_fxf_chain00", "c2_fxf_chain05", "c2_fxf_chain10"))
    scene_cases.run_batched_goldens(("c3_twisted_chain00", "c3_twisted_chain04", "c3_twisted_chain09"))


def test_twin_batched_variants(twin):
    import scene_cases
    scene_cases.run_batched_variants()


def test_twin_scene_program(twin):
    import scene_cases
    scene_cases.run_program_updates()
    scene_cases.run_program_history_off()
    scene_cases.run_chain_list_cache()


def test_twin_fused_readout(twin):
    import scene_cases
    scene_cases.run_fused_readout()


def test_twin_loop_list_prefix_sharing(twin):
    import scene_cases
    scene_cases.run_prefix_sharing()
