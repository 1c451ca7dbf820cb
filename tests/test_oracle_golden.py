"""The CPU oracle (oracle/art_oracle.py) against golden vectors produced by the reference itself
(tests/golden/generate_goldens.py).  Tolerances are written here: survivor indices exact; positions,
directions, paths 1e-11 relative to the scene scale (the oracle runs the same solver as the reference)."""
import numpy as np
import pytest

from conftest import chain_golden_names, load_golden
from oracle import art_oracle as orc


def source_from_arrays(a, scene):
    B = orc.make_bundle(a["src_point"], a["src_vector"], a["src_number"], a["src_intensity"], scene.get("wavelength"))
    return B


@pytest.mark.parametrize("name", chain_golden_names())
@pytest.mark.parametrize("fast", [True, False])
def test_oracle_chain_matches_reference(name, fast):
    scene, a = load_golden(name)
    if not fast and scene["n_source"] > 600 and len(scene["elements"]) > 3:
        pytest.skip("per-ray quaternion loop only on the small cases")
    src = source_from_arrays(a, scene)
    els = orc.elements_from_scene(scene, a)
    for e, d in zip(els, scene["elements"]):
        assert np.allclose(e.optic.centre(), d["centre"], rtol=0, atol=1e-12 * max(1.0, np.abs(d["centre"]).max()))
    out = orc.ray_tracing_calculation(src, els, IgnoreDefects=scene.get("IgnoreDefects", True), fast=fast)
    assert [len(o) for o in out] == scene["n_out"]
    scale = max(1.0, np.abs(a["src_point"]).max(), *(np.abs(np.array(e["position"])).max() for e in scene["elements"]))
    for k, o in enumerate(out):
        assert np.array_equal(o.number, a[f"out{k}_number"]), f"survivor indices differ at element {k}"
        if len(o) == 0:
            continue
        assert np.abs(o.point - a[f"out{k}_point"]).max() <= 1e-11 * scale
        assert np.abs(o.vector - a[f"out{k}_vector"]).max() <= 1e-11
        assert np.abs(o.incidence - a[f"out{k}_incidence"]).max() <= 1e-11
        ref_path = a[f"out{k}_path"]
        assert o.path.shape == ref_path.shape
        assert np.abs(o.path - ref_path).max() <= 1e-11 * scale
        assert np.array_equal(np.isnan(o.intensity), np.isnan(a[f"out{k}_intensity"]))
        m = ~np.isnan(o.intensity)
        assert np.array_equal(o.intensity[m], a[f"out{k}_intensity"][m])


@pytest.mark.parametrize("name", [n for n in chain_golden_names() if not n.startswith("frame_")])
def test_oracle_detector_matches_reference(name):
    scene, a = load_golden(name)
    if "detector" not in scene:
        pytest.skip("no detector in fixture")
    src = source_from_arrays(a, scene)
    els = orc.elements_from_scene(scene, a)
    out = orc.ray_tracing_calculation(src, els, IgnoreDefects=scene.get("IgnoreDefects", True))
    last = out[-1]
    d = scene["detector"]
    if name != "autofocus_c3":
        D = orc.detector_autoplace(last, d["distance"])
        scale = max(1.0, np.abs(np.array(d["centre"])).max())
        assert np.abs(D.centre - d["centre"]).max() <= 1e-10 * scale
        assert np.abs(D.normal - d["normal"]).max() <= 1e-11
        assert np.abs(D.refpoint - d["refpoint"]).max() <= 1e-10 * scale
        assert abs(orc.detector_distance(D) - d["distance"]) <= 1e-10 * max(1, d["distance"])
    # use the reference's detector pose for the read-out comparison
    D = orc.Detector(np.array(d["centre"]), np.array(d["normal"]), np.array(d["refpoint"]))
    scale = max(1.0, np.abs(a["det_points3d"]).max())
    assert np.abs(orc.detector_points3d(D, last) - a["det_points3d"]).max() <= 1e-11 * scale
    assert np.abs(orc.detector_points2d(D, last) - a["det_points2d"]).max() <= 1e-10 * scale
    assert np.abs(orc.detector_points2dcentre(D, last) - a["det_points2dcentre"]).max() <= 1e-10 * scale
    paths = orc.optical_paths(D, last)
    mean_t_fs = np.mean(paths) / orc.LightSpeed * 1e15
    # delays are path differences: tolerance normalised by the mean travel time
    assert np.abs(orc.detector_delays(D, last) - a["det_delays"]).max() <= 1e-12 * mean_t_fs
    assert abs(orc.standard_deviation(orc.detector_points2dcentre(D, last)) - scene["SpotSizeSD"]) <= 1e-9 * scale
    assert abs(orc.standard_deviation(orc.detector_delays(D, last)) - scene["DurationSD"]) <= 1e-11 * mean_t_fs
    et = 100 * np.sum(last.intensity) / np.sum(src.intensity)
    assert abs(et - scene["ETransmission"]) <= 1e-10


@pytest.mark.parametrize("name", ["analysis_c2", "analysis_c3"])
def test_oracle_analysis_matches_reference(name):
    """The oracle's restatement of the analysis behind the trace (e_transmission, detector_autoplace, result_summary,
    numerical_aperture, find_optimal_distance: position by position, like ART/ModuleProcessing.py:317-460) against what
    the reference computed for EVERY chain of the shipped loop lists (generate_analysis_goldens.py)."""
    scene, a = load_golden(name)
    src = source_from_arrays(a, scene)
    for i, c in enumerate(scene["chains"]):
        els = orc.elements_from_scene(c, a)
        last = orc.ray_tracing_calculation(src, els)[-1]
        assert np.array_equal(last.number, a[f"c{i}_last_number"])
        assert abs(orc.e_transmission(src, last) - c["ETransmission"]) <= 1e-10
        D = orc.detector_autoplace(last, scene["detector_distance"])
        d = c["detector"]
        assert np.abs(D.centre - d["centre"]).max() <= 1e-9 and np.abs(D.normal - d["normal"]).max() <= 1e-11
        D = orc.Detector(np.array(d["centre"]), np.array(d["normal"]), np.array(d["refpoint"]))
        spot, dur = orc.result_summary(D, last)
        assert abs(spot - c["SpotSizeSD"]) <= 1e-9 * c["SpotSizeSD"] and abs(dur - c["DurationSD"]) <= 1e-8 * c["DurationSD"]
        assert abs(orc.numerical_aperture(last) - c["NA"]) <= 1e-11
        keys = ("intensity_1", "intensity_0", "duration_1") if i % 3 == 0 else ("intensity_1",)      # (bounds the run time)
        for key in keys:
            optfor, weighted = key.rsplit("_", 1)
            Do, s, t = orc.find_optimal_distance(D, last, optfor, None, 3, bool(int(weighted)))
            dist, rs, rt = c["autofocus"][key]
            assert abs(orc.detector_distance(Do) - dist) <= 1e-9 * dist, (i, key, orc.detector_distance(Do), dist)
            if not np.isnan(rs):
                assert abs(s - rs) <= 1e-8 * rs, (i, key)
            assert abs(t - rt) <= 1e-7 * rt, (i, key)


def test_zernike_tierA():
    """Fixture generated from reference modules that import with no stand-in at all."""
    _, a = load_golden("zernike_tierA")
    x, y = a["x"], a["y"]
    nm = [tuple(int(v) for v in r) for r in a["nm"]]
    Z, GX, GY = orc.zernike_tables(x, y, max(n for n, _ in nm))
    for i, k in enumerate(nm):
        assert np.array_equal(Z[k] * np.ones_like(x), a["val"][i])
        assert np.array_equal(GX[k] * np.ones_like(x), a["gx"][i])
        assert np.array_equal(GY[k] * np.ones_like(x), a["gy"][i])
    D = orc.ZernikeDefect({(int(r[0]), int(r[1])): float(r[2]) for r in a["defect_coeffs"]}, float(a["defect_R"]))
    assert np.abs(orc.zernike_normal(D, a["defect_points"]) - a["defect_normal"]).max() <= 1e-18
    assert np.abs(orc.zernike_offset(D, a["defect_points"]) - a["defect_offset"]).max() <= 1e-18


def test_geometry_units():
    _, a = load_golden("geometry_units")
    r = orc.np_roots_batch(a["quad_in"])
    t = r.real
    ok = (np.abs(r.imag) < 1e-15) & ~np.isnan(r.real)
    for i in range(len(t)):
        mine = [v for v, o in zip(t[i], ok[i]) if o]
        ref = [v for v in a["quad_out"][i] if not np.isnan(v)]
        assert mine == ref
    assert np.abs(orc.angle_between(a["angle_U"], a["angle_V"]) - a["angle_out"]).max() <= 1e-15
    for p, a1, a2, o in zip(a["rot_P"], a["rot_A1"], a["rot_A2"], a["rot_out"]):
        assert np.abs(orc.rotation_point(p, a1, a2) - o).max() <= 1e-14
    for ax, an, p, o in zip(a["raa_axis"], a["raa_angle"], a["rot_P"], a["raa_out"]):
        assert np.abs(orc.rotation_around_axis(ax, an, p) - o).max() <= 1e-14
    assert np.array_equal(orc.spiral_vogel(7, 2.5), a["vogel_7_2p5"])
    assert np.array_equal(orc.spiral_vogel(1000, 1.0), a["vogel_1000_1"])
    ps = orc.apply_gaussian_intensity(orc.point_source([1.0, 2.0, 3.0], [0.3, -0.2, 0.9], 0.05, 50))
    assert np.abs(ps.point - a["src_pointsource_point"]).max() <= 1e-14
    assert np.abs(ps.vector - a["src_pointsource_vector"]).max() <= 1e-14
    assert np.abs(ps.intensity - a["src_pointsource_intensity"]).max() <= 1e-12
    pw = orc.apply_gaussian_intensity(orc.plane_wave_disk([1.0, 2.0, 3.0], [0.0, 1.0, 0.2], 12.0, 50))
    assert len(pw) == 49
    assert np.abs(pw.point - a["src_planewave_point"]).max() <= 1e-13
    assert np.abs(pw.vector - a["src_planewave_vector"]).max() <= 1e-14
    assert np.abs(pw.intensity - a["src_planewave_intensity"]).max() <= 1e-12
    assert np.abs(orc.normal_add(a["nadd_1"], a["nadd_2"]) - a["nadd_out"]).max() <= 1e-14
    pts, w, dl = a["stat_pts"], a["stat_w"], a["stat_delays"]
    mine = [orc.standard_deviation(pts), orc.weighted_standard_deviation(pts, w), orc.standard_deviation(dl),
            orc.weighted_standard_deviation(dl, w)]
    assert np.abs(np.array(mine) - a["stat_out"]).max() <= 1e-14
