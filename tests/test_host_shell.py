"""CPU: the host-side API shell (scene construction, sources, detector placement, autofocus, misalignment helpers,
statistics) against golden vectors from the reference.  Tracing inside these calls runs on the CPU twin."""
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


PLACED = [n for n in chain_golden_names() if not n.startswith("frame_")]


@pytest.mark.parametrize("name", PLACED)
def test_oeplacement_matches_reference(twin, name):
    """OEPlacement (ART/ModuleProcessing.py:32-246): same poses, same source bundle, same Gaussian weights."""
    import ART.ModuleProcessing as mp
    scene, a = load_golden(name)
    pl = scene.get("placement") or scene.get("placement_before_roll")
    if pl is None:
        pytest.skip("poses were modified after placement in this fixture")
    optics = [pc.build_optic(e, a) for e in scene["elements"]]
    SP = dict(pl["SourceProperties"])
    SP["NumberRays"] = int(SP["NumberRays"])
    chain = mp.OEPlacement(SP, optics, list(pl["DistanceList"]), list(pl["IncidenceAngleList"]),
                           list(pl["IncidencePlaneAngleList"]), "t")
    if "placement" in scene:
        for oe, e in zip(chain.optical_elements, scene["elements"]):
            scale = max(1.0, np.abs(np.array(e["position"])).max())
            assert np.abs(np.asarray(oe.position, float) - e["position"]).max() <= 1e-11 * scale
            assert np.abs(oe.normal - e["normal"]).max() <= 1e-12
            assert np.abs(oe.majoraxis - e["majoraxis"]).max() <= 1e-12
    if not pl.get("jittered_source"):
        src = chain.source_rays
        assert len(src) == scene["n_source"]
        assert np.array_equal(src.numbers(), a["src_number"])
        assert np.abs(src.points() - a["src_point"]).max() <= 1e-12 * max(1.0, np.abs(a["src_point"]).max())
        assert np.abs(src.vectors() - a["src_vector"]).max() <= 1e-13
        assert np.abs(src.intensities() - a["src_intensity"]).max() <= 1e-11


def test_geometry_helpers(twin):
    import ART.ModuleGeometry as mgeo
    import ART.ModuleProcessing as mp
    import ART.ModuleSource as msource
    import ART.ModuleOpticalElement as moe
    import ART.ModuleMirror as mmirror
    import ART.ModuleSupport as msupp
    _, a = load_golden("geometry_units")
    for q, o in zip(a["quad_in"], a["quad_out"]):
        assert mgeo.SolverQuadratic(*q) == [v for v in o if not np.isnan(v)]
    for u, v, o in zip(a["angle_U"], a["angle_V"], a["angle_out"]):
        assert abs(mgeo.AngleBetweenTwoVectors(u, v) - o) <= 1e-15
    for p, a1, a2, o in zip(a["rot_P"], a["rot_A1"], a["rot_A2"], a["rot_out"]):
        assert np.abs(mgeo.RotationPoint(p, a1, a2) - o).max() <= 1e-13
    for ax, an, p, o in zip(a["raa_axis"], a["raa_angle"], a["rot_P"], a["raa_out"]):
        assert np.abs(mgeo.RotationAroundAxis(ax, an, p) - o).max() <= 1e-13
    assert np.array_equal(mgeo.SpiralVogel(7, 2.5), a["vogel_7_2p5"])
    assert np.abs(np.array([mgeo.normal_add(x, y) for x, y in zip(a["nadd_1"], a["nadd_2"])]) - a["nadd_out"]).max() <= 1e-14
    # sources (ART/ModuleSource.py) incl. the N-1 quirk of PlaneWaveDisk and ExtendedSource numbering
    for tag, b in (("pointsource", msource.PointSource(np.array([1.0, 2.0, 3.0]), np.array([0.3, -0.2, 0.9]), 0.05, 50)),
                   ("planewave", msource.PlaneWaveDisk(np.array([1.0, 2.0, 3.0]), np.array([0.0, 1.0, 0.2]), 12.0, 50)),
                   ("extended", msource.ExtendedSource(np.array([0.0, 0.0, 0.0]), np.array([1.0, 0.0, 0.0]), 0.1, 0.02, 9000))):
        b = msource.ApplyGaussianIntensityToRayList(b, 1 / np.e ** 2)
        assert len(b) == len(a[f"src_{tag}_number"])
        assert np.array_equal(b.numbers(), a[f"src_{tag}_number"])
        assert np.abs(b.points() - a[f"src_{tag}_point"]).max() <= 1e-12 * 12
        assert np.abs(b.vectors() - a[f"src_{tag}_vector"]).max() <= 1e-13
        assert np.abs(b.intensities() - a[f"src_{tag}_intensity"]).max() <= 1e-11
    # OpticalElement misalignment helpers (ART/ModuleOpticalElement.py:169-250)
    n0 = np.array([0.2, -0.4, 0.7])
    oe = moe.OpticalElement(mmirror.MirrorPlane(msupp.SupportRound(5)), np.array([1.0, 2.0, 3.0]), n0,
                            np.cross(n0, np.array([0.0, 0.0, 1.0])))
    for (op, val), ref in zip((("rotate_pitch_by", 1.5), ("rotate_roll_by", -0.7), ("rotate_yaw_by", 12.0),
                               ("shift_along_normal", 0.3), ("shift_along_major", -0.2), ("shift_along_cross", 0.9)),
                              a["oe_seq"]):
        getattr(oe, op)(val)
        assert np.abs(np.concatenate([oe.position, oe.normal, oe.majoraxis]) - ref).max() <= 1e-13
    pts, w, dl = a["stat_pts"], a["stat_w"], a["stat_delays"]
    mine = [mp.StandardDeviation(list(pts)), mp.WeightedStandardDeviation(list(pts), list(w)),
            mp.StandardDeviation(list(dl)), mp.WeightedStandardDeviation(list(dl), list(w))]
    assert np.abs(np.array(mine) - a["stat_out"]).max() <= 1e-14


def test_autofocus_matches_reference(twin):
    """FindOptimalDistance (ART/ModuleProcessing.py:317-460) on all rays: same optimum, spot size and duration."""
    import ART.ModuleProcessing as mp
    import ART.ModuleDetector as mdet
    scene, a = load_golden("autofocus_c3")
    els = pc.build_elements(scene, a)
    src = pc.source_bundle(a, scene)
    last = mp.RayTracingCalculation(src, els)[-1]
    d = scene["detector"]
    det = mdet.Detector(np.array(d["refpoint"]), np.array(d["centre"]), np.array(d["normal"]))
    assert abs(mp.ReturnNumericalAperture(last, 1) - scene["NA"]) <= 1e-12
    assert abs(mp.ReturnAiryRadius(50e-6, scene["NA"]) - scene["Airy"]) <= 1e-15
    for key, (dist, spot, dur) in scene["autofocus"].items():
        optfor, weighted = key.rsplit("_", 1)
        D, s, t = mp.FindOptimalDistance(det, last, optfor, None, 3, bool(int(weighted)), False)
        assert abs(D.get_distance() - dist) <= 1e-9 * dist, key
        if not np.isnan(spot):
            assert abs(s - spot) <= 1e-9 * max(spot, 1e-3), key
        assert abs(t - dur) <= 1e-7 * max(dur, 1e-3), key
    with pytest.raises(NameError):
        mp.FindOptimalDistance(det, last, "spotsize")          # reference quirk kept: only 'size' passes the check


def test_bundle_list_protocol(twin):
    """RayBundle behaves like the reference's list of Ray objects."""
    import ART.ModuleProcessing as mp
    import ART.ModuleOpticalRay as mray
    scene, a = load_golden("c3_twisted_chain04")
    els = pc.build_elements(scene, a)
    out = mp.RayTracingCalculation(pc.source_bundle(a, scene), els)
    b = out[-1]
    assert len(b) == 673
    r0, rl = b[0], b[-1]
    assert isinstance(r0, mray.Ray)
    assert r0.number == a["out2_number"][0] and rl.number == a["out2_number"][-1]
    assert np.allclose(r0.point, a["out2_point"][0], rtol=0, atol=1e-7)
    assert len(r0.path) == 4 and r0.path[0] == 0.0
    assert np.allclose(r0.path, a["out2_path"][0], rtol=0, atol=1e-7)
    assert abs(r0.incidence - a["out2_incidence"][0]) <= 1e-9
    assert abs(r0.intensity - a["out2_intensity"][0]) <= 1e-15
    assert r0.wavelength == scene["wavelength"]
    nums = [r.number for r in b]
    assert nums == list(a["out2_number"])
    assert [r.number for r in b[5:8]] == list(a["out2_number"][5:8])
    with pytest.raises(IndexError):
        b[673]
    sub = b.subset([0, 10, 20])
    assert len(sub) == 3 and [r.number for r in sub] == [nums[0], nums[10], nums[20]]
    # a list of Ray objects is accepted wherever a bundle is
    rays = [b[i] for i in range(5)]
    again = mp.RayTracingCalculation(rays, [])
    assert again == []
    with pytest.raises(NameError):
        class Weird:
            type = "Lens"
            support = els[0].type.support
        import ART.ModuleOpticalElement as moe
        mp.RayTracingCalculation(rays, [moe.OpticalElement(Weird(), np.zeros(3), np.array([0, 0, 1.0]), np.array([1.0, 0, 0]))])


def test_chain_cache_and_misalignment(twin):
    """get_output_rays recomputes only when the source or an element changed (ModuleOpticalChain.py:183-202)."""
    import ART.ModuleOpticalChain as moc
    scene, a = load_golden("c2_fxf_chain05")
    chain = moc.OpticalChain(pc.source_bundle(a, scene), pc.build_elements(scene, a), "cache")
    o1 = chain.get_output_rays()
    assert chain.get_output_rays() is o1
    chain.rotate_OE(1, "pitch", 0.01)
    o2 = chain.get_output_rays()
    assert o2 is not o1
    assert np.abs(o2[-1].points() - o1[-1].points()).max() > 1e-3
    chain.rotate_OE(1, "pitch", -0.01)
    o3 = chain.get_output_rays()
    assert np.abs(o3[-1].points() - a["out2_point"]).max() <= 1e-8
    lst = chain.get_OE_loop_list(2, "shift_normal", np.linspace(-0.1, 0.1, 3))
    assert len(lst) == 3 and lst[1].loop_variable_value == 0.0
    assert np.abs(lst[1].get_output_rays()[-1].points() - a["out2_point"]).max() <= 1e-8
    lst = chain.get_source_loop_list("tilt_in_plane", [0.0, 0.01])
    assert np.abs(lst[0].get_output_rays()[-1].points() - a["out2_point"]).max() <= 1e-8
    assert len(lst[1].get_output_rays()[-1]) > 0
    lst = chain.get_source_loop_list("shift_vert", [0.0, 0.05])
    assert np.abs(lst[0].get_output_rays()[-1].points() - a["out2_point"]).max() <= 1e-8
    with pytest.raises(ValueError):
        chain.rotate_OE(1, "sideways", 1.0)
    with pytest.raises(TypeError):
        chain.source_rays = "nope"


def test_zernike_polynomial_tables_match_recurrences():
    """The monomial expansion handed to the kernels (ModuleDefects.zernike_monomials, Zernike._abi_table) equals
    the reference's recurrences (fixture zernike_tierA.npz, generated with no stand-in) for every (n, m) up to
    order 9, values and both gradients."""
    import ART.ModuleDefects as mdef
    import ART.ModuleSupport as msupp
    from attosecondraytracing_amd import _abi
    _, a = load_golden("zernike_tierA")
    x, y = a["x"], a["y"]
    Zm = mdef.zernike_monomials(9)
    for i, (n, m) in enumerate(a["nm"]):
        C = Zm[(int(n), int(m))].astype(float)
        P, Q = np.meshgrid(np.arange(C.shape[0]), np.arange(C.shape[1]), indexing="ij")
        val = (C[None] * x[:, None, None] ** P[None] * y[:, None, None] ** Q[None]).sum(axis=(1, 2))
        assert np.abs(val - a["val"][i]).max() <= 1e-12 * max(1.0, np.abs(a["val"][i]).max())
    # the packed table: offset and slopes of a defect vs Zernike.get_offset / get_normal of the reference
    S = msupp.SupportRectangle(40, 30)
    Zd = mdef.Zernike(S, {(int(r[0]), int(r[1])): float(r[2]) for r in a["defect_coeffs"]})
    t = Zd._abi_table()
    D = _abi.ART_ZERN_DIM
    R, N = t[0], int(t[1])
    A, GX, GY = (t[2 + k * D * D: 2 + (k + 1) * D * D].reshape(D, D) for k in range(3))
    px, py = a["defect_points"][:, 0] / R, a["defect_points"][:, 1] / R

    def ev(M):
        P, Q = np.meshgrid(np.arange(D), np.arange(D), indexing="ij")
        return (M[None] * px[:, None, None] ** P[None] * py[:, None, None] ** Q[None]).sum(axis=(1, 2))
    assert np.abs(ev(A) - a["defect_offset"]).max() <= 1e-15
    assert np.abs(-ev(GX) / R - a["defect_normal"][:, 0]).max() <= 1e-15
    assert np.abs(-ev(GY) / R - a["defect_normal"][:, 1]).max() <= 1e-15


def test_save_and_load_results(twin, tmp_path, monkeypatch):
    """save_compressed / load_compressed (ART/ModuleProcessing.py:612-633) round-trip a kept_data dictionary with
    device-resident bundles: archived as host arrays, usable again after loading."""
    import ART.ModuleProcessing as mp
    import ART.ModuleOpticalChain as moc
    monkeypatch.chdir(tmp_path)
    scene, a = load_golden("c2_fxf_chain05")
    chain = moc.OpticalChain(pc.source_bundle(a, scene), pc.build_elements(scene, a), "archive me")
    out = chain.get_output_rays()
    mp.save_compressed({"OpticalChain": [chain], "x": 1.5}, "kept")
    back = mp.load_compressed("kept_0")
    ch = back["OpticalChain"][0]
    assert back["x"] == 1.5 and ch.description == "archive me"
    o2 = ch._output_rays
    assert len(o2[-1]) == len(out[-1]) == 490
    assert np.array_equal(o2[-1].points(), out[-1].points())
    assert np.array_equal(o2[-1].path_segments(), out[-1].path_segments())
    assert [r.number for r in o2[-1][:3]] == [r.number for r in out[-1][:3]]


def test_zernike_gradient_module_matches_reference():
    """ART.recursive_zernike_generator.zernike_gradient (API parity module) against values produced by the
    reference's own generator (fixture zernike_tierA: no stand-in module involved)."""
    from ART.recursive_zernike_generator import zernike_gradient
    _, a = load_golden("zernike_tierA")
    N = int(a["nm"][:, 0].max())
    Z, GX, GY = zernike_gradient(a["x"], a["y"], N)
    assert [Z[tuple(k)][0][0] for k in a["nm"]] == list(range(len(a["nm"])))      # generation-order index
    for k, (n, m) in enumerate(a["nm"]):
        scale = max(1.0, np.abs(a["val"][k]).max(), np.abs(a["gx"][k]).max(), np.abs(a["gy"][k]).max())
        assert np.abs(Z[(n, m)][0][1] - a["val"][k]).max() <= 1e-13 * scale
        assert np.abs(GX[(n, m)][0][1] - a["gx"][k]).max() <= 1e-13 * scale
        assert np.abs(GY[(n, m)][0][1] - a["gy"][k]).max() <= 1e-13 * scale
    assert (1, 1) in zernike_gradient(a["x"], a["y"], 0)[0] and (2, 2) in zernike_gradient(a["x"], a["y"], 0)[0]


def test_lazy_history_is_bit_identical_and_traces_once(twin):
    """`history="lazy"`: the wanted bundle is traced alone; every other entry (and the Ray.path tuples of the wanted
    one, which need its parents) appears after ONE re-trace with the full history, bit-identical to a plain trace."""
    import ART.ModuleProcessing as mp
    import ART.ModuleOpticalChain as moc
    scene, a = load_golden("c3_twisted_chain04")
    els = pc.build_elements(scene, a)
    src = pc.source_bundle(a, scene)
    full = mp.RayTracingCalculation(src, els)
    lazy = mp.RayTracingCalculation(src, els, history="lazy")
    assert isinstance(lazy, mp.LazyHistory) and len(lazy) == 3 and lazy.retraces == 0
    last = lazy[-1]
    assert lazy.retraces == 0 and lazy[2] is last
    assert np.array_equal(last.numbers(), full[-1].numbers()) and np.array_equal(last.points(), full[-1].points())
    assert np.array_equal(last.paths_total(), full[-1].paths_total()) and lazy.retraces == 0
    segs = last.path_segments()                         # needs the parents: one re-trace
    assert lazy.retraces == 1 and np.array_equal(segs, full[-1].path_segments())
    assert last[0].path == full[-1][0].path and len(last[0].path) == 4
    for k in range(3):
        assert np.array_equal(lazy[k].points(), full[k].points()) and np.array_equal(lazy[k].numbers(), full[k].numbers())
    assert lazy.retraces == 1 and lazy[-1] is last and [len(b) for b in lazy] == [len(b) for b in full]
    # want = an inner bundle; slices; archives
    inner = mp.LazyHistory(src, els, want=1)
    assert inner.retraces == 0 and np.array_equal(inner[1].vectors(), full[1].vectors())
    assert [len(b) for b in inner[0:2]] == [len(full[0]), len(full[1])] and inner.retraces == 1
    import pickle
    again = pickle.loads(pickle.dumps(mp.RayTracingCalculation(src, els, history="lazy")))
    assert [b.alive.sum().item() for b in again._bundles] == [len(b) for b in full]
    # through the chain cache: a lazy result serves the plain call, ARTmain-style
    chain = moc.OpticalChain(src, els)
    o1 = chain.get_output_rays(history="lazy", want=-1)
    assert isinstance(o1, mp.LazyHistory) and chain.get_output_rays() is o1
    many = moc.trace_chain_list([moc.OpticalChain(src, els), moc.OpticalChain(src, els)], history="lazy")
    assert all(isinstance(o, mp.LazyHistory) and o.retraces == 0 for o in many)
    assert np.array_equal(many[1][-1].points(), full[-1].points()) and np.array_equal(many[0][0].points(), full[0].points())
    with pytest.raises(ValueError):
        mp.RayTracingCalculation(src, els, history="sometimes")
    # a chain that is modified AFTER its lazy history was handed out: the bundle already traced stays what it was, the
    # rest of the history can no longer be traced from the old scene -- a real exception, not a silently mixed history
    moved = moc.OpticalChain(src, els)
    held = moved.get_output_rays(history="lazy")
    before = held[-1].points()
    moved.optical_elements[1].shift_along_normal(0.25)
    with pytest.raises(RuntimeError, match="modified after this history was handed out"):
        held[0]
    assert np.array_equal(held[-1].points(), before)
    fresh = moved.get_output_rays(history="lazy")           # the chain itself re-traces the modified scene
    assert fresh is not held and not np.array_equal(fresh[-1].points(), before) and len(fresh[0]) == len(full[0])


def test_lazy_history_holds_no_reference_cycle(twin):
    """A bundle handed out by a lazy history refers to it WEAKLY (plus the recipe of the trace): dropping chain and
    history frees the traced bundles at once, without waiting for Python's cyclic collector -- 650 MB of device memory
    per 1e7-ray bundle; and a caller who kept only the bundle still gets its parents (Ray.path tuples) re-traced."""
    import gc
    import weakref
    import ART.ModuleProcessing as mp
    import ART.ModuleOpticalChain as moc
    scene, a = load_golden("c3_twisted_chain04")
    els = pc.build_elements(scene, a)
    src = pc.source_bundle(a, scene)
    full = mp.RayTracingCalculation(src, els)
    gc.collect()
    gc.disable()
    try:
        chain = moc.OpticalChain(src, els)
        hist = chain.get_output_rays(history="lazy")
        last = hist[-1]
        ref_last, ref_hist = weakref.ref(last), weakref.ref(hist)
        del hist, chain
        assert ref_hist() is None and ref_last() is last          # the history went with the chain; the kept bundle stays
        assert last[0].path == full[-1][0].path and len(last[0].path) == 4      # parents re-traced from the recipe
        del last
        assert ref_last() is None                                 # ... and the bundle goes without the collector's help
    finally:
        gc.enable()


def test_list_analysis_on_the_twin(twin):
    """ARTmain.analyse_chain_list: one device analysis for a whole loop list == run_ART chain by chain == NumPy."""
    import scene_cases
    scene_cases.run_list_analysis()


@pytest.mark.parametrize("name", ["analysis_c2", "analysis_c3"])
def test_list_analysis_matches_the_reference_for_every_chain(twin, name):
    """analyse_chain_list against the reference's FindOptimalDistance / GetResultSummary / getETransmission for all 11 C2
    and all 10 C3 chains (fixtures generated by running the reference: tests/golden/generate_analysis_goldens.py)."""
    import scene_cases
    scene_cases.run_analysis_goldens(name)


def test_list_analysis_edges_on_the_twin(twin):
    import scene_cases
    scene_cases.run_list_analysis_edges()


def test_guide_rays_on_the_twin(twin):
    import scene_cases
    scene_cases.run_guides()


def test_lockstep_placement_on_the_twin(twin):
    import scene_cases
    scene_cases.run_lockstep_placement()


def test_autofocus_of_a_tight_focus(twin):
    """ONE set of moments, taken at the start pose, serves every level of the autofocus search (mp._optimise_many).  A plane
    wave 0.05 deg off the axis of a parabola focuses to a 0.2-um spot 10 mm from the start pose, where the spot is 1 mm wide:
    the variances cancel by (1 mm / 0.2 um)^2 = 3e7 there.  The spot size and duration the moment form reports at its
    optimum must be those of a direct read-out at that position, and no position of a finer grid around it may be better
    by more than the finest level's step."""
    import ART.ModuleMirror as mmirror
    import ART.ModuleSupport as msupp
    import ART.ModuleProcessing as mp
    import ART.ModuleDetector as mdet
    SP = {"Divergence": 0, "SourceSize": 20, "Wavelength": 800e-6, "DeltaFT": 0.5, "NumberRays": 800}
    ch = mp.OEPlacement(SP, [mmirror.MirrorParabolic(100, 0, msupp.SupportRound(15))], [300], [0.05], Description="tight")
    last = ch.get_output_rays()[-1]
    det = mdet.Detector(np.asarray(ch.optical_elements[-1].position, float))
    det.autoplace(last, 90.0)                       # 10 mm short of the focus
    D, spot, dur = mp.FindOptimalDistance(det, last, "intensity", 20.0, 3, False, False)      # finest step: 2 um
    assert 99.5 < D.get_distance() < 100.5 and spot < 1e-3
    direct_spot = mp.StandardDeviation(list(D.get_PointList2DCentre(last)))
    direct_dur = mp.StandardDeviation(list(D.get_Delays(last)))
    assert abs(spot - direct_spot) <= 1e-6 * direct_spot and abs(dur - direct_dur) <= 1e-6 * direct_dur
    fits = []
    for s_ in np.linspace(-4e-3, 4e-3, 9):          # position by position, the reference's way, on a 1-um grid
        here = D.copy_detector()
        here.shiftByDistance(float(s_))
        fits.append((mp.StandardDeviation(list(here.get_PointList2DCentre(last))) ** 2
                     * mp.StandardDeviation(list(here.get_Delays(last))), float(s_)))
    assert abs(min(fits)[1]) <= 2e-3 + 1e-9, min(fits)
