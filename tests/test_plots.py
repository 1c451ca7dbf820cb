"""CPU: the matplotlib adaptors (attosecondraytracing_amd/_plots.py) draw what the reference draws -- checked on the
DATA inside the figures (scatter offsets, colour arrays, legend text) against the oracle, on the Agg backend."""
import matplotlib
matplotlib.use("Agg")
import numpy as np
import pytest

from conftest import load_golden
import parity_common as pc
from oracle import art_oracle as orc


@pytest.fixture(scope="module")
def twin():
    from twin_backend import TwinBackend
    from attosecondraytracing_amd import _lib
    old = _lib._BACKEND
    _lib._BACKEND = TwinBackend()
    yield _lib._BACKEND
    _lib._BACKEND = old


@pytest.fixture(scope="module")
def scene(twin):
    return build_plot_scene()


def build_plot_scene():
    """Golden chain c3_twisted_chain04 traced by whatever backend is active + the oracle's view of the same scene
    (shared with the GPU suite, tests/test_gpu_endtoend.py)."""
    import ART.ModuleDetector as mdet
    import ART.ModuleOpticalChain as moc
    sc, a = load_golden("c3_twisted_chain04")
    els = pc.build_elements(sc, a)
    src = pc.source_bundle(a, sc)
    chain = moc.OpticalChain(src, els)
    last = chain.get_output_rays()[-1]
    d = sc["detector"]
    D = mdet.Detector(np.array(d["refpoint"]), np.array(d["centre"]), np.array(d["normal"]))
    Do = orc.Detector(np.array(d["centre"]), np.array(d["normal"]), np.array(d["refpoint"]))
    src_o = orc.make_bundle(a["src_point"], a["src_vector"], a["src_number"], a["src_intensity"], sc.get("wavelength"))
    last_o = orc.ray_tracing_calculation(src_o, orc.elements_from_scene(sc, a))[-1]
    return {"chain": chain, "last": last, "D": D, "Do": Do, "last_o": last_o, "scene": sc, "a": a}


def test_spot_diagram_shows_the_reference_points(scene):
    import ART.ModuleAnalysisAndPlots as mplots
    import matplotlib.pyplot as plt
    P2 = orc.detector_points2dcentre(scene["Do"], scene["last_o"]) * 1e3
    delays = orc.detector_delays(scene["Do"], scene["last_o"])
    for coded in (None, "Delay", "Intensity", "Incidence"):
        fig = mplots.SpotDiagram(scene["last"], scene["D"], DrawAiryAndFourier=True, ColorCoded=coded)
        sc = fig.axes[0].collections[0]
        assert np.abs(np.asarray(sc.get_offsets()) - P2).max() <= 1e-6
        if coded == "Delay":
            assert np.abs(np.asarray(sc.get_array()) - delays).max() <= 1e-6
        elif coded == "Intensity":
            assert np.abs(np.asarray(sc.get_array()) - scene["last_o"].intensity).max() <= 1e-12
        elif coded == "Incidence":
            assert np.abs(np.asarray(sc.get_array()) - np.rad2deg(scene["last_o"].incidence)).max() <= 1e-7
        text = fig.axes[0].get_legend().get_texts()[0].get_text()
        assert "{:.1f} μm SD".format(orc.standard_deviation(P2 / 1e3) * 1e3) in text
        plt.close(fig)
    x, y, size, sd = mplots._getDetectorPoints(scene["last"], scene["D"])
    assert abs(sd - scene["scene"]["SpotSizeSD"]) <= 1e-9 and len(x) == len(P2)


def test_spot_diagram_key_press_moves_the_detector(scene):
    import ART.ModuleAnalysisAndPlots as mplots
    import matplotlib.pyplot as plt
    fig = mplots.SpotDiagram(scene["last"], scene["D"], ColorCoded="Delay")
    before = np.asarray(fig.axes[0].collections[0].get_offsets()).copy()
    d0 = scene["D"].get_distance()

    class Ev:
        key = "right"
    fig._art_press(Ev())
    after = np.asarray(fig.axes[0].collections[0].get_offsets())
    assert np.abs(after - before).max() > 0
    assert abs(scene["D"].get_distance() - d0) == 0          # the caller's detector is not moved
    Ev.key = "x"
    fig._art_press(Ev())                                     # other keys are ignored
    plt.close(fig)


def test_delay_graph_and_mirror_projection(scene):
    import ART.ModuleAnalysisAndPlots as mplots
    import matplotlib.pyplot as plt
    delays = orc.detector_delays(scene["Do"], scene["last_o"])
    fig = mplots.DelayGraph(scene["last"], scene["D"], 0.5, DrawAiryAndFourier=True, ColorCoded="Intensity")
    xs, ys, zs = fig.axes[0].collections[0]._offsets3d
    assert np.abs(np.asarray(zs) - delays).max() <= 1e-6
    plt.close(fig)
    # impact points on the second toroid, in its support frame, against the oracle's optic-frame hit points
    a, sc = scene["a"], scene["scene"]
    k = 2
    fig = mplots.MirrorProjection(scene["chain"], k, scene["D"], "Delay")
    pts = np.asarray(fig.axes[0].collections[0].get_offsets())
    e = sc["elements"][k]
    hit_lab = a[f"out{k}_point"] - np.array(e["position"])
    from attosecondraytracing_amd.ModuleGeometry import frame_maps
    fwd, _ = frame_maps(np.array(e["normal"]), np.array(e["majoraxis"]))
    assert np.abs(pts - (hit_lab @ fwd.T)[:, :2]).max() <= 1e-9
    assert len(fig.axes[0].patches) >= 1                     # the support contour
    assert np.abs(pts[:, 0]).max() <= e["support"]["p"][0] / 2 + 1e-9
    plt.close(fig)
    with pytest.raises(ValueError):
        mplots.MirrorProjection(scene["chain"], k, None, "Delay")


def test_large_bundles_are_down_sampled(scene, monkeypatch):
    import ART.ModuleAnalysisAndPlots as mplots
    from attosecondraytracing_amd import _plots
    import matplotlib.pyplot as plt
    monkeypatch.setattr(_plots, "MAX_POINTS", 50)
    fig = mplots.SpotDiagram(scene["last"], scene["D"])
    assert len(fig.axes[0].collections[0].get_offsets()) <= 50
    text = fig.axes[0].get_legend().get_texts()[0].get_text()
    assert "{:.1f} μm SD".format(scene["scene"]["SpotSizeSD"] * 1e3) in text     # statistics still over all rays
    plt.close(fig)


# ---- the 3-D scene render (RayRenderGraph, ART/ModuleAnalysisAndPlots.py:529-673) against tests/golden/render_grids.npz,
# made by running the reference's own _get_grid / _Contour_points / get_grid3D / _RenderRays / _RenderOpticalElement
def _render_golden():
    return load_golden("render_grids")


def test_support_sample_points_are_the_references():
    sc, z = _render_golden()
    for name, d in sc["supports"].items():
        S = pc.build_support(d)
        for n in (200, 37):
            got = np.array(S._get_grid(n)).reshape(-1, 2)
            assert got.shape == z[f"sup_{name}_grid{n}"].shape and np.abs(got - z[f"sup_{name}_grid{n}"]).max() <= 1e-12
            assert all(S._IncludeSupport(p) for p in got)
        pts, loops = S._Contour_points(40, edges=True)
        assert loops == d["contour40_edges"]
        assert np.abs(np.array(pts).reshape(-1, 2) - z[f"sup_{name}_contour40"]).max() <= 1e-12
        assert len(S._Contour_points(40)) == len(pts)


def test_surface_sample_points_are_the_references():
    sc, z = _render_golden()
    for name, d in sc["optics"].items():
        O = pc.build_optic(d)
        pts, loops = O.get_grid3D(300, edges=True)
        ref = z[f"opt_{name}_grid300"]
        got = np.array(pts).reshape(-1, 3)
        assert got.shape == ref.shape and np.abs(got - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max()), name
        assert loops == d["grid300_edges"], name
        assert len(O.get_grid3D(300)) == len(ref)
    import ART.ModuleMirror as mmirror
    import ART.ModuleDefects as mdef
    par = pc.build_optic(sc["optics"]["parabola"])
    deformed = mmirror.DeformedMirror(par, [mdef.Zernike(par.support, {(2, 1): 1e-4})])
    assert np.array_equal(np.array(deformed.get_grid3D(100)), np.array(par.get_grid3D(100)))   # the render ignores defects


def test_render_scene_is_the_references_geometry(scene):
    from attosecondraytracing_amd import _plots
    sc, z = _render_golden()
    assert sc["c3"]["fixture"] == "c3_twisted_chain04"
    r = _plots.render_scene(scene["chain"], None, maxRays=5000, OEpoints=sc["c3"]["OEpoints"])
    assert abs(r["EndDistance"] - sc["c3"]["EndDistance"]) <= 1e-9
    assert len(r["segments"]) == 4 and len(r["optics"]) == 3
    for k, seg in enumerate(r["segments"]):
        ref = z[f"c3_segments{k}"]
        assert seg.shape == ref.shape and np.abs(seg - ref).max() <= 1e-10 * np.abs(ref).max(), k
    for k, cloud in enumerate(r["optics"]):
        ref = z[f"c3_optic{k}"]
        assert cloud.shape == ref.shape and np.abs(cloud - ref).max() <= 1e-10 * np.abs(ref).max(), k
    # down-sampled: at most maxRays rays per stage, each segment still joins one ray's points in consecutive bundles
    few = _plots.render_scene(scene["chain"], 25.0, maxRays=40, OEpoints=100)
    history = [scene["chain"].source_rays] + list(scene["chain"].get_output_rays())
    for k, seg in enumerate(few["segments"]):
        assert len(seg) == 80
        nxt = history[min(k + 1, 3)]
        ends = seg[1::2] if k < 3 else seg[0::2]
        d = np.abs(ends[:, None, :] - nxt.points()[None, :, :]).max(axis=2).min(axis=1)
        assert d.max() == 0.0
    assert np.allclose(np.linalg.norm(few["segments"][3][1::2] - few["segments"][3][0::2], axis=1), 25.0, rtol=1e-12)


def test_render_scene_matches_rays_by_number_when_slots_differ(scene):
    """Bundles built from Ray lists do not share their slots: the segments are then matched by ray number."""
    from attosecondraytracing_amd import _plots
    from attosecondraytracing_amd.bundle import RayBundle
    history = [scene["chain"].source_rays] + list(scene["chain"].get_output_rays())
    same = _plots._ray_segments(history, 10.0, 5000)
    rebuilt = [history[0]] + [RayBundle.from_ray_list(list(b)) for b in history[1:]]
    assert not _plots._same_slots(rebuilt[0], rebuilt[1])
    other = _plots._ray_segments(rebuilt, 10.0, 5000)
    for a, b in zip(same, other):      # (a Ray renormalises its vector: the last stage agrees to rounding only)
        assert a.shape == b.shape and np.abs(a - b).max() <= 1e-12 * np.abs(a).max()


def test_ray_render_graph_draws_the_scene(scene):
    import ART.ModuleAnalysisAndPlots as mplots
    import matplotlib.pyplot as plt
    fig = mplots.RayRenderGraph(scene["chain"], maxRays=50, OEpoints=200, draw_mesh=True, cycle_ray_colors=True)
    ax = fig.axes[0]
    lines = [c for c in ax.collections if type(c).__name__ == "Line3DCollection"]
    clouds = [c for c in ax.collections if type(c).__name__ == "Path3DCollection"]
    assert len(lines) == 4 and len(clouds) == 3
    meshes = [c for c in ax.collections if type(c).__name__ == "Poly3DCollection"]
    assert len(meshes) == 3                                  # draw_mesh: one triangulated surface per optic
    for OE, cloud, tri in zip(scene["chain"].optical_elements, fig._art_scene["optics"], fig._art_scene["triangles"]):
        assert len(tri) > 50 and tri.min() >= 0 and tri.max() < len(cloud)
        pts = np.asarray(OE.type.get_grid3D(200), dtype=float)[:, :2] - np.asarray(OE.type.get_centre(), dtype=float)[:2]
        assert all(OE.type.support._IncludeSupport(m) for m in pts[tri].mean(axis=1))     # no triangle bridges the hole
    assert [len(s) // 2 for s in fig._art_scene["segments"]] == [50, 50, 50, 50]
    assert len(mplots.generate_distinct_colors(5)) == 5
    spans = np.array([np.diff(ax.get_xlim())[0], np.diff(ax.get_ylim())[0], np.diff(ax.get_zlim())[0]])
    ratio = spans / np.asarray(ax.get_box_aspect())
    assert np.abs(ratio / ratio[0] - 1).max() <= 1e-9                                          # equal scales on the three axes
    plt.close(fig)
    fig = scene["chain"].render()
    assert len(fig._art_scene["optics"]) == 3
    plt.close(fig)
