"""CPU: accuracy against a higher-precision truth.  The reference solves the torus through an expanded quartic
whose coefficients reach 1e15 and loses ~1e-10 mm on the hit distance; the kernels' convex-Newton
solver works on the well-conditioned implicit function.  For the toroid hits of the C2 / C3 fixtures the segment
length of every ray is refined in 80-bit long double (Newton on (sqrt(x^2+z^2)-R)^2 + y^2 - r^2 along the ray) and
both the reference's value and ours (CPU twin of the kernels) are compared with it.

Round 3: the same for EVERY optic kind, the Zernike / height-map deformation and the reflection
(tests/truth_common.py: one element acting on rays, in long double from the element's fp64 parameters): every element
of every golden chain, the reference judged on its own incoming rays (fixture arrays) and the product on its own, and
the loosened seeds of the differential fuzz harness adjudicated by that truth (tests/fuzz_common.py)."""
import numpy as np
import pytest

from conftest import load_golden
import parity_common as pc

LD = np.longdouble


@pytest.fixture(scope="module")
def twin():
    from twin_backend import TwinBackend
    from attosecondraytracing_amd import _lib
    old = _lib._BACKEND
    _lib._BACKEND = TwinBackend()
    yield
    _lib._BACKEND = old


def _truth_t(A, u, R, r, t0):
    A, u = A.astype(LD), u.astype(LD)
    t = t0.astype(LD)
    R, r = LD(R), LD(r)
    for _ in range(6):
        P = A + t[:, None] * u
        rho = np.sqrt(P[:, 0] ** 2 + P[:, 2] ** 2)
        F = (rho - R) ** 2 + P[:, 1] ** 2 - r ** 2
        dF = 2 * ((rho - R) * (P[:, 0] * u[:, 0] + P[:, 2] * u[:, 2]) / rho + P[:, 1] * u[:, 1])
        t = t - F / dF
    return t


@pytest.mark.parametrize("name", ["c2_fxf_chain05", "c3_twisted_chain04"])
def test_torus_hit_distance_vs_long_double_truth(twin, name):
    check_against_truth(name)


def check_against_truth(name):
    """Shared with the GPU suite (tests/test_gpu_parity.py), where the hardware-seeded 1/x, 1/sqrt(x) paths run."""
    import ART.ModuleProcessing as mp
    from attosecondraytracing_amd import ModuleGeometry as mgeo
    if np.finfo(LD).eps > 1e-18:
        pytest.skip("no extended precision long double on this platform")
    scene, a = load_golden(name)
    els = pc.build_elements(scene, a)
    out = mp.RayTracingCalculation(pc.source_bundle(a, scene), els)
    worst_ref, worst_ours = 0.0, 0.0
    for k in (1, 2):                                     # the two toroids
        oe = els[k]
        R, r = oe.type.majorradius, oe.type.minorradius
        fwd, _ = mgeo.frame_maps(oe.normal, oe.majoraxis)
        assert np.array_equal(out[k].numbers(), a[f"out{k}_number"])

        def optic_frame(points, vectors):
            Ao = (points.astype(LD) - np.asarray(oe.position, float).astype(LD)) @ fwd.T.astype(LD) \
                + oe.type.get_centre().astype(LD)
            return Ao, vectors.astype(LD) @ fwd.T.astype(LD)
        # each implementation is judged on ITS OWN incoming rays (bundle after element k-1), frame change in long double
        A, u = optic_frame(a[f"out{k-1}_point"], a[f"out{k-1}_vector"])
        t_ref = a[f"out{k}_path"][:, -1]
        worst_ref = max(worst_ref, float(np.abs(t_ref.astype(LD) - _truth_t(A, u, R, r, t_ref)).max()))
        A, u = optic_frame(out[k - 1].points(), out[k - 1].vectors())
        t_ours = out[k].path_segments()[:, -1]
        worst_ours = max(worst_ours, float(np.abs(t_ours.astype(LD) - _truth_t(A, u, R, r, t_ours)).max()))
    # what remains for the kernels is the fp64 rounding of positions of a ~1000 mm scene (ulp 1e-13 mm) seen at
    # 80 deg grazing incidence (x 1/cos 80 deg ~ 6) through two frame changes
    assert worst_ours <= 2e-11, worst_ours
    assert worst_ref <= 2e-9, worst_ref                  # the reference's own error (~1.8e-10 mm and up)
    assert worst_ours < worst_ref
    print(f"{name}: max |t - truth|  reference {worst_ref:.2e} mm, kernels {worst_ours:.2e} mm")


def check_every_element_against_truth(name):
    """Every element of a golden chain: hit point, direction, segment length and incidence of the reference (fixture)
    and of the product (active backend), each on its OWN incoming rays, against tests/truth_common.py.  Shared with the
    GPU suite."""
    import ART.ModuleProcessing as mp
    import truth_common as T
    import fuzz_common as fz
    from oracle import art_oracle as orc
    if not T.HAVE_LD:
        pytest.skip("no extended precision long double on this platform")
    scene, a = load_golden(name)
    ign = scene.get("IgnoreDefects", True)
    els_o = orc.elements_from_scene(scene, a)
    src = pc.source_bundle(a, scene)
    outs = mp.RayTracingCalculation(src, pc.build_elements(scene, a), IgnoreDefects=ign)
    scale = pc.scene_scale(a, scene)
    worst = {"reference": dict.fromkeys(fz.LOCAL_TOL, 0.0), "product": dict.fromkeys(fz.LOCAL_TOL, 0.0)}
    ref_prev = (a["src_number"], a["src_point"], a["src_vector"])
    our_prev = (a["src_number"], a["src_point"], src.vectors())
    for k, E in enumerate(els_o):
        num = a[f"out{k}_number"]
        assert np.array_equal(outs[k].numbers(), num)
        if len(num) == 0:
            break
        grazing = a[f"out{k}_incidence"] >= fz.GRAZING       # conditioning of the hit itself (1 / cos): not judged
        sides = (("reference", ref_prev, a[f"out{k}_point"], a[f"out{k}_vector"], a[f"out{k}_path"][:, -1], a[f"out{k}_incidence"]),
                 ("product", our_prev, outs[k].points(), outs[k].vectors(), outs[k].path_segments()[:, -1], outs[k].incidences()))
        for who, prev, pts, vec, seg, inc in sides:
            sel = np.searchsorted(prev[0], num)
            P, v, t, i = T.element_truth(E, prev[1][sel], prev[2][sel], seg, ign)
            m = ~grazing
            if not m.any():
                continue
            e = {"pos": np.abs(pts - P)[m].max() / scale, "dir": np.abs(vec - v)[m].max(), "seg": np.abs(seg - t)[m].max() / scale,
                 "inc": np.abs(inc - i)[m].max()}
            for key, val in e.items():
                worst[who][key] = max(worst[who][key], float(val))
        ref_prev = (num, a[f"out{k}_point"], a[f"out{k}_vector"])
        our_prev = (num, outs[k].points(), outs[k].vectors())
    noise = sum(fz.pose_noise(e) for e in scene["elements"])
    rcs = [fz.curvature_radius(e) for e in scene["elements"]]
    curv = 4.0 * max(worst["product"]["pos"], worst["product"]["seg"]) * scale / min([r for r in rcs if r] or [np.inf])
    for key, lim in fz.LOCAL_TOL.items():
        assert worst["product"][key] <= lim + noise + (curv if key in ("dir", "inc") else 0.0), (name, key, worst["product"][key])
        assert worst["reference"][key] <= 1e-10 + noise, (name, key, worst["reference"][key])     # the reference's own accuracy
    return worst


_WORST = {}


@pytest.mark.parametrize("name", [n for n in __import__("conftest").chain_golden_names()])
def test_every_element_vs_long_double_truth(twin, name):
    w = check_every_element_against_truth(name)
    for who in w:
        for key, v in w[who].items():
            if v > _WORST.setdefault(who, {}).get(key, (0.0, ""))[0]:
                _WORST[who][key] = (v, name)


def test_truth_summary(twin):
    from conftest import report
    for who, w in _WORST.items():
        report(f"[local error vs long-double truth, {who}] " + "  ".join(f"{k} {v:.1e} ({n})" for k, (v, n) in sorted(w.items())))


def test_loosened_fuzz_seeds_are_adjudicated_by_truth(twin):
    """Seeds whose product-vs-oracle difference exceeds 1e-10 (round 2 widened the tolerance for them): ellipsoid ->
    ellipsoid -> sphere chain 60039358 (directions 1.6e-9 apart) and 65.  Under the truth rule the product must be the
    closer one -- it is, by two to three orders of magnitude.  40030221 (frame axes 5e-7 rad from antiparallel) passes on
    the reference's own frame noise."""
    import fuzz_common as fz
    st = {}
    fz.run_differential([60039358, 65, 40030221], stats=st)
    assert set(st["adjudicated_seeds"]) == {60039358, 65}, st
    assert st["adjudicated_worst"]["product"] <= 1e-11 and st["adjudicated_worst"]["oracle"] >= 1e-10, st
