"""CPU: property tests (hypothesis) of the kernel math, run through the product API on the CPU twin -- independent
of the reference: reflected directions are unit vectors obeying v' = v - 2 (n.v) n; hit points lie on the implicit
surface; optical paths equal the geometric distance travelled; a paraboloid sends an on-axis plane wave through its
focus with equal optical paths (Fermat); an ellipsoid images one focus onto the other with equal paths."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st


@pytest.fixture(scope="module")
def twin():
    from twin_backend import TwinBackend
    from attosecondraytracing_amd import _lib
    old = _lib._BACKEND
    _lib._BACKEND = TwinBackend()
    yield
    _lib._BACKEND = old


def _trace(optic, pos, normal, major, pts, vec):
    import ART.ModuleOpticalElement as moe
    import ART.ModuleProcessing as mp
    from attosecondraytracing_amd.bundle import RayBundle
    from attosecondraytracing_amd import ModuleGeometry as mgeo
    oe = moe.OpticalElement(optic, np.asarray(pos, float), np.asarray(normal, float), np.asarray(major, float))
    src = RayBundle.from_arrays(pts, vec, np.arange(len(pts)), np.ones(len(pts)))
    out = mp.RayTracingCalculation(src, [oe])[0]
    fwd, bwd = mgeo.frame_maps(oe.normal, oe.majoraxis)
    return src, out, oe, fwd


@settings(max_examples=25, deadline=None)
@given(st.floats(50, 5000), st.floats(20, 400), st.floats(5, 85), st.integers(0, 2 ** 31 - 1))
def test_torus_hits_lie_on_surface_and_reflect_specularly(twin, R, r, inc_deg, seed):
    import ART.ModuleMirror as mmirror
    import ART.ModuleSupport as msupp
    rng = np.random.default_rng(seed)
    n = 200
    tor = mmirror.MirrorToroidal(R, r, msupp.SupportRectangle(0.8 * r, 0.5 * r))
    a = np.deg2rad(inc_deg)
    normal = np.array([-np.sin(a), 0.0, -np.cos(a)])
    major = np.array([np.cos(a), 0.0, -np.sin(a)])
    dist = 0.5 * r
    pts = rng.normal(0, 0.02 * r, (n, 3))
    vec = np.array([0.0, 0.0, 1.0]) + rng.normal(0, 0.05, (n, 3))
    src, out, oe, fwd = _trace(tor, [0.0, 0.0, dist], normal, major, pts, vec)
    m = out.alive.numpy().astype(bool)
    if m.sum() == 0:
        return
    P = out.data[0:3].numpy().T[m]
    Po = (P - oe.position) @ fwd.T + tor.get_centre()
    F = (np.sqrt(Po[:, 0] ** 2 + Po[:, 2] ** 2) - R) ** 2 + Po[:, 1] ** 2 - r ** 2
    assert np.abs(F).max() <= 1e-9 * (R + r) * r            # on the implicit surface
    assert (Po[:, 2] < -R).all()                             # the reference's side rule
    v_in = src.data[3:6].numpy().T[m]
    v_out = out.data[3:6].numpy().T[m]
    assert np.abs(np.linalg.norm(v_out, axis=1) - 1).max() <= 1e-14
    # specular: v_out - v_in is parallel to the surface normal at the hit point
    nrm = np.array([tor.get_normal(p) for p in Po]) @ fwd   # optic -> lab (fwd is orthogonal: inverse = transpose)
    d = v_out - v_in
    cross = np.cross(d, nrm)
    assert np.abs(cross).max() <= 1e-11
    assert np.abs(np.einsum("ij,ij->i", v_out, nrm) + np.einsum("ij,ij->i", v_in, nrm)).max() <= 1e-12
    # optical path = distance travelled; incidence = angle(-v_in, n)
    seg = np.linalg.norm(P - src.data[0:3].numpy().T[m], axis=1)
    assert np.abs(out.data[6].numpy()[m] - seg).max() <= 1e-10 * max(1.0, seg.max())
    inc = np.arccos(np.clip(-np.einsum("ij,ij->i", v_in, nrm), -1, 1))
    assert np.abs(out.data[7].numpy()[m] - inc).max() <= 1e-7


@settings(max_examples=20, deadline=None)
@given(st.floats(20, 500), st.floats(0, 120), st.floats(0.01, 0.3))
def test_parabola_focuses_plane_wave_with_equal_paths(twin, feff, offaxis_deg, rel_radius):
    """Fermat: an on-axis plane wave reflected by a paraboloid reaches the focus with equal optical paths."""
    import ART.ModuleMirror as mmirror
    import ART.ModuleSupport as msupp
    import ART.ModuleProcessing as mp
    import ART.ModuleDetector as mdet
    SP = {"Divergence": 0, "SourceSize": 2 * rel_radius * feff, "Wavelength": 800e-6, "DeltaFT": 1, "NumberRays": 300}
    par = mmirror.MirrorParabolic(feff, offaxis_deg, msupp.SupportRound(3 * rel_radius * feff))
    chain = mp.OEPlacement(SP, [par], [2 * feff], [0])
    out = chain.get_output_rays()[-1]
    assert len(out) == 299
    # focus of the paraboloid x^2 + y^2 = 2 p z is (0, 0, p/2) in the optic frame; any plane through it works
    from attosecondraytracing_amd import ModuleGeometry as mgeo
    oe = chain.optical_elements[0]
    fwd, _ = mgeo.frame_maps(oe.normal, oe.majoraxis)
    focus = fwd.T @ (np.array([0.0, 0.0, par.p / 2]) - par.get_centre()) + np.asarray(oe.position, float)
    central = mp.FindCentralRay(out).vector
    det = mdet.Detector(np.asarray(oe.position, float), focus, -central)
    P3 = det.get_PointList3D(out)
    assert np.abs(P3 - focus).max() <= 1e-9 * feff           # all rays through the focus
    opl = det.get_OpticalPaths(out)
    assert (opl.max() - opl.min()) <= 1e-10 * opl.mean()      # equal optical paths


@settings(max_examples=15, deadline=None)
@given(st.floats(100, 1000), st.floats(20, 160))
def test_ellipsoid_images_focus_to_focus(twin, f_o, offaxis_deg):
    f_i = f_o   # symmetric case: the mirror centre is the end of the minor axis, incidence = half the off-axis angle
    import ART.ModuleMirror as mmirror
    import ART.ModuleSupport as msupp
    import ART.ModuleProcessing as mp
    import ART.ModuleDetector as mdet
    ell = mmirror.MirrorEllipsoidal(msupp.SupportRound(0.02 * min(f_o, f_i)), OffAxisAngle=offaxis_deg, f_object=f_o, f_image=f_i)
    SP = {"Divergence": 0.005, "SourceSize": 0, "Wavelength": 50e-6, "DeltaFT": 1, "NumberRays": 200}
    chain = mp.OEPlacement(SP, [ell], [f_o], [offaxis_deg / 2])
    out = chain.get_output_rays()[-1]
    if len(out) < 150:
        return   # support clipped most of the bundle for this geometry
    # second focus: on the line of the central reflected ray, f_i behind the mirror centre
    from attosecondraytracing_amd import ModuleGeometry as mgeo
    oe = chain.optical_elements[0]
    fwd, _ = mgeo.frame_maps(oe.normal, oe.majoraxis)
    c = np.sqrt(ell.a ** 2 - ell.b ** 2)
    foci = [fwd.T @ (np.array([sx * c, 0.0, 0.0]) - ell.get_centre()) + np.asarray(oe.position, float) for sx in (1, -1)]
    image = max(foci, key=lambda F: np.linalg.norm(F))       # the source sits at the origin = the other focus
    assert min(np.linalg.norm(F) for F in foci) <= 1e-8 * f_o
    det = mdet.Detector(np.asarray(oe.position, float), image, -mp.FindCentralRay(out).vector)
    P3 = det.get_PointList3D(out)
    assert np.abs(P3 - image).max() <= 1e-8 * (f_o + f_i)
    opl = det.get_OpticalPaths(out)
    assert (opl.max() - opl.min()) <= 1e-10 * opl.mean()
    assert abs(opl.mean() - 2 * ell.a) <= 1e-9 * ell.a         # string construction: f_o + f_i = 2a


@settings(max_examples=20, deadline=None)
@given(st.floats(50, 800), st.floats(0, 75), st.floats(20, 400), st.integers(0, 2 ** 31 - 1))
def test_plane_mirror_images_a_point_source(twin, dist_, inc_deg, after, seed):
    """Plane mirror: every reflected ray, traced backwards, passes through the mirror image of the source point, and
    its optical path to any point equals the straight distance from that image point."""
    import ART.ModuleMirror as mmirror
    import ART.ModuleSupport as msupp
    rng = np.random.default_rng(seed)
    S = np.array([0.0, 0.0, 0.0])
    th = np.deg2rad(inc_deg)
    pos = np.array([dist_, 0.0, 0.0])
    normal = np.array([-np.cos(th), np.sin(th), 0.0])            # faces the source under the incidence angle
    major = np.array([np.sin(th), np.cos(th), 0.0])
    n = 200
    d = np.stack([np.ones(n), rng.uniform(-0.02, 0.02, n), rng.uniform(-0.02, 0.02, n)], axis=1)
    d /= np.linalg.norm(d, axis=1)[:, None]
    src, out, oe, _ = _trace(mmirror.MirrorPlane(msupp.SupportRound(1e4)), pos, normal, major, np.tile(S, (n, 1)), d)
    assert len(out) == n
    image = S - 2 * np.dot(S - pos, normal) * normal             # mirror image of the source
    P, V, path = out.points(), out.vectors(), out.paths_total()
    # the reflected rays diverge from the image point: (P - image) is parallel to V, |P - image| is the path so far
    back = P - image
    assert np.abs(np.cross(back, V)).max() <= 1e-10 * dist_
    assert np.abs(np.linalg.norm(back, axis=1) - path).max() <= 1e-10 * dist_
    Q = P + after * V
    assert np.abs(np.linalg.norm(Q - image, axis=1) - (path + after)).max() <= 1e-10 * (dist_ + after)


@settings(max_examples=15, deadline=None)
@given(st.floats(200, 4000), st.floats(1e-4, 2e-3))
def test_sphere_at_normal_incidence_focuses_paraxial_rays_at_half_the_radius(twin, R, h_rel):
    """Concave sphere, rays parallel to its axis at height h: they cross the axis at R/2 minus the spherical
    aberration R/2 (1/cos(asin(h/R)) - 1) -- the exact closed form, hence a known answer for any h."""
    import ART.ModuleMirror as mmirror
    import ART.ModuleSupport as msupp
    h = h_rel * R * np.array([1.0, 3.0, 10.0])
    pts = np.stack([np.full(3, 0.0), h, np.zeros(3)], axis=1)   # rays along +x at heights h above the axis
    d = np.tile(np.array([1.0, 0.0, 0.0]), (3, 1))
    pos = np.array([R, 0.0, 0.0])                               # vertex; centre of curvature at the origin
    src, out, oe, _ = _trace(mmirror.MirrorSpherical(R, msupp.SupportRound(R / 2)), pos, np.array([-1.0, 0.0, 0.0]),
                             np.array([0.0, 1.0, 0.0]), pts, d)
    assert len(out) == 3
    P, V = out.points(), out.vectors()
    t = -P[:, 1] / V[:, 1]                                       # where the reflected ray meets the axis y = 0
    x_cross = P[:, 0] + t * V[:, 0]
    theta = np.arcsin(h / R)
    expect = R / (2 * np.cos(theta))                             # distance of the crossing point from the centre
    assert np.abs(x_cross - expect).max() <= 1e-9 * R
    assert abs(x_cross[0] - R / 2) <= 1e-5 * R                   # paraxial limit


def test_toroid_2f_2f_images_the_source(twin):
    """A toroid from ReturnOptimalToroidalRadii(f, angle) used 2f-2f: the chief ray reaches the mirror after exactly
    2f (OEPlacement puts the mirror there), and a detector 2f behind the mirror sees an image far smaller than the
    footprint of the bundle on the mirror, with the chief ray's total path at 4f to within the cone's asymmetry."""
    import ART.ModuleMirror as mmirror
    import ART.ModuleSupport as msupp
    import ART.ModuleProcessing as mp
    import ART.ModuleDetector as mdet
    f, ang = 300.0, 80.0
    R, r = mmirror.ReturnOptimalToroidalRadii(f, ang)
    SP = {"Divergence": 2e-3, "SourceSize": 0, "Wavelength": 50e-6, "DeltaFT": 0.5, "NumberRays": 400}
    ch = mp.OEPlacement(SP, [mmirror.MirrorToroidal(R, r, msupp.SupportRectangle(150, 30))], [2 * f], [ang])
    out = ch.get_output_rays()[-1]
    assert len(out) == 400
    assert abs(out.paths_total()[0] - 2 * f) <= 1e-10 * f        # ray 0 of the Vogel cone is the chief ray
    assert np.abs(out.points()[0] - np.asarray(ch.optical_elements[0].position, float)).max() <= 1e-9 * f
    D = mdet.Detector(np.asarray(ch.optical_elements[-1].position, float))
    D.autoplace(out, 2 * f)
    opl = np.asarray(D.get_OpticalPaths(out))
    assert abs(opl[0] - 4 * f) <= 1e-4 * f                       # the detector is placed on the MEAN ray, not the chief ray
    spot = mp.StandardDeviation(D.get_PointList2DCentre(out))
    footprint = np.std(out.points(), axis=0).max()
    assert spot < 1e-2 * footprint
