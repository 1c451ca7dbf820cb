#!/usr/bin/env python3
"""Generate golden vectors by RUNNING THE REFERENCE (read-only /root/reference) in the build container.

TEST INFRASTRUCTURE.  Run once here (`python tests/golden/generate_goldens.py`); the resulting
`tests/golden/*.npz` files are committed.  Nothing on the GPU box imports this script or the
reference: the fixtures hold plain arrays (inputs + expected outputs) and a JSON scene description.

What executes: the reference's own `ModuleProcessing.OEPlacement`, `RayTracingCalculation`
(via `OpticalChain.get_output_rays`), `ModuleMirror`/`ModuleMask`/`ModuleSupport`/`ModuleDefects`,
`ModuleDetector` -- unmodified, imported from /root/reference.  The image lacks four third-party
packages the reference imports; they are replaced by the stand-ins in `tests/golden/_standin/`:
  * `quaternion` (numpy-quaternion): a 60-line Hamilton-algebra class of ours.  Consequence: the
    rotation arithmetic inside `RotationAroundAxis` (ART/ModuleGeometry.py:321-329) is the
    stand-in's, everything else (np.roots solvers, acceptance rules, reflection, masks, detector,
    Zernike recurrences) is executed reference code + NumPy/SciPy.  The stand-in is pinned independently:
    tests/test_quaternion_standin.py runs that call sequence on it against SciPy's Rotation.from_rotvec and an
    80-bit Rodrigues formula (<= 4 ulp (1 + |angle|) from the truth, as accurate as SciPy itself), and main()
    the __main__ block below refuses to generate fixtures if a spot check of the same comparison fails.
  * `pyvista`, `pyvistaqt`, `colorcet`: empty modules (only used inside plot functions).
Tier-A fixtures (`zernike_*.npz`) come from reference modules that import with NO stand-in
(`ART/recursive_zernike_generator.py`, `ART/ModuleDefects.py`).

Unseeded randomness in the reference is avoided: no `Fourrier` ctor with its own RNG call is
compared without the generated maps being stored; `np.random.seed` is set before those.
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("ART_REFERENCE", "/root/reference")
OUT = os.environ.get("ART_GOLDEN_OUT", HERE)      # set it to regenerate elsewhere and compare with the committed files
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(HERE, "_standin"))

import matplotlib  # noqa: E402

matplotlib.use("Agg")
import numpy as np  # noqa: E402

import ART.ModuleMirror as mmirror  # noqa: E402
import ART.ModuleMask as mmask  # noqa: E402
import ART.ModuleSupport as msupp  # noqa: E402
import ART.ModuleProcessing as mp  # noqa: E402
import ART.ModuleDefects as mdef  # noqa: E402
import ART.ModuleDetector as mdet  # noqa: E402
import ART.ModuleSource as msource  # noqa: E402
import ART.ModuleOpticalRay as mray  # noqa: E402
import ART.ModuleOpticalElement as moe  # noqa: E402
import ART.ModuleOpticalChain as moc  # noqa: E402
import ART.ModuleGeometry as mgeo  # noqa: E402
import ART.ModuleAnalysisAndPlots as mplots  # noqa: E402


# record the arguments of every reference placement on the chain it returns, so that the fixtures also pin
# OEPlacement (ART/ModuleProcessing.py:32-246): tests re-run the product's OEPlacement with the same arguments
_orig_single = mp._singleOEPlacement


def _recording_single(SourceProperties, OpticsList, DistanceList, IncidenceAngleList, IncidencePlaneAngleList,
                      Description):
    args = {"SourceProperties": {k: (None if v is None else float(v)) for k, v in SourceProperties.items()},
            "DistanceList": [float(v) for v in DistanceList],
            "IncidenceAngleList": [float(v) for v in IncidenceAngleList],
            "IncidencePlaneAngleList": [float(v) for v in IncidencePlaneAngleList]}
    chain = _orig_single(SourceProperties, OpticsList, DistanceList, IncidenceAngleList, IncidencePlaneAngleList,
                         Description)
    chain._placement = args
    return chain


mp._singleOEPlacement = _recording_single


# ----------------------------------------------------------------------------- describing objects
def describe_support(S):
    n = type(S).__name__
    if n == "SupportRound":
        return {"kind": "round", "p": [S.radius]}
    if n == "SupportRoundHole":
        return {"kind": "roundhole", "p": [S.radius, S.radiushole, S.centerholeX, S.centerholeY]}
    if n == "SupportRectangle":
        return {"kind": "rect", "p": [S.dimX, S.dimY]}
    if n == "SupportRectangleHole":
        return {"kind": "recthole", "p": [S.dimX, S.dimY, S.radiushole, S.centerholeX, S.centerholeY]}
    if n == "SupportRectangleRectHole":
        return {"kind": "rectrecthole", "p": [S.dimX, S.dimY, S.holeX, S.holeY, S.centerholeX, S.centerholeY]}
    raise ValueError(n)


def describe_optic(O, arrays, tag):
    n = type(O).__name__
    d = {"support": describe_support(O.support), "type": O.type,
         "centre": [float(v) for v in O.get_centre()]}
    if n == "MirrorPlane":
        d.update(kind="plane")
    elif n == "MirrorSpherical":
        d.update(kind="sphere", R=float(O.radius))
    elif n == "MirrorCylindrical":
        d.update(kind="cylinder", R=float(O.radius))
    elif n == "MirrorParabolic":
        d.update(kind="parabola", feff=float(O.feff), offaxis_rad=float(O.offaxisangle), p=float(O.p))
    elif n == "MirrorToroidal":
        d.update(kind="torus", R=float(O.majorradius), r=float(O.minorradius))
    elif n == "MirrorEllipsoidal":
        d.update(kind="ellipsoid", a=float(O.a), b=float(O.b), offaxis_rad=float(O._offaxisangle))
    elif n == "Mask":
        d.update(kind="mask")
    elif n == "DeformedMirror":
        d = describe_optic(O.Mirror, arrays, tag)
        defects = []
        for j, D in enumerate(O.DeformationList):
            dn = type(D).__name__
            if dn == "Zernike":
                defects.append({"kind": "zernike", "R": float(D.R),
                                "coeffs": [[int(k[0]), int(k[1]), float(c)] for k, c in D.coefficients.items()]})
            elif dn == "Fourrier":
                key = f"{tag}fourrier{j}_map"
                arrays[key] = np.asarray(D.deformation)
                defects.append({"kind": "fourrier", "map": key, "ctor": getattr(D, "_ctor", None)})
            else:
                raise ValueError("defect kind not captured: " + dn)
        d["defects"] = defects
    else:
        raise ValueError(n)
    return d


def bundle_arrays(rays, prefix, arrays, with_path=True):
    n = len(rays)
    arrays[prefix + "number"] = np.array([r.number if r.number is not None else -1 for r in rays], dtype=np.int64)
    arrays[prefix + "point"] = np.array([r.point for r in rays], dtype=np.float64).reshape(n, 3)
    arrays[prefix + "vector"] = np.array([r.vector for r in rays], dtype=np.float64).reshape(n, 3)
    arrays[prefix + "incidence"] = np.array(
        [np.nan if r.incidence is None else r.incidence for r in rays], dtype=np.float64)
    arrays[prefix + "intensity"] = np.array(
        [np.nan if r.intensity is None else r.intensity for r in rays], dtype=np.float64)
    if with_path:
        plen = len(rays[0].path) if n else 1
        arrays[prefix + "path"] = np.array([r.path for r in rays], dtype=np.float64).reshape(n, plen)


def dump_chain(name, chain, detector_distance=None, ignore_defects=None, extra=None, detector=None):
    """Trace `chain` with the reference and store inputs, poses and every intermediate bundle."""
    arrays = {}
    scene = {"name": name, "description": chain.description, "elements": []}
    bundle_arrays(chain.source_rays, "src_", arrays)
    wl = chain.source_rays[0].wavelength
    scene["wavelength"] = None if wl is None else float(wl)
    for k, oe in enumerate(chain.optical_elements):
        e = describe_optic(oe.type, arrays, f"el{k}_")
        e["position"] = [float(v) for v in oe.position]
        e["normal"] = [float(v) for v in oe.normal]
        e["majoraxis"] = [float(v) for v in oe.majoraxis]
        scene["elements"].append(e)
    kw = {}
    if ignore_defects is not None:
        kw["IgnoreDefects"] = ignore_defects
        scene["IgnoreDefects"] = bool(ignore_defects)
    t0 = time.perf_counter()
    out = chain.get_output_rays(**kw)
    scene["reference_trace_seconds"] = time.perf_counter() - t0
    scene["n_source"] = len(chain.source_rays)
    scene["n_out"] = [len(o) for o in out]
    for k, o in enumerate(out):
        bundle_arrays(o, f"out{k}_", arrays)
    last = out[-1]
    if (detector_distance is not None or detector is not None) and len(last) > 0:
        if detector is None:
            det = mdet.Detector(chain.optical_elements[-1].position)
            det.autoplace(last, detector_distance)
        else:
            det = detector
        scene["detector"] = {"centre": [float(v) for v in det.centre], "normal": [float(v) for v in det.normal],
                             "refpoint": [float(v) for v in det.refpoint], "distance": float(det.get_distance())}
        arrays["det_points3d"] = np.array(det.get_PointList3D(last))
        arrays["det_points2d"] = np.array(det.get_PointList2D(last))
        arrays["det_points2dcentre"] = np.array(det.get_PointList2DCentre(last))
        arrays["det_delays"] = np.array(det.get_Delays(last))
        sd_spot, sd_dur = mplots.GetResultSummary(det, last, False)
        scene["SpotSizeSD"] = float(sd_spot)
        scene["DurationSD"] = float(sd_dur)
        scene["ETransmission"] = float(mplots.getETransmission(chain.source_rays, last))
    if extra:
        scene.update(extra)
    if getattr(chain, "_placement", None) is not None and not getattr(chain, "_modified_after_placement", False):
        scene["placement"] = chain._placement
    arrays["scene_json"] = np.array(json.dumps(scene))
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: src {scene['n_source']} -> {scene['n_out']}  ({scene['reference_trace_seconds']:.2f}s)  "
          f"{os.path.getsize(path)/1024:.0f} kB", flush=True)
    return scene


# ----------------------------------------------------------------------------- scenes (BASELINE configs)
def scene_c1(n_rays, name):
    """examples/CONFIG_singleparabola.py:20-50 (plane wave on a 90deg off-axis parabola with a hole, rolled 50 urad)."""
    SourceProperties = {"Divergence": 0, "SourceSize": 50, "Wavelength": 800e-6, "DeltaFT": 2.7, "NumberRays": n_rays}
    Support = msupp.SupportRoundHole(30, 5, 10, 5)
    Parabola = mmirror.MirrorParabolic(100, 90, Support)
    chain = mp.OEPlacement(SourceProperties, [Parabola], [200], [0.00],
                           Description="A 90deg off-axis parabola with a hole, illuminated by a plane wave.")
    chain._modified_after_placement = True   # rolled: poses no longer those of OEPlacement; keep the source only
    placement = chain._placement
    chain.optical_elements[0].rotate_roll_by(np.rad2deg(50e-6))
    return dump_chain(name, chain, detector_distance=100, extra={"placement_before_roll": placement})


def scene_c2(n_rays=1000):
    """examples/CONFIG_2toroidals_f-x-f.py:19-54 (mask + 2 toroids, second distance looped)."""
    SourceProperties = {"Divergence": 50e-3 / 2, "SourceSize": 0, "Wavelength": 50e-6, "DeltaFT": 0.5,
                        "NumberRays": n_rays}
    Mask = mmask.Mask(msupp.SupportRoundHole(20, 14e-3 * 500, 0, 0))
    Support = msupp.SupportRectangle(150, 32)
    R, r = mmirror.ReturnOptimalToroidalRadii(500, 80)
    Tor = mmirror.MirrorToroidal(R, r, Support)
    chains = mp.OEPlacement(SourceProperties, [Mask, Tor, Tor], [400, 100, np.linspace(300, 700, 11)],
                            [0, 80, -80], [0, 0, 0], "2 toroidal mirrors in f-d-f config")
    for i in (0, 5, 10):
        dump_chain(f"c2_fxf_chain{i:02d}", chains[i], detector_distance=500,
                   extra={"loop_variable_value": float(chains[i].loop_variable_value)})


def scene_c3(n_rays=1000):
    """examples/CONFIG_2toroidals_twisted.py:19-52 (mask + 2 toroids, incidence plane of the 2nd twisted)."""
    SourceProperties = {"Divergence": 50e-3 / 2, "SourceSize": 0, "Wavelength": 50e-6, "DeltaFT": 0.5,
                        "NumberRays": n_rays}
    Mask = mmask.Mask(msupp.SupportRoundHole(30, 41e-3 / 2 * 500, 0, 0))
    Support = msupp.SupportRectangle(200, 30)
    R, r = mmirror.ReturnOptimalToroidalRadii(600, 80)
    Tor = mmirror.MirrorToroidal(R, r, Support)
    chains = mp.OEPlacement(SourceProperties, [Mask, Tor, Tor], [500, 100, 600], [0, 80, -80],
                            [0, 0, np.linspace(-90, 90, 10)], "2 toroidal mirrors, twisted")
    for i in (0, 4, 9):
        dump_chain(f"c3_twisted_chain{i:02d}", chains[i], detector_distance=600,
                   extra={"loop_variable_value": float(chains[i].loop_variable_value)})


def scene_c5(n_rays=1000):
    """Geometry of examples/CONFIG_deformed.py:19-46 with a Zernike defect (ART/ModuleDefects.py:149-174)."""
    SourceProperties = {"Divergence": 0, "SourceSize": 100, "Wavelength": 800e-6, "DeltaFT": 0, "NumberRays": n_rays}
    Support = msupp.SupportRectangle(40, 40)
    Mirror = mmirror.MirrorParabolic(25.4, 0, Support)
    coeffs = {(2, 1): 1e-4, (3, 1): 5e-5, (4, 2): 2e-5, (3, 3): -3e-5, (5, 2): 1e-5, (6, 3): -4e-6, (2, 0): 2.5e-5}
    Defect = mdef.Zernike(Support, coeffs)
    Deformed = mmirror.DeformedMirror(Mirror, [Defect])
    chain = mp.OEPlacement(SourceProperties, [Deformed], [15], [0], Description="deformed parabola (Zernike)")
    dump_chain("c5_zernike_ignoredefects", chain, detector_distance=25.4, ignore_defects=True)
    chain2 = chain.copy_chain()
    chain2._placement = chain._placement
    dump_chain("c5_zernike_withdefects", chain2, detector_distance=25.4, ignore_defects=False)
    # two stacked defects (normal_add applied twice, ART/ModuleMirror.py:952-961)
    Defect2 = mdef.Zernike(Support, {(1, 1): 2e-5, (4, 0): -1e-5, (7, 3): 3e-6})
    Deformed2 = mmirror.DeformedMirror(Mirror, [Defect, Defect2])
    chain3 = mp.OEPlacement(SourceProperties, [Deformed2], [15], [0], Description="deformed parabola (2 Zernike)")
    dump_chain("c5_zernike2_withdefects", chain3, detector_distance=25.4, ignore_defects=False)


# ----------------------------------------------------------------------------- single-element scenes
def _jittered_point_source(n, half_angle, seed, origin_jitter=0.0):
    """PointSource along +x from the origin with seeded per-ray jitter on origin (so rays are not all co-punctual)."""
    rays = msource.PointSource(np.array([0.0, 0.0, 0.0]), np.array([1.0, 0.0, 0.0]), half_angle, n, Wavelength=50e-6)
    rays = msource.ApplyGaussianIntensityToRayList(rays, 1 / np.e**2)
    if origin_jitter > 0:
        rng = np.random.default_rng(seed)
        for r in rays:
            r.point = r.point + rng.uniform(-origin_jitter, origin_jitter, 3)
    return rays


def scene_single(name, optic, distance, incidence_deg, n=600, half_angle=0.05, jitter=0.0, seed=7,
                 plane_angle=0.0, detector_distance=None, yaw_deg=0.0, extra_shift=None):
    SourceProperties = {"Divergence": half_angle, "SourceSize": 0, "Wavelength": 50e-6, "DeltaFT": 0.5,
                        "NumberRays": n}
    chain = mp.OEPlacement(SourceProperties, [optic], [distance], [incidence_deg], [plane_angle], name)
    if jitter > 0:
        chain.source_rays = _jittered_point_source(n, half_angle, seed, jitter)
        chain._placement["jittered_source"] = True
    if yaw_deg:
        chain.optical_elements[0].rotate_yaw_by(yaw_deg)
        chain._modified_after_placement = True
    if extra_shift is not None:
        chain.optical_elements[0].position = chain.optical_elements[0].position + np.asarray(extra_shift, float)
    return dump_chain(name, chain, detector_distance=detector_distance)


def scenes_single_elements():
    S_round = msupp.SupportRound(12)
    S_roundhole = msupp.SupportRoundHole(14, 3, 5, -3)
    S_rect = msupp.SupportRectangle(30, 18)
    S_recthole = msupp.SupportRectangleHole(30, 20, 4, -6, 4)
    S_rectrect = msupp.SupportRectangleRectHole(30, 20, 8, 5, 7, -4)
    # plane mirror, each of the 5 supports (ART/ModuleSupport.py:68-70,151-155,228-230,322-326,431-435)
    for tag, S in (("round", S_round), ("roundhole", S_roundhole), ("rect", S_rect), ("recthole", S_recthole),
                   ("rectrecthole", S_rectrect)):
        scene_single(f"single_plane_{tag}", mmirror.MirrorPlane(S), 200, 35, n=800, half_angle=0.09, jitter=0.5,
                     detector_distance=150)
    # masks (ART/ModuleMask.py:51-61) with two supports
    scene_single("single_mask_roundhole", mmask.Mask(msupp.SupportRoundHole(15, 6, 1, 0.5)), 180, 0, n=800,
                 half_angle=0.1, jitter=0.3, detector_distance=50)
    scene_single("single_mask_rect", mmask.Mask(msupp.SupportRectangle(10, 6)), 150, 10, n=800, half_angle=0.08,
                 jitter=0.3, detector_distance=50)
    # spherical concave / convex (ART/ModuleMirror.py:142-187; CX flip ART/ModuleProcessing.py:94-95)
    scene_single("single_sphere_cc", mmirror.MirrorSpherical(400, msupp.SupportRound(20)), 300, 12, n=800,
                 half_angle=0.08, jitter=0.4, detector_distance=180)
    scene_single("single_sphere_cx", mmirror.MirrorSpherical(-500, msupp.SupportRound(20)), 250, 20, n=800,
                 half_angle=0.08, jitter=0.4, detector_distance=100)
    # cylinder concave / convex (ART/ModuleMirror.py:803-853)
    scene_single("single_cylinder_cc", mmirror.MirrorCylindrical(600, msupp.SupportRectangle(40, 30)), 300, 30,
                 n=800, half_angle=0.05, jitter=0.4, detector_distance=200, yaw_deg=15.0)
    scene_single("single_cylinder_cx", mmirror.MirrorCylindrical(-600, msupp.SupportRectangle(40, 30)), 300, 25,
                 n=600, half_angle=0.05, jitter=0.4, detector_distance=100)
    # ellipsoid (ART/ModuleMirror.py:593-714), both ctor styles
    ell = mmirror.MirrorEllipsoidal(msupp.SupportRectangle(60, 20), OffAxisAngle=150, f_object=300, f_image=500)
    scene_single("single_ellipsoid_foci", ell, 300, 75, n=800, half_angle=0.03, jitter=0.0, detector_distance=500)
    a, b = mmirror.ReturnOptimalEllipsoidalAxes(400, 70)
    ell2 = mmirror.MirrorEllipsoidal(msupp.SupportRound(25), SemiMajorAxis=a, SemiMinorAxis=b)
    scene_single("single_ellipsoid_axes", ell2, 400, 70, n=800, half_angle=0.02, jitter=0.2, detector_distance=400)
    # off-axis parabolas at several angles, point source at the focus (collimating), and a twisted plane
    scene_single("single_parabola_oap30", mmirror.MirrorParabolic(150, 30, msupp.SupportRound(15)), 150, 0, n=800,
                 half_angle=0.08, jitter=0.0, detector_distance=300)
    scene_single("single_parabola_oap120_jit", mmirror.MirrorParabolic(80, 120, msupp.SupportRoundHole(20, 4, 8, 0)),
                 80, 0, n=800, half_angle=0.2, jitter=1.0, detector_distance=120, plane_angle=37.0)
    # toroid with a jittered, over-filling bundle (misses on the aperture; rays from outside the tube)
    R, r = mmirror.ReturnOptimalToroidalRadii(300, 75)
    scene_single("single_torus_overfill", mmirror.MirrorToroidal(R, r, msupp.SupportRectangle(60, 14)), 300, 75,
                 n=1000, half_angle=0.04, jitter=1.5, detector_distance=300)
    R, r = mmirror.ReturnOptimalToroidalRadii(50, 20)
    scene_single("single_torus_steep", mmirror.MirrorToroidal(R, r, msupp.SupportRound(30)), 100, 20,
                 n=1000, half_angle=0.25, jitter=2.0, detector_distance=100, plane_angle=-60.0)


def scene_fourrier(n_rays=1500):
    """examples/CONFIG_deformed.py:19-46 with a coarser Fourrier map (smallest = 1 mm -> 80 x 80) and a seeded RNG:
    the only unseeded call of the ctor is one np.random.uniform (ART/ModuleDefects.py:92)."""
    SourceProperties = {"Divergence": 0, "SourceSize": 100, "Wavelength": 800e-6, "DeltaFT": 0, "NumberRays": n_rays}
    Support = msupp.SupportRectangle(40, 40)
    Mirror = mmirror.MirrorParabolic(25.4, 0, Support)
    for tag, ctor in (("a", dict(RMS=1e-1, smallest=1.0, seed=2024)),
                      ("b", dict(RMS=2e-2, slope=-1.5, smallest=0.5, biggest=25.0, seed=99))):
        kw = {k: v for k, v in ctor.items() if k != "seed"}
        np.random.seed(ctor["seed"])
        Defect = mdef.Fourrier(Support, **kw)
        Defect._ctor = ctor
        Deformed = mmirror.DeformedMirror(Mirror, [Defect])
        chain = mp.OEPlacement(SourceProperties, [Deformed], [15], [0], Description="deformed parabola (Fourrier)")
        dump_chain("c5_fourrier_" + tag, chain, detector_distance=25.4, ignore_defects=True)
    # Fourrier + Zernike stacked on one mirror
    np.random.seed(5)
    D1 = mdef.Fourrier(Support, RMS=5e-2, smallest=2.0)
    D1._ctor = dict(RMS=5e-2, smallest=2.0, seed=5)
    D2 = mdef.Zernike(Support, {(2, 1): 1e-4, (4, 2): 2e-5})
    chain = mp.OEPlacement(SourceProperties, [mmirror.DeformedMirror(Mirror, [D1, D2])], [15], [0], Description="mixed defects")
    dump_chain("c5_fourrier_zernike", chain, detector_distance=25.4, ignore_defects=True)


def scene_mixed8(n_rays=1000):
    """8-element mixed chain (BASELINE config 4 analogue): OAP collimate -> plane -> toroid pair -> planes -> OAP focus."""
    SourceProperties = {"Divergence": 0.03, "SourceSize": 0, "Wavelength": 50e-6, "DeltaFT": 0.5,
                        "NumberRays": n_rays}
    oap = mmirror.MirrorParabolic(200, 60, msupp.SupportRound(20))
    plane = mmirror.MirrorPlane(msupp.SupportRound(30))
    R, r = mmirror.ReturnOptimalToroidalRadii(400, 78)
    tor = mmirror.MirrorToroidal(R, r, msupp.SupportRectangle(180, 30))
    oap2 = mmirror.MirrorParabolic(150, 45, msupp.SupportRound(25))
    optics = [oap, plane, tor, tor, plane, plane, oap2, plane]
    dist = [200, 150, 250, 800, 650, 120, 140, 60]
    inc = [0, 45, 78, -78, 30, -30, 0, 20]
    plane_angles = [0, 0, 0, 0, 90, 0, 0, 45]
    chain = mp.OEPlacement(SourceProperties, optics, dist, inc, plane_angles, "8-element mixed chain")
    dump_chain("c4_mixed8", chain, detector_distance=90)


def scene_frames():
    """Degenerate frame cases of RotationPoint (ART/ModuleGeometry.py:333-343): normal parallel / antiparallel to ez."""
    rng = np.random.default_rng(11)
    rays = []
    for k in range(300):
        p = np.array([rng.uniform(-5, 5), rng.uniform(-5, 5), rng.uniform(40, 60)]) * (1 if k % 2 else -1)
        v = np.array([rng.normal(0, 0.05), rng.normal(0, 0.05), -1.0]) * (1 if k % 2 else -1)
        rays.append(mray.Ray(p, v, Number=k, Wavelength=50e-6, Intensity=1.0))
    for tag, normal, major in (("par", [0, 0, 1.0], [1.0, 0, 0]), ("anti", [0, 0, -1.0], [1.0, 0, 0]),
                               ("anti_majneg", [0, 0, -1.0], [-1.0, 0, 0]), ("par_majy", [0, 0, 1.0], [0, 1.0, 0]),
                               ("x_majz", [1.0, 0, 0], [0, 0, 1.0])):
        optic = mmirror.MirrorSpherical(300, msupp.SupportRound(40))
        oe = moe.OpticalElement(optic, np.array([0.3, -0.2, 1.0]), np.array(normal), np.array(major))
        rr = rays
        if tag == "x_majz":
            rr = [mray.Ray(np.array([r.point[2], r.point[1], r.point[0]]),
                           np.array([r.vector[2], r.vector[1], r.vector[0]]), Number=r.number, Intensity=1.0)
                  for r in rays]
        chain = moc.OpticalChain(rr, [oe], "degenerate frame " + tag)
        dump_chain("frame_" + tag, chain, detector_distance=None)


def scene_zernike_tierA():
    """Tier A: reference modules that import without any stand-in."""
    import importlib.util
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, 64)
    y = rng.uniform(-1, 1, 64)
    spec = importlib.util.spec_from_file_location("rzg", os.path.join(REF, "ART", "recursive_zernike_generator.py"))
    rzg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rzg)
    max_order = 9
    val, gx, gy = rzg.zernike_gradient(list(x), list(y), max_order)
    nm = [(n, m) for n in range(max_order + 1) for m in range(n + 1)]
    arrays = {"x": x, "y": y, "nm": np.array(nm, dtype=np.int64),
              "val": np.array([val[k][0][1] for k in nm], dtype=np.float64),
              "gx": np.array([gx[k][0][1] for k in nm], dtype=np.float64),
              "gy": np.array([gy[k][0][1] for k in nm], dtype=np.float64)}
    # Zernike.get_normal / get_offset (ART/ModuleDefects.py:156-174) on a rectangular support
    S = msupp.SupportRectangle(40, 30)
    coeffs = {(2, 1): 1e-4, (3, 0): -2e-5, (5, 4): 7e-6, (6, 6): 1e-6, (8, 3): -2e-6}
    Z = mdef.Zernike(S, coeffs)
    pts = np.stack([rng.uniform(-20, 20, 64), rng.uniform(-15, 15, 64), rng.uniform(-1, 1, 64)], axis=1)
    arrays["defect_points"] = pts
    arrays["defect_normal"] = np.array([Z.get_normal(p) for p in pts])
    arrays["defect_offset"] = np.array([Z.get_offset(p) for p in pts])
    arrays["defect_R"] = np.array(Z.R)
    arrays["defect_coeffs"] = np.array([[k[0], k[1], c] for k, c in coeffs.items()])
    np.savez_compressed(os.path.join(OUT, "zernike_tierA.npz"), **arrays)
    print("zernike_tierA: %d polys x %d points" % (len(nm), len(x)))


def scene_geometry_units():
    """Unit-level vectors for the geometry helpers (ART/ModuleGeometry.py)."""
    rng = np.random.default_rng(3)
    arrays = {}
    # np.roots plumbing: SolverQuadratic / SolverQuartic (:80-106) on well-separated and awkward inputs
    quads = [[1.0, -3.0, 2.0], [2.5e-9, -0.2, 21.25], [0.0, -0.2, 21.25], [1e-34, 2.0, -3.0], [1.0, 2.0, 5.0],
             [1.0, -2.0, 1.0], [1.0, 0.0, -4.0]]
    arrays["quad_in"] = np.array(quads)
    qo = np.full((len(quads), 2), np.nan)
    for i, q in enumerate(quads):
        s = mgeo.SolverQuadratic(*q)
        qo[i, :len(s)] = s
    arrays["quad_out"] = qo
    # Kahan angle (:40-44)
    U = rng.normal(size=(50, 3))
    V = rng.normal(size=(50, 3))
    V[:5] = U[:5] * 1.0000001 + 1e-9 * rng.normal(size=(5, 3))
    V[5:8] = -U[5:8]
    arrays["angle_U"], arrays["angle_V"] = U, V
    arrays["angle_out"] = np.array([mgeo.AngleBetweenTwoVectors(u, v) for u, v in zip(U, V)])
    # RotationPoint incl. degenerate cases (:333-343)
    A1 = rng.normal(size=(20, 3))
    A2 = rng.normal(size=(20, 3))
    A2[0] = A1[0] * 2.0
    A2[1] = -A1[1] * 0.5
    P = rng.normal(size=(20, 3)) * 10
    arrays["rot_A1"], arrays["rot_A2"], arrays["rot_P"] = A1, A2, P
    arrays["rot_out"] = np.array([mgeo.RotationPoint(p, a, b) for p, a, b in zip(P, A1, A2)])
    # RotationAroundAxis (:321-329)
    ax = rng.normal(size=(20, 3))
    ang = rng.uniform(-4, 4, 20)
    arrays["raa_axis"], arrays["raa_angle"] = ax, ang
    arrays["raa_out"] = np.array([mgeo.RotationAroundAxis(a, t, p) for a, t, p in zip(ax, ang, P)])
    # SpiralVogel (:61-76) and the sources (ART/ModuleSource.py)
    arrays["vogel_7_2p5"] = mgeo.SpiralVogel(7, 2.5)
    arrays["vogel_1000_1"] = mgeo.SpiralVogel(1000, 1.0)
    for tag, rays in (("pointsource", msource.PointSource(np.array([1.0, 2.0, 3.0]), np.array([0.3, -0.2, 0.9]), 0.05, 50)),
                      ("planewave", msource.PlaneWaveDisk(np.array([1.0, 2.0, 3.0]), np.array([0.0, 1.0, 0.2]), 12.0, 50)),
                      ("extended", msource.ExtendedSource(np.array([0.0, 0.0, 0.0]), np.array([1.0, 0.0, 0.0]), 0.1, 0.02, 9000))):
        rays = msource.ApplyGaussianIntensityToRayList(rays, 1 / np.e**2)
        bundle_arrays(rays, f"src_{tag}_", arrays, with_path=False)
    # normal_add (:394-407)
    N1 = rng.normal(size=(10, 3)); N1[:, 2] = np.abs(N1[:, 2]) + 1
    N2 = rng.normal(size=(10, 3)); N2[:, 2] = np.abs(N2[:, 2]) + 1
    arrays["nadd_1"], arrays["nadd_2"] = N1, N2
    arrays["nadd_out"] = np.array([mgeo.normal_add(a, b) for a, b in zip(N1, N2)])
    # OpticalElement misalignment helpers (ART/ModuleOpticalElement.py:169-250)
    oe = moe.OpticalElement(mmirror.MirrorPlane(msupp.SupportRound(5)), np.array([1.0, 2.0, 3.0]),
                            np.array([0.2, -0.4, 0.7]), np.cross(np.array([0.2, -0.4, 0.7]), np.array([0.0, 0.0, 1.0])))
    seq = []
    for op, val in (("rotate_pitch_by", 1.5), ("rotate_roll_by", -0.7), ("rotate_yaw_by", 12.0),
                    ("shift_along_normal", 0.3), ("shift_along_major", -0.2), ("shift_along_cross", 0.9)):
        getattr(oe, op)(val)
        seq.append(np.concatenate([oe.position, oe.normal, oe.majoraxis]))
    arrays["oe_seq"] = np.array(seq)
    # statistics helpers (ART/ModuleProcessing.py:485-532)
    pts = [np.array(p) for p in rng.normal(size=(40, 2))]
    w = list(rng.uniform(0.1, 1, 40))
    dl = [float(v) for v in rng.normal(size=40)]
    arrays["stat_pts"], arrays["stat_w"], arrays["stat_delays"] = np.array(pts), np.array(w), np.array(dl)
    arrays["stat_out"] = np.array([mp.StandardDeviation(pts), mp.WeightedStandardDeviation(pts, w),
                                   mp.StandardDeviation(dl), mp.WeightedStandardDeviation(dl, w)])
    np.savez_compressed(os.path.join(OUT, "geometry_units.npz"), **arrays)
    print("geometry_units done")


def scene_autofocus():
    """FindOptimalDistance on the C3 chain 4 bundle (ART/ModuleProcessing.py:317-460), all rays, unweighted+weighted."""
    SourceProperties = {"Divergence": 50e-3 / 2, "SourceSize": 0, "Wavelength": 50e-6, "DeltaFT": 0.5,
                        "NumberRays": 400}
    Mask = mmask.Mask(msupp.SupportRoundHole(30, 41e-3 / 2 * 500, 0, 0))
    R, r = mmirror.ReturnOptimalToroidalRadii(600, 80)
    Tor = mmirror.MirrorToroidal(R, r, msupp.SupportRectangle(200, 30))
    chain = mp.OEPlacement(SourceProperties, [Mask, Tor, Tor], [500, 100, 600], [0, 80, -80], [0, 0, 30.0], "autofocus")
    out = chain.get_output_rays()[-1]
    det = mdet.Detector(chain.optical_elements[-1].position)
    det.autoplace(out, 600)
    res = {}
    for optfor in ("intensity", "duration"):
        for weighted in (False, True):
            d, s, t = mp.FindOptimalDistance(det, out, optfor, None, 3, weighted, False)
            res[f"{optfor}_{int(weighted)}"] = [float(d.get_distance()), float(s), float(t)]
    na = mp.ReturnNumericalAperture(out, 1)
    dump_chain("autofocus_c3", chain, detector=det,
               extra={"autofocus": res, "NA": float(na), "Airy": float(mp.ReturnAiryRadius(50e-6, na))})


def check_standin_rotations():
    """Spot check before anything is generated: the reference's RotationAroundAxis (ART/ModuleGeometry.py:321-329),
    running on the stand-in quaternion class, against SciPy (the full comparison: tests/test_quaternion_standin.py)."""
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(7)
    for ang in list(rng.uniform(-np.pi, np.pi, 200)) + [0.0, 1e-12, np.pi, np.pi - 1e-12]:
        axis, v = rng.normal(size=3), rng.normal(size=3)
        got = mgeo.RotationAroundAxis(axis, ang, v)
        ref = Rotation.from_rotvec(ang * axis / np.linalg.norm(axis)).apply(v)
        assert np.abs(got - ref).max() <= 6 * np.finfo(float).eps * (1 + abs(ang)) * np.linalg.norm(v), (ang, got, ref)


if __name__ == "__main__":
    check_standin_rotations()
    np.random.seed(12345)
    t0 = time.perf_counter()
    scene_zernike_tierA()
    scene_geometry_units()
    scene_c1(1000, "c1_singleparabola")
    scene_c1(10000, "c1_singleparabola_1e4")     # BASELINE.json config 1 at its own size (N = 1000 and 1e4)
    scene_c2()
    scene_c3()
    scene_c5()
    scene_fourrier()
    scenes_single_elements()
    scene_mixed8()
    scene_frames()
    scene_autofocus()
    print("all goldens written in %.1fs" % (time.perf_counter() - t0))
