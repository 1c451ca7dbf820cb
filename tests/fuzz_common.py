"""Differential fuzzing: random single-element scenes traced by the product (CPU twin or GPU) and by the pinned CPU
oracle.  The golden fixtures cover the shipped configurations; this covers the parameter space between them: every
optic kind, every aperture kind, arbitrary poses, sources inside and outside the optic's body, footprints that
overfill the aperture.  Tolerances are those of the parity tests (survivors exact, 1e-10 relative).

Scene dictionaries use the schema of the golden fixtures, so both `oracle.elements_from_scene` and
`parity_common.build_elements` accept them."""
import numpy as np

from oracle import art_oracle as orc
import parity_common as pc
from attosecondraytracing_amd.bundle import RayBundle

KINDS = ["plane", "sphere_cc", "sphere_cx", "cylinder_cc", "cylinder_cx", "parabola", "torus", "torus_steep",
         "ellipsoid", "mask"]
TYPE = {"plane": "Plane Mirror", "sphere_cc": "SphericalCC Mirror", "sphere_cx": "SphericalCX Mirror",
        "cylinder_cc": "CylindricalCC Mirror", "cylinder_cx": "CylindricalCX Mirror", "parabola": "Parabolic Mirror",
        "torus": "Toroidal Mirror", "torus_steep": "Toroidal Mirror", "ellipsoid": "Ellipsoidal Mirror",
        "mask": "Mask"}


def random_support(rng, size):
    k = rng.choice(["round", "roundhole", "rect", "recthole", "rectrecthole"])
    if k == "round":
        return {"kind": k, "p": [size]}
    if k == "roundhole":
        return {"kind": k, "p": [size, size * rng.uniform(0.1, 0.4), size * rng.uniform(-0.3, 0.3),
                                 size * rng.uniform(-0.3, 0.3)]}
    X, Y = 2 * size, 2 * size * rng.uniform(0.3, 1.0)
    if k == "rect":
        return {"kind": k, "p": [X, Y]}
    if k == "recthole":
        return {"kind": k, "p": [X, Y, 0.2 * Y * rng.uniform(0.3, 1.0), X * rng.uniform(-0.2, 0.2),
                                 Y * rng.uniform(-0.2, 0.2)]}
    return {"kind": k, "p": [X, Y, X * rng.uniform(0.1, 0.4), Y * rng.uniform(0.1, 0.4), X * rng.uniform(-0.2, 0.2),
                             Y * rng.uniform(-0.2, 0.2)]}


def random_optic(rng, kind):
    """Scene-dictionary entry of one optic of the given kind with random parameters and aperture (no pose yet)."""
    size = float(rng.uniform(8.0, 40.0))
    e = {"kind": kind.split("_")[0], "type": TYPE[kind], "support": random_support(rng, size)}
    if kind.startswith("sphere") or kind.startswith("cylinder"):
        e["R"] = float(rng.uniform(4 * size, 3000.0))
    elif kind == "parabola":
        e["feff"] = float(rng.uniform(3 * size, 600.0))
        e["offaxis_rad"] = float(np.deg2rad(rng.uniform(5.0, 120.0)))
        e["p"] = e["feff"] * (1 + np.cos(e["offaxis_rad"]))
    elif kind == "torus":
        e["r"] = float(rng.uniform(3 * size, 500.0))
        e["R"] = float(rng.uniform(1.2 * e["r"], 9000.0))
    elif kind == "torus_steep":      # minor radius larger than the major one: the quartic's second factor has roots
        e["R"] = float(rng.uniform(3 * size, 300.0))
        e["r"] = float(e["R"] * rng.uniform(1.2, 3.0))
    elif kind == "ellipsoid":
        while True:   # some (a, b, angle) have no surface point under that angle: the reference's centre is NaN
            e["a"] = float(rng.uniform(300.0, 900.0))
            e["b"] = float(e["a"] * rng.uniform(0.35, 0.9))
            e["offaxis_rad"] = float(np.deg2rad(rng.uniform(40.0, 130.0)))
            with np.errstate(invalid="ignore"):
                if np.isfinite(orc.ellipsoid_centre(e["a"], e["b"], e["offaxis_rad"])).all():
                    break
    O = orc.optic_from_desc(e)
    e["centre"] = [float(v) for v in O.centre()]
    return e, size, O


def random_pose(rng, e, pos, towards, theta):
    """Orient optic `e` at `pos` so that the unit vector `towards` (pointing from the optic to where the light comes
    from) makes the angle theta with its normal; the roll about the normal is random."""
    t = rng.normal(size=3)
    t -= np.dot(t, towards) * towards
    t /= np.linalg.norm(t)
    normal = np.cos(theta) * towards + np.sin(theta) * t
    major = np.cross(normal, rng.normal(size=3))
    major /= np.linalg.norm(major)
    e["position"], e["normal"], e["majoraxis"] = pos.tolist(), normal.tolist(), major.tolist()


def random_scene(seed, n_rays=1500):
    """One to four optics of random kinds in random poses + a bundle of rays aimed at the first; every further optic
    sits on the reflected / transmitted chief ray of the chain so far.  Returns (scene, arrays)."""
    rng = np.random.default_rng(seed)
    kind = KINDS[seed % len(KINDS)]
    e, size, O = random_optic(rng, kind)
    deformed = (seed // len(KINDS)) % 3 == 0 and e["kind"] != "mask"
    gridded = (seed // len(KINDS)) % 9 == 4 and e["kind"] != "mask"
    arrays_extra = {}
    if gridded:    # DeformedMirror with a Fourrier height map (+ sometimes a Zernike term), offsets only
        import ART.ModuleDefects as mdef
        ctor = {"RMS": float(rng.uniform(1e-5, 5e-4)), "smallest": float(rng.uniform(1.5, 4.0)), "seed": int(seed)}
        np.random.seed(ctor["seed"])
        kw = {k: v for k, v in ctor.items() if k != "seed"}
        arrays_extra["el0_map"] = mdef.Fourrier(pc.build_support(e["support"]), **kw).deformation
        e["defects"] = [{"kind": "fourrier", "map": "el0_map", "ctor": ctor}]
        if rng.uniform() < 0.5:
            e["defects"].append({"kind": "zernike", "coeffs": [[3, 1, float(rng.uniform(-1, 1) * 1e-4)]],
                                 "R": O.support.circum_circ()})
    if deformed:   # DeformedMirror with 1-2 Zernike defects, traced with IgnoreDefects=False
        e["defects"] = []
        for _ in range(int(rng.integers(1, 3))):
            coeffs = []
            for _ in range(int(rng.integers(1, 6))):
                n = int(rng.integers(1, 9))
                coeffs.append([n, int(rng.integers(0, n + 1)), float(rng.uniform(-1.0, 1.0) * 2e-4)])
            e["defects"].append({"kind": "zernike", "coeffs": coeffs, "R": O.support.circum_circ()})
    # source: distance d from the element's position, chief ray at `theta` from the normal, cone wide enough to
    # overfill the aperture in part of the trials
    pos = rng.uniform(-500.0, 500.0, 3)
    w = rng.normal(size=3)
    w /= np.linalg.norm(w)
    theta = np.deg2rad(rng.uniform(0.0, 80.0))
    random_pose(rng, e, pos, w, theta)
    d = float(rng.uniform(60.0, 1200.0))
    S = pos + d * w
    div = float(rng.uniform(0.3, 1.6) * size * max(np.cos(theta), 0.25) / d)
    if (seed // 7) % 4 == 3:
        # collimated beam (parallel rays: the degenerate leading coefficient of the parabola's quadratic, grazing
        # cylinders, ...), wide enough to overfill the aperture in part of the trials; PlaneWaveDisk emits n - 1 rays
        B = orc.plane_wave_disk(S, -w, float(rng.uniform(0.3, 1.6) * size), n_rays + 1)
    else:
        B = orc.point_source(S, -w, div, n_rays)
        # jitter the origins so that they are not all one point (exercises per-ray origins in the transforms)
        B.point = B.point + rng.normal(scale=0.05 * size, size=B.point.shape)
    elements = [e]
    # further optics on the chief ray (0-3 of them, as long as the chief ray survives): each sits on the ray leaving
    # the previous one, under a random incidence
    chief = orc.make_bundle(S[None, :], -w[None, :], np.array([0]), np.array([np.nan]), None)
    for _ in range((seed // (3 * len(KINDS))) % 4):
        after = orc.ray_tracing_calculation(chief, orc.elements_from_scene({"elements": elements}, arrays_extra),
                                            IgnoreDefects=True)[-1]
        if len(after) != 1:
            break
        nxt, _, _ = random_optic(rng, KINDS[int(rng.integers(0, len(KINDS)))])
        pos_n = after.point[0] + float(rng.uniform(50.0, 800.0)) * after.vector[0]
        random_pose(rng, nxt, pos_n, -after.vector[0], np.deg2rad(rng.uniform(0.0, 75.0)))
        elements.append(nxt)
    scene = {"elements": elements, "n_source": n_rays, "IgnoreDefects": not deformed}
    arrays = {"src_point": B.point, "src_vector": B.vector, "src_number": B.number,
              "src_intensity": np.full(n_rays, np.nan), **arrays_extra}
    return scene, arrays


GRAZING = 1.5   # rad (86 deg): beyond it the hit point is ill-conditioned (error amplified by 1/cos(incidence))


def pose_noise(e):
    """Rounding noise (rad) the REFERENCE's frame construction carries for this pose.  Its two RotationPoint steps
    (normal -> ez, then the rotated major axis -> ex; ART/ModuleProcessing.py:284-295, ART/ModuleGeometry.py:333-343)
    rotate about Axis1 x Axis2; when the two are within an angle d of antiparallel (but outside the 1e-10 special case)
    that cross product is d long and carries ~1e-16 of rounding, i.e. the axis of a half-turn is uncertain by 1e-16/d
    and so is everything behind it, twice.  Any two evaluations of the same formulas (the reference with
    numpy-quaternion, the oracle, the package's frame_maps) differ by that much: seed 40030221 has d = 5.2e-7 and the
    oracle and the kernels 5e-10 apart in every direction, with both builds of the kernels."""
    n, m = np.asarray(e["normal"], float), np.asarray(e["majoraxis"], float)
    ez, ex = np.array([0.0, 0.0, 1.0]), np.array([1.0, 0.0, 0.0])
    m1 = orc.rotation_point(m[None, :], n, ez)[0]
    noise = 0.0
    for u, v in ((n, ez), (m1, ex)):
        d = np.linalg.norm(np.cross(u, v)) / (np.linalg.norm(u) * np.linalg.norm(v))     # sin of the angle between them
        if np.dot(u, v) < 0.0 and d > 1e-10:
            noise += 8e-16 / d
    return noise


# Local accuracy of ONE element of the product against the long-double truth (tests/truth_common.py), each element judged
# on the product's OWN incoming rays: relative to the scene scale for positions and segment lengths, absolute for
# directions (unit vectors) and incidence angles (rad).  The bars do NOT grow along a chain -- a 1e-11-level regression
# of any intersector, normal or reflection fails here whatever the conditioning of the scene.
# Measured over 6 000 random scenes on the CPU twin and 20 000 on the GPU: pos <= 1.1e-12 (a toroid with r = 37 mm under
# grazing incidence in a 1 300-mm scene: 1.5e-9 mm, the torus solver's stopping criterion), segment <= 8e-13.  Direction
# and incidence follow the hit point: an error dP turns the normal of a surface with curvature radius rc by dP / rc and
# the reflected direction by twice that, so their bar is 5e-12 + 4 dP / rc with the MEASURED dP of the same element
# (local_dir_tol below): a direction error without a hit-point error behind it fails at 5e-12.
#
# FROZEN (round 4): these bars -- LOCAL_TOL 3e-12 / 5e-12 (+ 4 dP / rc for directions and incidences), STRICT_TOL 1e-10
# with truth adjudication -- are CONTRACT, not tuning parameters.  They followed a measurement four times in rounds 1-3
# (last: seed 3016796, 1.115e-12 on a toroid, 1e-12 -> 3e-12); from here on an exceedance is a FINDING to be explained and
# fixed in the kernels (or recorded as a named, explained regression case below), never a reason to move a bar.
# tests/test_fuzz_differential.py::test_fuzz_bars_are_frozen pins the numbers.
LOCAL_TOL = {"pos": 3e-12, "dir": 5e-12, "seg": 3e-12, "inc": 5e-12}
# Seeds that moved a bar or needed the truth to adjudicate in earlier rounds: part of every differential run from now on.
#   3016796   toroid r = 37 mm under grazing incidence in a 1300-mm scene: local position error 1.115e-12 (the torus
#             solver's stopping criterion), the case behind the 3e-12 bar
#   60039358  near-antiparallel frame axes: oracle and kernels 5e-10 apart in every direction, adjudicated by the truth
#   65        product vs oracle beyond 1e-10 on an ill-conditioned chain, product within the bar of the truth
#   20797, 23917, 40030221   earlier adjudicated seeds (round 2)
REGRESSION_SEEDS = (3016796, 60039358, 65, 20797, 23917, 40030221)


def curvature_radius(e):
    """Smallest curvature radius (mm) of an optic's surface, None for flat ones."""
    return {"sphere": e.get("R"), "cylinder": e.get("R"), "torus": e.get("r"), "parabola": e.get("p"),
            "ellipsoid": (e.get("b", 0) ** 2 / e["a"]) if "a" in e else None}.get(e["kind"])


def local_dir_tol(e, loc, scale):
    rc = curvature_radius(e)
    return LOCAL_TOL["dir"] + (4.0 * max(loc["pos"], loc["seg"]) * scale / rc if rc else 0.0)
STRICT_TOL = {"pos": pc.REL_TOL, "dir": pc.REL_TOL, "path": pc.REL_TOL, "inc": 1e-9}    # product vs oracle, no allowances


def _local_truth_errors(E, prev, out, scale, ignore_defects, well):
    """Worst errors of the product's bundle `out` (after element E) against the truth computed from its bundle `prev`."""
    import truth_common as T
    num = out.numbers()
    if len(num) == 0:
        return None
    sel = np.searchsorted(prev["number"], num)
    seg = out.path_segments()[:, -1]
    P, v, t, inc = T.element_truth(E, prev["point"][sel], prev["vector"][sel], seg, ignore_defects)
    m = well[num]
    if not m.any():
        return None
    return {"pos": float(np.abs(out.points() - P)[m].max() / scale), "dir": float(np.abs(out.vectors() - v)[m].max()),
            "seg": float(np.abs(seg - t)[m].max() / scale), "inc": float(np.abs(out.incidences() - inc)[m].max())}


def run_differential(seeds, modes=("chain", "element"), n_rays=1500, stats=None):
    """Trace every seeded scene with the active product backend and with the oracle.

    1. Survivor indices must agree for every ray after every element.
    2. Every element of the product is checked against the long-double truth on the product's own incoming rays
       (LOCAL_TOL): its intersection, normal, reflection, path and incidence are accurate to ~1e-12 whatever the scene.
    3. Product vs oracle on every ray that met no optic above GRAZING incidence (for tangential rays both answers lie on
       the surface to 1e-15 but apart along the ray): within STRICT_TOL -- 1e-10 relative, no curvature or chain-length
       allowance -- or else ADJUDICATED BY TRUTH: the rays are traced through the whole chain in long double from the
       source, and the product must be at least as close to that truth as the oracle (the reference's algorithm) is,
       or within the bar itself.  Ill-conditioned chains (a direction difference d arriving at a curved optic after a
       flight of L leaves as d (1 + 2 L / rc)) amplify BOTH implementations' rounding; what is asserted is that the
       product's is not the larger one.
    `stats` (optional dict) receives counters: scenes, adjudicated, the worst local errors, the worst adjudicated pair.
    Returns the worst product-vs-oracle differences and how many scenes produced hits on their last element."""
    import ART.ModuleProcessing as mp
    import truth_common as T
    worst = {"pos": 0.0, "dir": 0.0, "path": 0.0, "inc": 0.0}
    stats = {} if stats is None else stats
    stats.setdefault("scenes", 0)
    stats.setdefault("adjudicated", 0)
    stats.setdefault("adjudicated_seeds", [])
    stats.setdefault("local_worst", {k: 0.0 for k in LOCAL_TOL})
    stats.setdefault("adjudicated_worst", {"product": 0.0, "oracle": 0.0})
    hits = 0
    for seed in seeds:
        scene, a = random_scene(seed, n_rays)
        stats["scenes"] += 1
        tag = f"seed {seed} (" + " -> ".join(
            f"{e['type']}{' + Zernike' if e.get('defects') else ''} [{e['support']['kind']}]" for e in scene["elements"]) + ")"
        src_o = orc.make_bundle(a["src_point"], a["src_vector"], a["src_number"], a["src_intensity"], None)
        els_o = orc.elements_from_scene(scene, a)
        refs = orc.ray_tracing_calculation(src_o, els_o, IgnoreDefects=scene["IgnoreDefects"])
        hits += int(len(refs[-1]) > 0)
        scale = pc.scene_scale(a, scene)
        els = pc.build_elements(scene, a)
        src = RayBundle.from_arrays(a["src_point"], a["src_vector"], a["src_number"], None, None)
        noise = [pose_noise(e) for e in scene["elements"]]      # nearly antiparallel frame axes: the reference's own noise
        for mode in modes:
            outs = mp.RayTracingCalculation(src, els, IgnoreDefects=scene["IgnoreDefects"], mode=mode)
            well = np.ones(len(a["src_number"]), dtype=bool)      # per source ray: no grazing hit so far
            prev = {"number": np.asarray(a["src_number"]), "point": a["src_point"], "vector": src.vectors()}
            frame_noise = 0.0
            for k, (out, ref, e) in enumerate(zip(outs, refs, scene["elements"])):
                assert np.array_equal(out.numbers(), ref.number), f"{tag}, mode {mode}: survivors differ after {k}"
                well[ref.number[ref.incidence >= GRAZING]] = False
                frame_noise += noise[k]
                # -- 2. local truth (grazing hits of THIS element excluded as well: conditioning of the hit itself)
                if T.HAVE_LD:
                    loc = _local_truth_errors(els_o[k], prev, out, scale, scene["IgnoreDefects"], well)
                    if loc is not None:
                        for key, v in loc.items():
                            lim = (local_dir_tol(e, loc, scale) if key in ("dir", "inc") else LOCAL_TOL[key]) + noise[k]
                            assert v <= lim, (f"{tag}, mode {mode}, element {k}: LOCAL {key} error {v:.3e} > {lim:.1e} "
                                              f"against the long-double truth")
                            stats["local_worst"][key] = max(stats["local_worst"][key], v)
                prev = {"number": out.numbers(), "point": out.points(), "vector": out.vectors()}
                # -- 3. product vs oracle
                m = well[ref.number]
                if not m.any():
                    continue
                mean_path = max(float(np.mean(np.sum(ref.path[m], axis=1))), 1.0)
                err = {"pos": np.abs(out.points() - ref.point)[m].max() / scale,
                       "dir": np.abs(out.vectors() - ref.vector)[m].max(),
                       "path": max(np.abs(out.paths_total() - np.sum(ref.path, axis=1))[m].max(),
                                   np.abs(out.path_segments() - ref.path)[m].max()) / mean_path,
                       "inc": np.abs(out.incidences() - ref.incidence)[m].max()}
                for key, v in err.items():
                    worst[key] = max(worst[key], float(v))
                tols = {key: STRICT_TOL[key] + frame_noise for key in STRICT_TOL}
                if all(err[key] <= tols[key] for key in err):
                    continue
                # -- adjudication by truth: whole chain in long double for the rays that got this far
                assert T.HAVE_LD, f"{tag}, mode {mode}, element {k}: {err} beyond {tols} and no long double to adjudicate"
                stats["adjudicated"] += 1
                stats["adjudicated_seeds"].append(int(seed))
                surv = [o.numbers() for o in outs[:k + 1]]
                seeds_t = [o.path_segments()[:, -1] for o in outs[:k + 1]]
                keep, truth = T.chain_truth(els_o[:k + 1], a["src_point"], src.vectors(), a["src_number"], surv, seeds_t,
                                            scene["IgnoreDefects"])
                tP, tv, tpath, tinc = truth[-1]
                mk = well[keep]
                sel_p = np.searchsorted(out.numbers(), keep)
                sel_o = np.searchsorted(ref.number, keep)

                def vs_truth(pts, vec, path, inc):
                    return {"pos": float(np.abs(pts - tP)[mk].max() / scale), "dir": float(np.abs(vec - tv)[mk].max()),
                            "path": float(np.abs(path - tpath)[mk].max() / mean_path), "inc": float(np.abs(inc - tinc)[mk].max())}
                ep = vs_truth(out.points()[sel_p], out.vectors()[sel_p], out.paths_total()[sel_p], out.incidences()[sel_p])
                eo = vs_truth(ref.point[sel_o], ref.vector[sel_o], np.sum(ref.path, axis=1)[sel_o], ref.incidence[sel_o])
                for key in ep:
                    assert ep[key] <= max(tols[key], eo[key]), (
                        f"{tag}, mode {mode}, element {k}: {key} differs from the oracle by {err[key]:.3e} and the product is the "
                        f"one farther from the long-double truth ({ep[key]:.3e} vs the oracle's {eo[key]:.3e})")
                stats["adjudicated_worst"]["product"] = max(stats["adjudicated_worst"]["product"], max(ep.values()))
                stats["adjudicated_worst"]["oracle"] = max(stats["adjudicated_worst"]["oracle"], max(eo.values()))
            if mode == "chain" and seed % 3 == 0:
                # The scene-table launch (two copies of the chain in one launch) runs the fused kernel's per-ray code:
                # bit-identical to the single launch when that is the same kernel body (>= 2 elements, no defects); a
                # one-element chain is dispatched to the kind-specialised kernel and chains with defects go through the
                # run-time-dispatch body -- other instruction orders, so agreement to rounding there.
                many = mp.RayTracingCalculationMany([src, src], [els, els], IgnoreDefects=scene["IgnoreDefects"])
                same_body = len(els) >= 2 and not any(e.get("defects") for e in scene["elements"])
                for o2 in many:
                    for x, y in zip(o2, outs):
                        al = y.alive.cpu().numpy()
                        assert np.array_equal(x.alive.cpu().numpy(), al), f"{tag}: scene launch, survivors"
                        m_ = al.astype(bool)
                        xd, yd = x.data.cpu().numpy()[:, m_], y.data.cpu().numpy()[:, m_]
                        if same_body:
                            assert np.array_equal(xd, yd), f"{tag}: scene launch"
                        elif m_.any():
                            # the incidence angle of a near-normal ray amplifies rounding: 1e-9 rad as everywhere
                            assert np.abs(xd[0:3] - yd[0:3]).max() <= 1e-12 * scale and np.abs(xd[3:6] - yd[3:6]).max() <= 1e-12 \
                                and np.abs(xd[6] - yd[6]).max() <= 1e-12 * max(scale, np.abs(yd[6]).max()) \
                                and np.abs(xd[7] - yd[7]).max() <= 1e-9, f"{tag}: scene launch (rounding)"
    return {"worst": worst, "scenes_with_hits": hits, "scenes": len(list(seeds))}


def run_detector_fuzz(seeds, n_rays=1200):
    """Random scene -> autoplaced detector at a random distance (and a second, hand-posed one, tilted and off-centre):
    3-D hit points, detector-plane coordinates, centred coordinates and delays of the product against the oracle."""
    import ART.ModuleProcessing as mp
    import ART.ModuleDetector as mdet
    checked = 0
    for seed in seeds:
        scene, a = random_scene(seed, n_rays)
        rng = np.random.default_rng(seed + 77_000_000)
        src_o = orc.make_bundle(a["src_point"], a["src_vector"], a["src_number"], a["src_intensity"], None)
        last_o = orc.ray_tracing_calculation(src_o, orc.elements_from_scene(scene, a),
                                             IgnoreDefects=scene["IgnoreDefects"])[-1]
        if len(last_o) < 10 or (last_o.incidence >= GRAZING).any():
            continue          # conditioning of the traced bundle itself is the business of run_differential
        els = pc.build_elements(scene, a)
        src = RayBundle.from_arrays(a["src_point"], a["src_vector"], a["src_number"], None, None)
        last = mp.RayTracingCalculation(src, els, IgnoreDefects=scene["IgnoreDefects"])[-1]
        assert np.array_equal(last.numbers(), last_o.number)
        # the read-out is what is under test here: both sides read out the SAME bundle (the product's)
        last_o = orc.Bundle(last.points(), last.vectors(), last.numbers(), last.path_segments(), last.incidences(),
                            np.full(len(last), np.nan), None)
        dist_ = float(rng.uniform(20.0, 900.0))
        Do = orc.detector_autoplace(last_o, dist_)
        D = mdet.Detector(np.array(els[-1].position, float))
        D.autoplace(last, dist_)
        scale = max(1.0, np.abs(Do.centre).max(), np.abs(last_o.point).max())
        tag = f"detector seed {seed}"
        assert np.abs(D.centre - Do.centre).max() <= 1e-10 * scale, tag
        assert np.abs(D.normal - Do.normal).max() <= 1e-10, tag
        assert abs(D.get_distance() - orc.detector_distance(Do)) <= 1e-10 * scale, tag
        # second pose: tilted by up to 30 degrees and shifted sideways, same for both implementations
        tilt = rng.normal(size=3)
        n2 = Do.normal + 0.5 * rng.uniform(0, 1) * (tilt - np.dot(tilt, Do.normal) * Do.normal) / np.linalg.norm(tilt)
        n2 /= np.linalg.norm(n2)
        c2 = Do.centre + rng.normal(scale=2.0, size=3)
        poses = [(Do.refpoint, Do.centre, Do.normal), (Do.refpoint, c2, n2)]
        if seed % 7 == 0:       # detector normal exactly along -z / +z: RotationPoint's special cases in the read-out
            poses.append((Do.refpoint, Do.centre, np.array([0.0, 0.0, -1.0 if seed % 2 else 1.0])))
        for ref_pt, c, nrm in poses:
            Dq = orc.Detector(np.array(c, float), np.array(nrm, float), np.array(ref_pt, float))
            den = np.abs(last_o.vector @ Dq.normal)
            if den.min() < 0.05:
                continue      # rays nearly parallel to the detector plane: ill-conditioned read-out
            Dp = mdet.Detector(np.array(ref_pt, float), np.array(c, float), np.array(nrm, float))
            P3o = orc.detector_points3d(Dq, last_o)
            sc = max(1.0, np.abs(P3o).max())
            amp = 1.0 / den.min()
            assert np.abs(Dp.get_PointList3D(last) - P3o).max() <= 1e-10 * sc * amp, tag
            assert np.abs(Dp.get_PointList2D(last) - orc.detector_points2d(Dq, last_o)).max() <= 1e-10 * sc * amp, tag
            assert np.abs(Dp.get_PointList2DCentre(last) - orc.detector_points2dcentre(Dq, last_o)).max() \
                <= 1e-10 * sc * amp, tag
            mean_t_fs = np.mean(orc.optical_paths(Dq, last_o)) / orc.LightSpeed * 1e15
            assert np.abs(Dp.get_Delays(last) - orc.detector_delays(Dq, last_o)).max() <= 1e-10 * mean_t_fs * amp, tag
            # the read-out fused behind the tracing launch: per-ray values bit-identical to the separate read-out,
            # counts / minima / maxima exact, sums to rounding
            fus_last = mp.RayTracingCalculation(src, els, IgnoreDefects=scene["IgnoreDefects"], detector=Dp)[-1]
            fus = Dp.readout(fus_last, sync=False)
            assert fus_last._fused_readout is not None and fus["X"] is fus_last._fused_readout[2]["X"], tag
            # ... of the SAME bundle (a one-element or defect chain is traced by another kernel body with the read-out
            # fused than without: equal to rounding, not bit for bit)
            fus_last._fused_readout = None
            sep = Dp.readout(fus_last, sync=False)
            assert sep["X"] is not fus["X"], tag
            m_ = fus_last.alive.cpu().numpy().astype(bool)
            assert np.array_equal(m_, last.alive.cpu().numpy().astype(bool)), tag
            for key in ("X", "Y", "opl"):
                assert np.array_equal(fus[key].cpu().numpy()[m_], sep[key].cpu().numpy()[m_]), f"{tag}: fused {key}"
            fs, ss = fus["stats_dev"].cpu().numpy(), sep["stats_dev"].cpu().numpy()
            assert all(fs[k] == ss[k] for k in (0, 2, 3, 4, 5, 12, 13)), f"{tag}: fused statistics (exact slots)"
            # second moments are about provisional centres (0, 0, 0): compare against the magnitude of the summands
            mag = max(1.0, float(np.abs(ss).max()))
            assert np.abs(fs - ss).max() <= 1e-11 * mag, f"{tag}: fused statistics (sums)"
            checked += 1
    return checked


def run_source_fuzz(seeds):
    """Point sources and plane-wave disks with random and special axes (parallel / antiparallel to ez: the
    RotationPoint special cases) + Gaussian weights, product (device-generated) against the oracle."""
    import ART.ModuleSource as msource
    for seed in seeds:
        rng = np.random.default_rng(seed + 55_000_000)
        axis = [rng.normal(size=3), np.array([0.0, 0.0, 1.0]), np.array([0.0, 0.0, -1.0]),
                np.array([0.0, 0.0, -3.0]), np.array([1e-12, 0.0, 1.0])][seed % 5]
        S = rng.uniform(-300.0, 300.0, 3)
        n = int(rng.integers(50, 4000))
        div = float(rng.uniform(1e-4, 0.3))
        rad = float(rng.uniform(0.1, 60.0))
        for tag, b, q in (("point", msource.PointSource(S, axis, div, n, 50e-6), orc.point_source(S, axis, div, n)),
                          ("disk", msource.PlaneWaveDisk(S, axis, rad, n, 50e-6), orc.plane_wave_disk(S, axis, rad, n))):
            b = msource.ApplyGaussianIntensityToRayList(b)
            q = orc.apply_gaussian_intensity(q)
            t = f"source seed {seed} ({tag})"
            assert len(b) == len(q), t
            assert np.abs(b.points() - q.point).max() <= 1e-12 * max(1.0, np.abs(q.point).max()), t
            assert np.abs(b.vectors() - q.vector).max() <= 1e-12, t
            assert np.abs(b.intensities() - q.intensity).max() <= 1e-10, t
