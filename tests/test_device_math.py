"""Scalar helpers of csrc/art_device.h checked on their own through the CPU twin (oracle/_twin, the same header compiled
by g++; on the device only the reciprocal / square-root seeds differ): the Kahan angle with one norm, atan on [0, 1], and
the constants prepare_element() derives.  The tracing tests cover them only through whole scenes."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from attosecondraytracing_amd import _abi  # noqa: E402
import twin_backend  # noqa: E402


def _lib():
    lib = C.CDLL(twin_backend.build_twin())
    dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
    lib.art_cpu_kahan_angle_unit.argtypes = [dp, dp, C.c_int64, dp]
    lib.art_cpu_kahan_angle_unit.restype = None
    lib.art_cpu_atan01.argtypes = [dp, C.c_int64, dp]
    lib.art_cpu_atan01.restype = None
    lib.art_cpu_prepare_element.argtypes = [C.POINTER(_abi.ArtElementDesc), C.POINTER(_abi.ArtElementDesc)]
    lib.art_cpu_prepare_element.restype = None
    return lib


def _unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def test_atan01_against_long_double():
    lib = _lib()
    q = np.concatenate([np.linspace(0.0, 1.0, 20001), [0.19891236737965800, 0.19891236737965803, 0.66817863791929891,
                                                        0.66817863791929902, 1e-300, 1e-17, 1.0 - 1e-16]])
    out = np.empty_like(q)
    lib.art_cpu_atan01(np.ascontiguousarray(q), len(q), out)
    want = np.arctan(q.astype(np.longdouble))
    err = np.abs(out.astype(np.longdouble) - want)
    assert float(err.max()) <= 2.3e-16          # ~1 ulp at pi/4
    assert float((err / np.maximum(want, np.longdouble(1e-300))).max()) <= 4e-16


def test_kahan_angle_of_unit_vectors_against_long_double():
    lib = _lib()
    rng = np.random.default_rng(11)
    n = 20000
    u = _unit(rng.normal(size=(n, 3)))
    # angles from 1e-12 rad to pi - 1e-12 rad, log-spaced at both ends + uniform in between
    ang = np.concatenate([10.0 ** rng.uniform(-12, 0, n // 4), np.pi - 10.0 ** rng.uniform(-12, 0, n // 4),
                          rng.uniform(0.0, np.pi, n - 2 * (n // 4))])
    t = _unit(np.cross(u, rng.normal(size=(n, 3))))            # unit, orthogonal to u
    v = _unit(np.cos(ang)[:, None] * u + np.sin(ang)[:, None] * t)
    out = np.empty(n)
    lib.art_cpu_kahan_angle_unit(np.ascontiguousarray(u), np.ascontiguousarray(v), n, out)
    # truth from the stored (rounded) vectors in long double: 2 atan2(|u|v - v|u||, |u|v + v|u||)
    ul, vl = u.astype(np.longdouble), v.astype(np.longdouble)
    nu, nv = np.sqrt((ul * ul).sum(1))[:, None], np.sqrt((vl * vl).sum(1))[:, None]
    a, b = ul * nv - vl * nu, ul * nv + vl * nu
    want = 2 * np.arctan2(np.sqrt((a * a).sum(1)), np.sqrt((b * b).sum(1)))
    err = np.abs(out.astype(np.longdouble) - want)
    # From 1e-3 rad on (long double is a good enough truth there): 1.5e-15 of the angle below 1 rad; beyond, and near a
    # half-turn where the result is pi - 2 atan(.), 1.5 ulp of pi (measured worst: 5.0e-16 near 2.7 rad).
    tol = np.where(want < 1e-3, np.inf, np.where(want < 1.0, np.longdouble(1.5e-15) * want, np.longdouble(7e-16)))
    assert bool((err <= tol).all()), (float((err / tol).max()), float(want[np.argmax(err / tol)]))
    # Small angles keep their RELATIVE accuracy (what Kahan's form is for).  Truth in 60-digit decimal arithmetic (at
    # 1e-8 rad long double resolves the angle between two stored vectors to 1e-11 only).  Allowed: 1.5e-15 of the angle
    # plus what treating the vectors as exactly unit costs -- |u| and |v| differ by d ~ 2e-16, a radial component that
    # enters |u - v|^2 beside the angular one: relative (d / angle)^2 / 2, i.e. 1e-8 at 1e-12 rad (an absolute 1e-20
    # rad) and below 1e-16 from 1e-8 rad on.
    from decimal import Decimal, getcontext
    getcontext().prec = 60
    small = np.flatnonzero(want < 1e-3)[:400]
    assert len(small) > 100
    for i in small:
        U, V = [Decimal(float(c)) for c in u[i]], [Decimal(float(c)) for c in v[i]]
        lu, lv = sum(c * c for c in U).sqrt(), sum(c * c for c in V).sqrt()
        A = sum((p * lv - q * lu) ** 2 for p, q in zip(U, V)).sqrt()
        B = sum((p * lv + q * lu) ** 2 for p, q in zip(U, V)).sqrt()
        x = A / B
        th, term, k = Decimal(0), x, 0
        while abs(term) > Decimal(10) ** -55:
            th += term / (2 * k + 1) * (-1) ** k
            term *= x * x
            k += 1
        th *= 2
        rel = abs(Decimal(float(out[i])) - th) / th
        assert rel <= Decimal(1.5e-15) + (Decimal(3e-16) / th) ** 2, (float(th), float(rel))
    # exact cases
    e = np.array([[0.0, 0.0, 1.0], [0.0, 0.0, 1.0], [1.0, 0.0, 0.0]])
    f = np.array([[0.0, 0.0, 1.0], [0.0, 0.0, -1.0], [0.0, 1.0, 0.0]])
    o3 = np.empty(3)
    lib.art_cpu_kahan_angle_unit(e, f, 3, o3)
    assert o3[0] == 0.0 and o3[1] == np.pi and abs(o3[2] - np.pi / 2) <= 2.3e-16


def test_prepared_descriptor_slots():
    lib = _lib()
    rng = np.random.default_rng(5)
    d = _abi.ArtElementDesc()
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    pos, centre = rng.uniform(-2000, 2000, 3), np.array([0.0, 0.0, -5585.0 - 173.0])
    d.kind = 3                               # ART_TORUS
    d.fwd[:] = q.ravel()
    d.bwd[:] = q.T.ravel()
    d.pos[:] = pos
    d.centre[:] = centre
    R, r = 5585.1223, 173.6482
    d.mp[0], d.mp[1] = R, r
    before = bytes(d)
    out = _abi.ArtElementDesc()
    lib.art_cpu_prepare_element(C.byref(d), C.byref(out))
    assert bytes(d) == before                                   # the caller's struct is not written
    ql, pl, cl = q.astype(np.longdouble), pos.astype(np.longdouble), centre.astype(np.longdouble)
    in_off, out_off = cl - ql @ pl, pl - ql.T @ cl
    got = np.array(out.bwd[:])
    assert np.abs(got[0:3] - in_off.astype(float)).max() <= 1e-12 and np.abs(got[3:6] - out_off.astype(float)).max() <= 1e-12
    # a point goes in and comes back to 1e-12 mm through the two offsets
    P = rng.uniform(-3000, 3000, 3)
    back = q.T @ (q @ P + got[0:3]) + got[3:6]
    assert np.abs(back - P).max() <= 2e-12
    assert got[6] == r * r and got[7] == (R + 0.7 * r) ** 2 and got[8] == (0.7 * r) ** 2
    assert out.pos[0] == R + r and out.pos[1] == R * R + r * r and out.pos[2] == R * R - r * r
    assert out.mp[0] == R and out.mp[1] == 0.5 / R and out.mp[2] == 1.0 / ((R + r) ** 2) and out.mp[3] == 1.0 / (r * (R + r))
    assert out.flags & 0x80000000 == 0
    d.mp[0], d.mp[1] = 100.0, 150.0                             # r > R: the "lemon" flag
    lib.art_cpu_prepare_element(C.byref(d), C.byref(out))
    assert out.flags & 0x80000000


def test_detector_placement_on_the_device_matches_the_host_shell():
    """analysis_place (art_device.h, what art_analyse_bundles runs between its passes): the detector of
    Detector.autoplace (ART/ModuleDetector.py:109-137) from the bundle's sums -- normal, centre, reference point -- and the
    matrix of RotationPoint(., normal, ez) with its special cases (ART/ModuleGeometry.py:333-343), against the host shell's
    NumPy versions; the provisional path centre is the mean path + the mean ray's distance to the detector."""
    import ART.ModuleGeometry as mgeo
    lib = C.CDLL(twin_backend.build_twin())
    place = lib.art_cpu_analysis_place
    place.restype = C.c_int
    place.argtypes = [_abi.c_double_p, C.c_int32, C.c_double, _abi.c_double_p, _abi.c_double_p, _abi.c_double_p, _abi.c_double_p]
    rng = np.random.default_rng(5)
    ez = np.array([0.0, 0.0, 1.0])
    specials = [np.array([0.0, 0.0, -1.0]), np.array([0.0, 0.0, 1.0]), _unit(np.array([1e-11, 0.0, -1.0])),
                _unit(np.array([0.0, 3e-11, 1.0])), _unit(np.array([1e-9, 0.0, -1.0])), np.array([1.0, 0.0, 0.0])]
    worst = 0.0
    for k in range(300):
        cnt = float(rng.integers(1, 10 ** 7))
        mv = specials[k] if k < len(specials) else _unit(rng.normal(size=3)) * rng.uniform(0.9, 1.0)
        mp_ = rng.normal(size=3) * 500
        D = float(rng.uniform(1, 2000))
        mean_path = float(rng.uniform(0, 3000))
        sums = np.concatenate([[cnt], mp_ * cnt, mv * cnt, [cnt * 0.7], [mean_path * cnt]])
        out = (C.c_double * 22)()
        zero = (C.c_double * 3)()
        assert place((C.c_double * 9)(*sums), 0, D, zero, zero, zero, out) == 0
        o = np.array(out)
        # the host shell's sequence: mean vector -> Ray.vector setter -> negated -> Detector.normal setter
        pt, v = sums[1:4] / cnt, sums[4:7] / cnt
        v = v / np.linalg.norm(v)
        n = -v
        n = n / np.linalg.norm(n)
        assert np.abs(o[3:6] - n).max() <= 4.5e-16 and np.abs(o[15:18] - pt).max() <= 1e-12
        assert np.abs(o[0:3] - (pt - n * D)).max() <= 1e-12 * max(1.0, D)
        assert np.abs(o[18:21] - v).max() <= 4.5e-16          # (2 ulp: the norm is an fma chain on the device side)
        M = mgeo.rotation_matrix(n, ez)
        worst = max(worst, np.abs(o[6:15].reshape(3, 3) - M).max())
        assert np.abs(o[6:15].reshape(3, 3) @ o[3:6] - (M @ n)).max() <= 2e-15          # (a few ulp of 1)
        assert abs(o[21] - (mean_path + D)) <= 1e-12 * (mean_path + D)
    assert worst <= 2e-15, worst          # entries of the rotation: the device's Kahan angle (polynomial atan) then sin / cos of half of it
    # manual mode: pose used bit for bit
    c_, n_, r_ = np.array([1.0, 2.0, 3.0]), _unit(np.array([0.3, -0.2, 0.9])), np.array([4.0, 5.0, 6.0])
    out = (C.c_double * 22)()
    assert place((C.c_double * 9)(*sums), 1, 0.0, (C.c_double * 3)(*c_), (C.c_double * 3)(*n_), (C.c_double * 3)(*r_), out) == 0
    o = np.array(out)
    assert np.array_equal(o[0:3], c_) and np.array_equal(o[3:6], n_) and np.array_equal(o[15:18], r_)
