"""The oracle's pin rests on golden vectors produced by the reference with OUR stand-in for `numpy-quaternion`
(tests/golden/_standin/quaternion.py; the package is absent from the image).  This test pins the stand-in against an
independent implementation: the reference's `RotationAroundAxis` call sequence (ART/ModuleGeometry.py:321-329)
executed on the stand-in must equal SciPy's `Rotation.from_rotvec(angle * axis).apply(v)` and an 80-bit Rodrigues
formula to a few ulp, for random
axes and angles including the limits angle -> 0 and angle -> pi that `RotationPoint` special-cases (:333-343)."""
import importlib.util
import os

import numpy as np
from scipy.spatial.transform import Rotation

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("standin_quaternion", os.path.join(HERE, "golden", "_standin", "quaternion.py"))
_q = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_q)
quaternion = _q.quaternion


def rotation_around_axis_via_standin(Axis, Angle, Vector):
    """The statements of ART/ModuleGeometry.py:321-329, with the stand-in class in place of numpy-quaternion."""
    a = np.array([0.0] + np.asarray(Axis, dtype=float))      # elementwise 0.0 + Axis, as in the reference
    rot_axis = a / np.linalg.norm(a)
    axis_angle = (Angle * 0.5) * rot_axis
    vec = quaternion(*Vector)
    qlog = quaternion(*axis_angle)
    q = np.exp(qlog)                                           # object-dtype ufunc dispatch -> qlog.exp()
    return (q * vec * np.conjugate(q)).imag


def rodrigues_long_double(Axis, Angle, Vector):
    """Second independent implementation: Rodrigues' formula in 80-bit long double (the 'truth' both are measured on)."""
    L = np.longdouble
    a = np.array(Axis, dtype=L)
    a = a / np.sqrt((a * a).sum())
    v = np.array(Vector, dtype=L)
    c, s = np.cos(L(Angle)), np.sin(L(Angle))
    return v * c + np.cross(a, v) * s + a * (a @ v) * (1 - c)


def _worst(angles, rng, scale_axis=1.0):
    """Worst errors in ulp of |v|, divided by (1 + |angle|): the rotation vector angle * axis is itself rounded, which
    moves the result by |angle| ulp before any algorithm starts."""
    eps = np.finfo(float).eps
    w_truth = w_scipy = 0.0
    for ang in angles:
        axis = rng.normal(size=3) * scale_axis
        v = rng.normal(size=3) * 10 ** rng.uniform(-3, 3)
        got = rotation_around_axis_via_standin(axis, ang, v)
        ref = Rotation.from_rotvec(ang * axis / np.linalg.norm(axis)).apply(v)
        tru = rodrigues_long_double(axis, ang, v)
        unit = eps * np.linalg.norm(v) * (1.0 + abs(ang))
        w_truth = max(w_truth, float(np.abs(got - tru).max()) / unit)
        w_scipy = max(w_scipy, float(np.abs(got - ref).max()) / unit)
    return w_truth, w_scipy


def test_standin_rotation_matches_scipy_and_long_double():
    """Measured here: stand-in vs 80-bit truth 3.4 ulp for angles in [0, pi] (SciPy's rotation matrix: 3.7 ulp), 3.8 ulp
    next to pi, 7 ulp at |angle| = 2 pi (SciPy 5.8) -- i.e. the stand-in is as accurate as SciPy; the two differ from
    each other by up to 4.8 ulp ([0, pi]).  Everything is 5 orders of magnitude inside the 1e-10 parity tolerance."""
    rng = np.random.default_rng(2024)
    cases = {
        "generic [-2pi, 2pi]": rng.uniform(-2 * np.pi, 2 * np.pi, 4000),
        "[0, pi] (RotationPoint's range)": rng.uniform(0, np.pi, 4000),
        "angle -> 0": np.concatenate([10.0 ** rng.uniform(-16, -6, 500), -10.0 ** rng.uniform(-16, -6, 500), [0.0]]),
        "angle -> pi": np.pi + np.concatenate([10.0 ** rng.uniform(-16, -6, 500), -10.0 ** rng.uniform(-16, -6, 500), [0.0]]),
    }
    for name, angles in cases.items():
        w_truth, w_scipy = _worst(angles, rng)
        assert w_truth <= 4.0, f"{name}: {w_truth:.2f} ulp (1 + |angle|) from the long-double truth"
        assert w_scipy <= 6.0, f"{name}: {w_scipy:.2f} ulp (1 + |angle|) from SciPy"
    w_truth, w_scipy = _worst(rng.uniform(-np.pi, np.pi, 1000), rng, scale_axis=1e3)      # axis far from unit length
    assert w_truth <= 4.0 and w_scipy <= 6.0, (w_truth, w_scipy)


def test_standin_surface_used_by_the_reference():
    """3-argument constructor = pure-vector quaternion, exp of a pure quaternion, Hamilton product, conjugate, .imag."""
    q = quaternion(1.0, 2.0, 3.0)
    assert (q.w, q.x, q.y, q.z) == (0.0, 1.0, 2.0, 3.0)
    i, j, k = quaternion(0, 1, 0, 0), quaternion(0, 0, 1, 0), quaternion(0, 0, 0, 1)
    assert ((i * j).w, (i * j).x, (i * j).y, (i * j).z) == (0.0, 0.0, 0.0, 1.0)          # ij = k
    assert ((j * i).z, (k * i).y, (j * k).x) == (-1.0, 1.0, 1.0)                          # ji = -k, ki = j, jk = i
    e = np.exp(quaternion(0.0, 0.0, np.pi / 2))
    assert abs(e.w) < 1e-16 and abs(e.z - 1.0) < 1e-16                                     # exp(pi/2 k) = k
    z = np.exp(quaternion(0.0, 0.0, 0.0))
    assert (z.w, z.x, z.y, z.z) == (1.0, 0.0, 0.0, 0.0)
    c = np.conjugate(quaternion(1.0, 2.0, 3.0, 4.0))
    assert (c.w, c.x, c.y, c.z) == (1.0, -2.0, -3.0, -4.0)
    assert np.array_equal(quaternion(1.0, 2.0, 3.0, 4.0).imag, np.array([2.0, 3.0, 4.0]))
