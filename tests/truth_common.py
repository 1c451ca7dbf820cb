"""Extended-precision truth for ONE element acting on rays: 80-bit long double (64-bit significand, eps 1.1e-19) from
the element's fp64 parameters and the rays' fp64 state to the hit point, reflected direction, segment length and
incidence angle.  TEST INFRASTRUCTURE (the judge of the differential fuzz harness and of tests/test_accuracy_truth.py).

It is not a third implementation of the candidate rules: which root of the surface equation a ray takes is decided by
the implementations under test (they must agree on the survivors bit for bit); the truth REFINES that root -- for the
quadrics the closed-form root of the long-double quadratic that is nearest to the seed, for the torus Newton on the
implicit function from the seed -- and then follows the reference's formulas in long double:

  frames        ART/ModuleProcessing.py:284-295, :306-309 with RotationPoint's special cases (ART/ModuleGeometry.py:333-343)
  surfaces      ART/ModuleMirror.py:73-82 (plane), :163-178 (sphere), :325-347 (parabola), :443-478 (torus, implicit form),
                :662-683 (ellipsoid), :824-844 (cylinder); ART/ModuleMask.py:51-61
  normals       :84-87, :180-183, :349-355, :480-498, :685-693, :846-849
  deformation   ART/ModuleMirror.py:952-980 with the Zernike recurrences of ART/recursive_zernike_generator.py:35-254
                (oracle.zernike_tables evaluated on long-double arguments) and a bilinear height-map look-up
  reflection    ART/ModuleMirror.py:878-906, incidence = Kahan angle ART/ModuleGeometry.py:40-44
"""
import numpy as np

from oracle import art_oracle as orc

LD = np.longdouble
HAVE_LD = np.finfo(LD).eps < 1e-18
PI = LD(4) * np.arctan(LD(1))


def ld(a):
    return np.asarray(a, dtype=LD)


def _norm(v):
    return np.sqrt((v * v).sum(axis=-1))


def kahan_angle(U, V):
    """ART/ModuleGeometry.py:40-44 row-wise in long double."""
    u, v = _norm(U)[..., None], _norm(V)[..., None]
    return 2 * np.arctan2(_norm(U * v - V * u), _norm(U * v + V * u))


def rotation_matrix(axis1, axis2):
    """RotationPoint(., axis1, axis2) (ART/ModuleGeometry.py:333-343) as a 3x3 long-double matrix: identity below 1e-10 rad,
    the point inversion -I within 1e-10 of pi, else the rotation by the Kahan angle about axis1 x axis2."""
    a, b = ld(axis1), ld(axis2)
    ang = kahan_angle(a[None, :], b[None, :])[0]
    I = np.eye(3, dtype=LD)
    if abs(ang) < 1e-10:
        return I
    if abs(ang - PI) < 1e-10:
        return -I
    k = np.cross(a, b)
    k = k / _norm(k)
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]], dtype=LD)
    return I + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)


def frame_map(E):
    """fwd (lab -> optic frame) of an oracle Element: R2 R1, R1 = normal -> ez, R2 = R1 majoraxis -> ex."""
    R1 = rotation_matrix(E.normal, [0.0, 0.0, 1.0])
    m1 = R1 @ ld(E.majoraxis)
    R2 = rotation_matrix(m1, [1.0, 0.0, 0.0])
    return R2 @ R1


def _surface(O, A, u, t_seed):
    """Ray parameter of the hit on the UNDEFORMED optic: the root of the surface equation nearest to t_seed."""
    k, q = O.kind, O.params
    ux, uy, uz = u[:, 0], u[:, 1], u[:, 2]
    xA, yA, zA = A[:, 0], A[:, 1], A[:, 2]
    if k in ("plane", "mask"):
        return -zA / uz
    if k == "torus":
        R, r = LD(q["R"]), LD(q["r"])
        t = ld(t_seed).copy()
        for _ in range(8):
            P = A + t[:, None] * u
            rho = np.sqrt(P[:, 0] ** 2 + P[:, 2] ** 2)
            # the quartic's two factors: the torus proper (rho - R) and, for r > R, the inner "lemon" (rho + R); the
            # seed lies on one of them -- refine on the one whose residual is smaller
            F1 = (rho - R) ** 2 + P[:, 1] ** 2 - r ** 2
            F2 = (rho + R) ** 2 + P[:, 1] ** 2 - r ** 2
            drho = (P[:, 0] * ux + P[:, 2] * uz) / rho
            d1 = 2 * ((rho - R) * drho + P[:, 1] * uy)
            d2 = 2 * ((rho + R) * drho + P[:, 1] * uy)
            use2 = (np.abs(F2) < np.abs(F1)) & (r > R)
            t = t - np.where(use2, F2 / d2, F1 / d1)
        return t
    if k == "sphere":
        a, b, c = (u * u).sum(1), 2 * (u * A).sum(1), (A * A).sum(1) - LD(q["R"]) ** 2
    elif k == "parabola":
        p = LD(q["p"])
        a, b, c = ux ** 2 + uy ** 2, 2 * (ux * xA + uy * yA) - 2 * p * uz, xA ** 2 + yA ** 2 - 2 * p * zA
    elif k == "ellipsoid":
        ea, eb = LD(q["a"]), LD(q["b"])
        a = (uy ** 2 + uz ** 2) / eb ** 2 + (ux / ea) ** 2
        b = 2 * ((uy * yA + uz * zA) / eb ** 2 + (ux * xA) / ea ** 2)
        c = (yA ** 2 + zA ** 2) / eb ** 2 + (xA / ea) ** 2 - 1
    elif k == "cylinder":
        a, b, c = uy ** 2 + uz ** 2, 2 * (uy * yA + uz * zA), yA ** 2 + zA ** 2 - LD(q["R"]) ** 2
    else:
        raise ValueError(k)
    disc = np.maximum(b * b - 4 * a * c, 0)
    qq = -(b + np.where(b >= 0, 1, -1) * np.sqrt(disc)) / 2
    with np.errstate(divide="ignore", invalid="ignore"):
        t1 = np.where(a != 0, qq / np.where(a != 0, a, 1), -c / b)     # a == 0: the linear root (np.roots drops the zero)
        t2 = np.where(qq != 0, c / np.where(qq != 0, qq, 1), t1)
    ts = ld(t_seed)
    return np.where(np.abs(t1 - ts) <= np.abs(t2 - ts), t1, t2)


def _base_normal(O, P):
    k, q = O.kind, O.params
    x, y, z = P[:, 0], P[:, 1], P[:, 2]
    if k in ("plane", "mask"):
        g = np.stack([0 * x, 0 * x, 0 * x + 1], axis=1)
    elif k == "sphere":
        g = -P
    elif k == "parabola":
        g = np.stack([-x, -y, 0 * x + LD(q["p"])], axis=1)
    elif k == "torus":
        R, r = LD(q["R"]), LD(q["r"])
        S = x * x + y * y + z * z
        g = -np.stack([x * (S - R * R - r * r), y * (S + R * R - r * r), z * (S - R * R - r * r)], axis=1)
    elif k == "ellipsoid":
        a, b = LD(q["a"]), LD(q["b"])
        g = np.stack([-x / a ** 2, -y / b ** 2, -z / b ** 2], axis=1)
    else:
        g = np.stack([0 * x, -y, -z], axis=1)
    return g / _norm(g)[:, None]


def _grid_offset(G, P):
    """Bilinear look-up of a `Fourrier` height map on the reference's grid (ART/ModuleDefects.py:104-110, :131-137)."""
    d = np.asarray(G.deformation)
    X = np.linspace(-G.rect[0] / 2, G.rect[0] / 2, num=d.shape[1])
    Y = np.linspace(-G.rect[1] / 2, G.rect[1] / 2, num=d.shape[0])
    x, y = P[:, 0], P[:, 1]
    ix = np.clip(np.searchsorted(X, x.astype(np.float64), side="right") - 1, 0, len(X) - 2)
    iy = np.clip(np.searchsorted(Y, y.astype(np.float64), side="right") - 1, 0, len(Y) - 2)
    tx = (x - ld(X[ix])) / (ld(X[ix + 1]) - ld(X[ix]))
    ty = (y - ld(Y[iy])) / (ld(Y[iy + 1]) - ld(Y[iy]))
    h = ld(d.T)                          # [ix, iy]
    return (h[ix, iy] * (1 - tx) + h[ix + 1, iy] * tx) * (1 - ty) + (h[ix, iy + 1] * (1 - tx) + h[ix + 1, iy + 1] * tx) * ty


def element_truth(E, point, vector, t_seed, ignore_defects=True, fwd=None):
    """One oracle Element acting on rays (point, vector: (n, 3), any float type; they are taken as exact) that all hit it;
    t_seed: an approximation of each ray's segment length.  Returns long-double (point', vector', segment, incidence)."""
    O = E.optic
    F = frame_map(E) if fwd is None else fwd
    C = ld(O.centre())
    P0, v0 = ld(point), ld(vector)
    A = (P0 - ld(E.position)) @ F.T + C
    u = v0 @ F.T
    u = u / _norm(u)[:, None]
    t = _surface(O, A, u, t_seed)
    P = A + t[:, None] * u
    if O.kind == "mask":
        v = u
        inc = kahan_angle(u, ld([[0.0, 0.0, 1.0]]))
    else:
        n = _base_normal(O, P)
        if O.defects or O.grids:
            h = 0
            for D in O.defects:
                h = h + orc.zernike_offset(D, P - C)
            for G in O.grids:
                h = h + _grid_offset(G, P - C)
            s = h / np.cos(kahan_angle(-u, n))
            t = t - s
            P = P - u * s[:, None]
            n = _base_normal(O, P)
            if O.defects and not ignore_defects:
                gx, gy = -n[:, 0] / n[:, 2], -n[:, 1] / n[:, 2]
                for D in O.defects:
                    zn = orc.zernike_normal(D, P - C)          # (-dX, -dY, 1)
                    gx, gy = gx - zn[:, 0], gy - zn[:, 1]
                n = np.stack([-gx, -gy, 0 * gx + 1], axis=1)
                n = n / _norm(n)[:, None]
        dn = (u * n).sum(1)
        v = u - 2 * dn[:, None] * n
        v = v / _norm(v)[:, None]
        inc = kahan_angle(-u, n)
    return (P - C) @ F + ld(E.position), v @ F, t, inc


def chain_truth(elements, point, vector, numbers, survivors, seeds, ignore_defects=True):
    """The whole chain in long double for the rays `numbers` (source order) that survive every element: survivors[k] =
    ray numbers after element k, seeds[k] = segment lengths of those survivors (from an implementation under test).
    Returns, per element, (point, vector, cumulative path, incidence) for the rays that survive the whole chain."""
    keep = np.asarray(survivors[-1])
    sel = np.searchsorted(np.asarray(numbers), keep)
    P, v = ld(point)[sel], ld(vector)[sel]
    path = np.zeros(len(keep), dtype=LD)
    out = []
    for k, E in enumerate(elements):
        pos_k = np.searchsorted(np.asarray(survivors[k]), keep)
        P, v, t, inc = element_truth(E, P, v, np.asarray(seeds[k])[pos_k], ignore_defects)
        path = path + t
        out.append((P, v, path, inc))
    return keep, out
