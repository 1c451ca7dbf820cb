"""CPU ORACLE for ART's ray-bundle propagation hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module.  The product path (package `attosecondraytracing_amd`, alias `ART`) never does; it calls
the HIP library and fails loudly when that is missing.

What this is: a ray-vectorised NumPy restatement of the *reference's algorithm* for
`ModuleProcessing.RayTracingCalculation` and the detector read-out -- the same step sequence
(translate, two `RotationRay`s with renormalisation, translate, intersect, reflect, and back), the
same root finder (`np.roots` == eigenvalues of the companion matrix, here batched through
`np.linalg.eigvals`, which runs the same LAPACK `geev` per matrix), the same root filters
(|imag| < 1e-15, t > 1e-12), the same candidate rules (1 -> take, 2 -> closest, else miss), the
same Kahan angle, the same quaternion-sandwich rotations.  Each function cites the reference
file:line it follows (paths relative to /root/reference).

Parity status: PINNED against golden vectors produced by running the reference itself in the
build container (tests/golden/generate_goldens.py; fixtures tests/golden/*.npz).  Caveat stated
there and in DESIGN.md: the image has no `numpy-quaternion`, so the reference ran with our
60-line Hamilton-algebra stand-in; `zernike_tierA.npz` needed no stand-in at all.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

LightSpeed = 299792458000  # mm/s, ART/ModuleDetector.py:21

EZ = np.array([0.0, 0.0, 1.0])
EX = np.array([1.0, 0.0, 0.0])


# =============================================================================== scene description
@dataclass
class Support:
    """ART/ModuleSupport.py: kind in {round, roundhole, rect, recthole, rectrecthole}; p = ctor args."""
    kind: str
    p: Sequence[float]

    def circum_circ(self) -> float:
        # _CircumCirc: ModuleSupport.py:95-96, :181-182 (round*) ; :259-260, :353-354, :475-476 (rect*)
        if self.kind in ("round", "roundhole"):
            return float(self.p[0])
        return float(np.sqrt(self.p[0] ** 2 + self.p[1] ** 2) / 2)


@dataclass
class ZernikeDefect:
    """ART/ModuleDefects.py:149-174."""
    coeffs: Dict[Tuple[int, int], float]
    R: float


@dataclass
class GridDefect:
    """Height map of a `Fourrier` defect (ART/ModuleDefects.py:69-146): `deformation` as the reference stores it
    (shape [ny, nx]); offset = RegularGridInterpolator((X, Y), deformation.T, "linear") (:104-110, :131-137)."""
    deformation: np.ndarray
    rect: Sequence[float]

    def offset(self, P):
        from scipy.interpolate import RegularGridInterpolator
        d = self.deformation
        X = np.linspace(-self.rect[0] / 2, self.rect[0] / 2, num=d.shape[1])
        Y = np.linspace(-self.rect[1] / 2, self.rect[1] / 2, num=d.shape[0])
        return RegularGridInterpolator((X, Y), np.transpose(d), method="linear")(P[:, :2])


@dataclass
class Optic:
    """kind in {plane, sphere, parabola, torus, ellipsoid, cylinder, mask}; params per kind."""
    kind: str
    support: Support
    params: Dict[str, float] = field(default_factory=dict)
    defects: List[ZernikeDefect] = field(default_factory=list)
    type: str = ""
    grids: List[GridDefect] = field(default_factory=list)

    def is_mirror(self) -> bool:
        return self.kind != "mask"

    def centre(self) -> np.ndarray:
        k, q = self.kind, self.params
        if k in ("plane", "mask"):
            return np.array([0.0, 0.0, 0.0])  # ModuleMirror.py:89-91, ModuleMask.py:68-70
        if k in ("sphere", "cylinder"):
            return np.array([0.0, 0.0, -q["R"]])  # ModuleMirror.py:185-187, :851-853
        if k == "parabola":  # ModuleMirror.py:357-365
            return np.array([q["feff"] * np.sin(q["offaxis_rad"]), 0.0,
                             q["p"] * 0.5 - q["feff"] * np.cos(q["offaxis_rad"])])
        if k == "torus":
            return np.array([0.0, 0.0, -q["R"] - q["r"]])  # ModuleMirror.py:500-502
        if k == "ellipsoid":
            return ellipsoid_centre(q["a"], q["b"], q["offaxis_rad"])
        raise NameError("I don`t recognize the type of optical element " + k + ".")


def ellipsoid_centre(a, b, offaxis):
    """ART/ModuleMirror.py:695-714."""
    foci = 2 * np.sqrt(a ** 2 - b ** 2)
    h = -foci / 2 / np.tan(offaxis)
    R = np.sqrt(foci ** 2 / 4 + h ** 2)
    sign = 1
    if math.isclose(offaxis, np.pi / 2):
        h = 0
    elif offaxis > np.pi / 2:
        h = -h
        sign = -1
    aa = 1 - a ** 2 / b ** 2
    bb = -2 * h
    cc = a ** 2 + h ** 2 - R ** 2
    z = (-bb + sign * np.sqrt(bb ** 2 - 4 * aa * cc)) / (2 * aa)
    if math.isclose(z ** 2, b ** 2):
        return np.array([0.0, 0.0, -b])
    x = a * np.sqrt(1 - z ** 2 / b ** 2)
    return np.array([x, 0.0, sign * z])


@dataclass
class Element:
    """Pose of an optic in the lab frame: ART/ModuleOpticalElement.py:75-105."""
    optic: Optic
    position: np.ndarray
    normal: np.ndarray
    majoraxis: np.ndarray


@dataclass
class Bundle:
    """SoA image of a list of ART Ray objects (ART/ModuleOpticalRay.py:11-63).  `path` keeps the
    reference's tuple of per-segment lengths as columns."""
    point: np.ndarray      # (n,3)
    vector: np.ndarray     # (n,3) unit
    number: np.ndarray     # (n,) int64
    path: np.ndarray       # (n,k) segments, k>=1, first column is the initial 0.0
    incidence: np.ndarray  # (n,) nan where None
    intensity: np.ndarray  # (n,) nan where None
    wavelength: Optional[float] = None

    def __len__(self):
        return self.point.shape[0]

    def select(self, keep):
        return Bundle(self.point[keep], self.vector[keep], self.number[keep], self.path[keep],
                      self.incidence[keep], self.intensity[keep], self.wavelength)


def make_bundle(point, vector, number=None, intensity=None, wavelength=None) -> Bundle:
    point = np.ascontiguousarray(point, dtype=np.float64).reshape(-1, 3)
    vector = normalize_rows(np.ascontiguousarray(vector, dtype=np.float64).reshape(-1, 3))  # Ray.vector setter :85-90
    n = point.shape[0]
    number = np.arange(n, dtype=np.int64) if number is None else np.asarray(number, dtype=np.int64)
    intensity = np.full(n, np.nan) if intensity is None else np.asarray(intensity, dtype=np.float64)
    return Bundle(point, vector, number, np.zeros((n, 1)), np.full(n, np.nan), intensity, wavelength)


# =============================================================================== ModuleGeometry
def norm_rows(v):
    return np.sqrt(np.einsum("ij,ij->i", v, v))


def normalize_rows(v):
    """ART/ModuleGeometry.py:17-19 applied per row."""
    return v / norm_rows(v)[:, None]


def angle_between(U, V):
    """Kahan angle, ART/ModuleGeometry.py:40-44; U, V of shape (n,3) or (3,) broadcastable."""
    U = np.atleast_2d(U)
    V = np.atleast_2d(V)
    u = norm_rows(U)[:, None]
    v = norm_rows(V)[:, None]
    return 2 * np.arctan2(norm_rows(U * v - V * u), norm_rows(U * v + V * u))


def angle_between1(U, V) -> float:
    return float(angle_between(np.asarray(U, float), np.asarray(V, float))[0])


def rotation_around_axis(axis, angle, vec):
    """ART/ModuleGeometry.py:321-329: v' = q v q*, q = exp(angle/2 * axis/|axis|) (unit quaternion).
    `axis` (3,), `angle` scalar, `vec` (n,3) or (3,)."""
    single = np.ndim(vec) == 1
    vec = np.atleast_2d(np.asarray(vec, dtype=np.float64))
    axis = np.asarray(axis, dtype=np.float64)
    rot_axis = axis / np.linalg.norm(axis)
    aa = (angle * 0.5) * rot_axis
    vn = math.sqrt(aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2])
    if vn == 0.0:
        qw, qx, qy, qz = 1.0, 0.0, 0.0, 0.0
    else:
        s = math.sin(vn) / vn
        qw, qx, qy, qz = math.cos(vn), s * aa[0], s * aa[1], s * aa[2]
    x, y, z = vec[:, 0], vec[:, 1], vec[:, 2]
    # t = q * (0, v)
    tw = -qx * x - qy * y - qz * z
    tx = qw * x + qy * z - qz * y
    ty = qw * y - qx * z + qz * x
    tz = qw * z + qx * y - qy * x
    # r = t * conj(q); keep the vector part
    cx, cy, cz = -qx, -qy, -qz
    rx = tw * cx + tx * qw + ty * cz - tz * cy
    ry = tw * cy - tx * cz + ty * qw + tz * cx
    rz = tw * cz + tx * cy - ty * cx + tz * qw
    out = np.stack([rx, ry, rz], axis=1)
    return out[0] if single else out


def rotation_point(P, axis1, axis2):
    """ART/ModuleGeometry.py:333-343 incl. the two special cases (identity; point inversion -P)."""
    ang = angle_between1(axis1, axis2)
    if abs(ang) < 1e-10:
        return np.array(P, dtype=np.float64, copy=True)
    if abs(ang - np.pi) < 1e-10:
        return -np.asarray(P, dtype=np.float64)
    N = np.cross(axis1, axis2)
    return rotation_around_axis(N, ang, P)


def rotation_rays(point, vector, axis1, axis2):
    """ART/ModuleGeometry.py:357-378: rotate OA and OB = OA + u, u' = normalize(OB' - OA') (Ray setter)."""
    OAp = rotation_point(point, axis1, axis2)
    OBp = rotation_point(vector + point, axis1, axis2)
    return OAp, normalize_rows(OBp - OAp)


def np_roots_batch(coeffs):
    """Row-wise `np.roots` (ART/ModuleGeometry.py:84, :99).  Rows without leading/trailing zero
    coefficients go through one batched eigvals of their companion matrices (what np.roots does per
    row); the rare others fall back to np.roots itself.  Returns complex (n, deg) padded with nan."""
    coeffs = np.asarray(coeffs, dtype=np.float64)
    n, m = coeffs.shape
    deg = m - 1
    out = np.full((n, deg), np.nan + 0j, dtype=np.complex128)
    if n == 0:
        return out
    regular = (coeffs[:, 0] != 0) & (coeffs[:, -1] != 0) & np.all(np.isfinite(coeffs), axis=1)
    if regular.any():
        c = coeffs[regular]
        A = np.zeros((c.shape[0], deg, deg))
        if deg > 1:
            idx = np.arange(deg - 1)
            A[:, idx + 1, idx] = 1.0
        A[:, 0, :] = -c[:, 1:] / c[:, :1]
        out[regular] = np.linalg.eigvals(A)
    for i in np.nonzero(~regular)[0]:
        r = np.roots(coeffs[i])
        out[i, :len(r)] = r
    return out


def real_positive_roots(roots):
    """SolverQuadratic/SolverQuartic real filter (:87-89, :102-104) + KeepPositiveSolution (:110-120).
    Returns (t, valid) with t real (n,deg)."""
    valid = np.abs(roots.imag) < 1e-15          # nan compares False
    t = roots.real
    with np.errstate(invalid="ignore"):
        valid &= t > 1e-12
    return t, valid


def include_support(S: Support, x, y):
    """ART/ModuleSupport.py:68-70, :151-155, :228-230, :322-326, :431-435; ModuleGeometry.py:249-268."""
    p = S.p

    def disk(R, xx, yy):
        return (xx ** 2 + yy ** 2) <= R ** 2

    def rect(X, Y, xx, yy):
        return (np.abs(xx) <= abs(X / 2)) & (np.abs(yy) <= abs(Y / 2))

    if S.kind == "round":
        return disk(p[0], x, y)
    if S.kind == "roundhole":
        return disk(p[0], x, y) & ~disk(p[1], x - p[2], y - p[3])
    if S.kind == "rect":
        return rect(p[0], p[1], x, y)
    if S.kind == "recthole":
        return rect(p[0], p[1], x, y) & ~disk(p[2], x - p[3], y - p[4])
    if S.kind == "rectrecthole":
        return rect(p[0], p[1], x, y) & ~rect(p[2], p[3], x - p[4], y - p[5])
    raise ValueError(S.kind)


# =============================================================================== Zernike (ModuleDefects / recursive_zernike_generator)
def zernike_tables(x, y, max_order):
    """ART/recursive_zernike_generator.py:35-254, vectorised over points: dicts {(n,m): array}."""
    if max_order < 2:
        max_order = 2
    one = np.ones_like(x)
    zero = np.zeros_like(x)
    Z = {(0, 0): one, (1, 0): y, (1, 1): x}
    GX = {(0, 0): zero, (1, 0): zero, (1, 1): one}
    GY = {(0, 0): zero, (1, 0): one, (1, 1): zero}
    for n in range(2, max_order + 1):
        for m in range(0, n + 1):
            if m == 0:  # :79-95
                z = x * Z[(n - 1, 0)] + y * Z[(n - 1, n - 1)]
                gx = n * Z[(n - 1, 0)]
                gy = n * Z[(n - 1, n - 1)]
            elif m == n:  # :97-110
                z = x * Z[(n - 1, n - 1)] - y * Z[(n - 1, 0)]
                gx = n * Z[(n - 1, n - 1)]
                gy = -1.0 * n * Z[(n - 1, 0)]
            elif n % 2 != 0 and m == (n - 1) / 2:  # :112-145
                z = (y * Z[(n - 1, n - 1 - m)] + x * Z[(n - 1, m - 1)] - y * Z[(n - 1, n - m)] - Z[(n - 2, m - 1)])
                gx = n * Z[(n - 1, m - 1)] + GX[(n - 2, m - 1)]
                gy = n * Z[(n - 1, n - 1 - m)] - n * Z[(n - 1, n - m)] + GY[(n - 2, m - 1)]
            elif n % 2 != 0 and m == (n - 1) / 2 + 1:  # :147-177
                z = (x * Z[(n - 1, m)] + y * Z[(n - 1, n - 1 - m)] + x * Z[(n - 1, m - 1)] - Z[(n - 2, m - 1)])
                gx = n * Z[(n - 1, m)] + n * Z[(n - 1, m - 1)] + GX[(n - 2, m - 1)]
                gy = n * Z[(n - 1, n - 1 - m)] + GY[(n - 2, m - 1)]
            elif n % 2 == 0 and m == n / 2:  # :179-209
                z = 2.0 * x * Z[(n - 1, m)] + 2.0 * y * Z[(n - 1, m - 1)] - Z[(n - 2, m - 1)]
                gx = 2.0 * n * Z[(n - 1, m)] + GX[(n - 2, m - 1)]
                gy = 2.0 * n * Z[(n - 1, n - 1 - m)] + GY[(n - 2, m - 1)]
            else:  # :211-246
                z = (x * Z[(n - 1, m)] + y * Z[(n - 1, n - 1 - m)] + x * Z[(n - 1, m - 1)]
                     - y * Z[(n - 1, n - m)] - Z[(n - 2, m - 1)])
                gx = n * Z[(n - 1, m)] + n * Z[(n - 1, m - 1)] + GX[(n - 2, m - 1)]
                gy = n * Z[(n - 1, n - 1 - m)] - n * Z[(n - 1, n - m)] + GY[(n - 2, m - 1)]
            Z[(n, m)], GX[(n, m)], GY[(n, m)] = z, gx, gy
    return Z, GX, GY


def zernike_normal(D: ZernikeDefect, P):
    """ART/ModuleDefects.py:156-166; P (n,3) relative to the mirror centre.  Returns (n,3) un-normalised."""
    x = P[:, 0] / D.R
    y = P[:, 1] / D.R
    max_order = max(k[0] for k in D.coeffs)
    _, GX, GY = zernike_tables(x, y, max_order)
    dX = np.zeros_like(x)
    dY = np.zeros_like(x)
    for k, c in D.coeffs.items():
        dX = dX + c * GX[k]
        dY = dY + c * GY[k]
    dX = dX / D.R
    dY = dY / D.R
    return np.stack([-dX, -dY, np.ones_like(x)], axis=1)


def zernike_offset(D: ZernikeDefect, P):
    """ART/ModuleDefects.py:168-174."""
    x = P[:, 0] / D.R
    y = P[:, 1] / D.R
    max_order = max(k[0] for k in D.coeffs)
    Z, _, _ = zernike_tables(x, y, max_order)
    h = np.zeros_like(x)
    for k, c in D.coeffs.items():
        h = h + c * Z[k]
    return h


def normal_add(N1, N2):
    """ART/ModuleGeometry.py:394-407 row-wise."""
    n1 = normalize_rows(N1)
    n2 = normalize_rows(N2)
    gX = -n1[:, 0] / n1[:, 2] + -n2[:, 0] / n2[:, 2]
    gY = -n1[:, 1] / n1[:, 2] + -n2[:, 1] / n2[:, 2]
    return np.stack([-gX, -gY, np.ones_like(gX)], axis=1)


# =============================================================================== ModuleMirror / ModuleMask
def base_normal(O: Optic, P):
    """get_normal of the undeformed mirror; P (n,3) in the optic frame."""
    k, q = O.kind, O.params
    if k in ("plane", "mask"):
        return np.tile(np.array([0.0, 0.0, 1.0]), (P.shape[0], 1))  # ModuleMirror.py:84-87, ModuleMask.py:63-66
    if k == "sphere":
        return normalize_rows(-P)  # :180-183
    if k == "parabola":  # :349-355
        return normalize_rows(np.stack([-P[:, 0], -P[:, 1], np.full(P.shape[0], q["p"])], axis=1))
    if k == "torus":  # :480-498
        x, y, z = P[:, 0], P[:, 1], P[:, 2]
        R, r = q["R"], q["r"]
        A = R ** 2 - r ** 2
        gx = 4 * (x ** 3 + x * y ** 2 + x * z ** 2 + x * A) - 8 * x * R ** 2
        gy = 4 * (y ** 3 + y * x ** 2 + y * z ** 2 + y * A)
        gz = 4 * (z ** 3 + z * x ** 2 + z * y ** 2 + z * A) - 8 * z * R ** 2
        return normalize_rows(-np.stack([gx, gy, gz], axis=1))
    if k == "ellipsoid":  # :685-693
        a, b = q["a"], q["b"]
        return normalize_rows(np.stack([-P[:, 0] / a ** 2, -P[:, 1] / b ** 2, -P[:, 2] / b ** 2], axis=1))
    if k == "cylinder":  # :846-849
        return normalize_rows(np.stack([np.zeros(P.shape[0]), -P[:, 1], -P[:, 2]], axis=1))
    raise ValueError(k)


def deformed_normal(O: Optic, P):
    """DeformedMirror.get_normal, ART/ModuleMirror.py:952-961."""
    n = base_normal(O, P)
    C = O.centre()
    for D in O.defects:
        n = normal_add(n, zernike_normal(D, P - C))
        n = n / norm_rows(n)[:, None]
    return n


def base_intersection(O: Optic, A, u):
    """`_get_intersection` of the undeformed optic.  Returns (hit mask (n,), P (n,3))."""
    k, q, S = O.kind, O.params, O.support
    n = A.shape[0]
    if k in ("plane", "mask"):
        # ModuleMirror.py:73-82 ; ModuleMask.py:51-61 (mask passes where the support is NOT hit)
        with np.errstate(divide="ignore", invalid="ignore"):
            t = -A[:, 2] / u[:, 2]
            I = u * t[:, None] + A
            inside = include_support(S, I[:, 0], I[:, 1])
            hit = (t > 0) & (inside if k == "plane" else ~inside)
        return hit, I
    ux, uy, uz = u[:, 0], u[:, 1], u[:, 2]
    xA, yA, zA = A[:, 0], A[:, 1], A[:, 2]
    if k == "sphere":  # :163-178
        co = np.stack([np.einsum("ij,ij->i", u, u), 2 * np.einsum("ij,ij->i", u, A),
                       np.einsum("ij,ij->i", A, A) - q["R"] ** 2], axis=1)
    elif k == "parabola":  # :325-347
        co = np.stack([ux ** 2 + uy ** 2, 2 * (ux * xA + uy * yA) - 2 * q["p"] * uz,
                       xA ** 2 + yA ** 2 - 2 * q["p"] * zA], axis=1)
    elif k == "ellipsoid":  # :662-683
        a, b = q["a"], q["b"]
        co = np.stack([(uy ** 2 + uz ** 2) / b ** 2 + (ux / a) ** 2,
                       2 * ((uy * yA + uz * zA) / b ** 2 + (ux * xA) / a ** 2),
                       (yA ** 2 + zA ** 2) / b ** 2 + (xA / a) ** 2 - 1], axis=1)
    elif k == "cylinder":  # :824-844
        co = np.stack([uy ** 2 + uz ** 2, 2 * (uy * yA + uz * zA), yA ** 2 + zA ** 2 - q["R"] ** 2], axis=1)
    elif k == "torus":  # :443-478
        R, r = q["R"], q["r"]
        G = 4.0 * R ** 2 * (ux ** 2 + uz ** 2)
        H = 8.0 * R ** 2 * (ux * xA + uz * zA)
        I_ = 4.0 * R ** 2 * (xA ** 2 + zA ** 2)
        J = np.einsum("ij,ij->i", u, u)
        K = 2.0 * np.einsum("ij,ij->i", u, A)
        L = np.einsum("ij,ij->i", A, A) + R ** 2 - r ** 2
        co = np.stack([J ** 2, 2 * J * K, 2 * J * L + K ** 2 - G, 2 * K * L - H, L ** 2 - I_], axis=1)
    else:
        raise ValueError(k)
    roots = np_roots_batch(co)
    t, valid = real_positive_roots(roots)
    deg = t.shape[1]
    cand = u[:, None, :] * np.where(valid, t, 0.0)[:, :, None] + A[:, None, :]       # (n,deg,3)
    C = O.centre()
    if k == "sphere" or k == "cylinder":
        ok = (cand[:, :, 2] < 0) & include_support(S, cand[:, :, 0], cand[:, :, 1])
    elif k == "parabola":
        ok = include_support(S, cand[:, :, 0] - C[0], cand[:, :, 1] - C[1])
    elif k == "ellipsoid":
        ok = (cand[:, :, 2] < 0) & include_support(S, cand[:, :, 0] - C[0], cand[:, :, 1] - C[1])
    else:  # torus
        ok = (cand[:, :, 2] < -q["R"]) & include_support(S, cand[:, :, 0], cand[:, :, 1])
    ok &= valid
    cnt = ok.sum(axis=1)
    # _IntersectionRayMirror, ModuleMirror.py:27-38: 1 -> it; 2 -> ClosestPoint (first if strictly closer,
    # else second; ModuleGeometry.py:138-147), in the order np.roots returned them; otherwise None.
    order = np.argsort(~ok, axis=1, kind="stable")          # accepted candidates first, original order kept
    first = np.take_along_axis(cand, order[:, 0][:, None, None].repeat(3, 2), axis=1)[:, 0, :]
    second = np.take_along_axis(cand, order[:, min(1, deg - 1)][:, None, None].repeat(3, 2), axis=1)[:, 0, :]
    d1 = np.einsum("ij,ij->i", first - A, first - A)
    d2 = np.einsum("ij,ij->i", second - A, second - A)
    P = np.where(((cnt == 1) | ((cnt == 2) & (d1 < d2)))[:, None], first, second)
    hit = (cnt == 1) | (cnt == 2)
    return hit, P


def intersection(O: Optic, A, u):
    """`_get_intersection` incl. DeformedMirror (ART/ModuleMirror.py:969-980)."""
    hit, P = base_intersection(O, A, u)
    if (O.defects or O.grids) and hit.any():
        C = O.centre()
        Ph = P[hit]
        h = np.zeros(Ph.shape[0])
        for D in O.defects:
            h = h + zernike_offset(D, Ph - C)
        for G in O.grids:
            h = h + G.offset(Ph - C)
        alpha = angle_between(-u[hit], base_normal(O, Ph))
        P = P.copy()
        P[hit] = Ph - u[hit] * (h / np.cos(alpha))[:, None]
    return hit, P


def reflect_bundle(O: Optic, B: Bundle, IgnoreDefects: bool) -> Bundle:
    """ReflectionMirrorRayList + _ReflectionMirrorRay, ART/ModuleMirror.py:878-939."""
    hit, P = intersection(O, B.point, B.vector)
    Bh = B.select(hit)
    P = P[hit]
    if O.defects and not IgnoreDefects:
        nrm = deformed_normal(O, P)
    else:
        nrm = base_normal(O, P)
    v = Bh.vector
    # SymmetricalVector(-v, n) = RotationAroundAxis(n, pi, -v), ModuleGeometry.py:272-276 (per ray axis)
    refl = np.empty_like(v)
    for i in range(v.shape[0]):
        refl[i] = rotation_around_axis(nrm[i], np.pi, -v[i])
    refl = normalize_rows(refl) if len(refl) else refl            # Ray.vector setter
    inc = angle_between(-v, nrm) if len(v) else np.zeros(0)
    seg = norm_rows(P - Bh.point)
    return Bundle(P, refl, Bh.number, np.concatenate([Bh.path, seg[:, None]], axis=1), inc, Bh.intensity,
                  Bh.wavelength)


def reflect_bundle_fast(O: Optic, B: Bundle, IgnoreDefects: bool) -> Bundle:
    """Same as reflect_bundle with the per-ray quaternion sandwich written out vectorised over rays
    (identical arithmetic order; used for the timed CPU baseline and large parity cases)."""
    hit, P = intersection(O, B.point, B.vector)
    Bh = B.select(hit)
    P = P[hit]
    nrm = deformed_normal(O, P) if (O.defects and not IgnoreDefects) else base_normal(O, P)
    v = Bh.vector
    ax = nrm / norm_rows(nrm)[:, None]
    aa = (np.pi * 0.5) * ax
    vn = norm_rows(aa)
    s = np.sin(vn) / vn
    qw, qx, qy, qz = np.cos(vn), s * aa[:, 0], s * aa[:, 1], s * aa[:, 2]
    x, y, z = -v[:, 0], -v[:, 1], -v[:, 2]
    tw = -qx * x - qy * y - qz * z
    tx = qw * x + qy * z - qz * y
    ty = qw * y - qx * z + qz * x
    tz = qw * z + qx * y - qy * x
    cx, cy, cz = -qx, -qy, -qz
    rx = tw * cx + tx * qw + ty * cz - tz * cy
    ry = tw * cy - tx * cz + ty * qw + tz * cx
    rz = tw * cz + tx * cy - ty * cx + tz * qw
    refl = normalize_rows(np.stack([rx, ry, rz], axis=1)) if len(v) else v
    inc = angle_between(-v, nrm) if len(v) else np.zeros(0)
    seg = norm_rows(P - Bh.point)
    return Bundle(P, refl, Bh.number, np.concatenate([Bh.path, seg[:, None]], axis=1), inc, Bh.intensity,
                  Bh.wavelength)


def transmit_mask(O: Optic, B: Bundle) -> Bundle:
    """TransmitMaskRayList + _TransmitMaskRay, ART/ModuleMask.py:93-136."""
    hit, I = base_intersection(O, B.point, B.vector)
    Bh = B.select(hit)
    I = I[hit]
    inc = angle_between(Bh.vector, EZ[None, :]) if len(Bh) else np.zeros(0)
    seg = norm_rows(I - Bh.point)
    return Bundle(I, Bh.vector, Bh.number, np.concatenate([Bh.path, seg[:, None]], axis=1), inc, Bh.intensity,
                  Bh.wavelength)


# =============================================================================== ModuleProcessing.RayTracingCalculation
def ray_tracing_calculation(source: Bundle, elements: Sequence[Element], IgnoreDefects=True, fast=True):
    """ART/ModuleProcessing.py:250-313.  Returns one Bundle per element (survivors, source order)."""
    out: List[Bundle] = []
    reflect = reflect_bundle_fast if fast else reflect_bundle
    for k, E in enumerate(elements):
        B = source if k == 0 else out[k - 1]
        Position = np.asarray(E.position, dtype=np.float64)
        n = np.asarray(E.normal, dtype=np.float64)
        m = np.asarray(E.majoraxis, dtype=np.float64)
        C = E.optic.centre()
        # lab -> optic frame (:289-295)
        pt = B.point + (-Position)
        pt, vec = rotation_rays(pt, B.vector, n, EZ)
        mPrime = rotation_point(m, n, EZ)
        pt, vec = rotation_rays(pt, vec, mPrime, EX)
        pt = pt + C
        Bo = Bundle(pt, vec, B.number, B.path, B.incidence, B.intensity, B.wavelength)
        # act (:298-303)
        if "Mirror" in E.optic.type or (E.optic.type == "" and E.optic.is_mirror()):
            Bo = reflect(E.optic, Bo, IgnoreDefects)
        elif E.optic.type == "Mask" or E.optic.kind == "mask":
            Bo = transmit_mask(E.optic, Bo)
        else:
            raise NameError("I don`t recognize the type of optical element " + E.optic.type + ".")
        # optic -> lab frame (:306-309)
        pt = Bo.point + (-C)
        if len(Bo):
            pt, vec = rotation_rays(pt, Bo.vector, EX, mPrime)
            pt, vec = rotation_rays(pt, vec, EZ, n)
        else:
            vec = Bo.vector
        pt = pt + Position
        out.append(Bundle(pt, vec, Bo.number, Bo.path, Bo.incidence, Bo.intensity, Bo.wavelength))
    return out


# =============================================================================== ModuleDetector
@dataclass
class Detector:
    centre: np.ndarray
    normal: np.ndarray
    refpoint: np.ndarray


def find_central_ray(B: Bundle):
    """ART/ModuleProcessing.py:464-482 (np.mean over lists)."""
    return np.mean(B.point, axis=0), normalize_rows(np.mean(B.vector, axis=0)[None, :])[0]


def detector_autoplace(B: Bundle, distance: float) -> Detector:
    """ART/ModuleDetector.py:109-137."""
    cp, cv = find_central_ray(B)
    normal = -cv
    normal = normal / np.linalg.norm(normal)
    return Detector(cp - normal * distance, normal, cp)


def detector_distance(D: Detector) -> float:
    """ART/ModuleDetector.py:139-145."""
    I = intersection_line_plane(D.refpoint[None, :], -D.normal[None, :], D.centre, D.normal)[0]
    return float(np.linalg.norm(D.refpoint - I))


def intersection_line_plane(A, u, P, n):
    """ART/ModuleGeometry.py:48-57 row-wise."""
    t = ((-A + P) @ n) / (u @ n)
    return u * t[:, None] + A


def detector_points3d(D: Detector, B: Bundle):
    """ART/ModuleDetector.py:191-210."""
    return intersection_line_plane(B.point, B.vector, D.centre, D.normal)


def detector_points2d(D: Detector, B: Bundle):
    """ART/ModuleDetector.py:212-234."""
    P = detector_points3d(D, B) - D.centre
    P = rotation_point(P, D.normal, EZ)
    return P[:, 0:2]


def centre_point_list(P2):
    """ART/ModuleGeometry.py:222-245."""
    cx = (np.amax(P2[:, 0]) + np.amin(P2[:, 0])) * 0.5
    cy = (np.amax(P2[:, 1]) + np.amin(P2[:, 1])) * 0.5
    return P2 - np.array([cx, cy])


def detector_points2dcentre(D: Detector, B: Bundle):
    """ART/ModuleDetector.py:236-252."""
    return centre_point_list(detector_points2d(D, B))


def optical_paths(D: Detector, B: Bundle):
    """ART/ModuleDetector.py:272-275."""
    I = detector_points3d(D, B)
    return norm_rows(B.point - I) + np.sum(B.path, axis=1)


def detector_delays(D: Detector, B: Bundle):
    """ART/ModuleDetector.py:254-279 (fs)."""
    paths = optical_paths(D, B)
    return (paths - np.mean(paths)) / LightSpeed * 1e15


def standard_deviation(v):
    """ART/ModuleProcessing.py:485-507."""
    v = np.asarray(v)
    if v.ndim == 1:
        return float(np.std(v))
    return float(np.sqrt(np.var(v, axis=0).sum()))


def weighted_standard_deviation(v, w):
    """ART/ModuleProcessing.py:510-532."""
    v = np.asarray(v)
    average = np.average(v, axis=0, weights=w)
    variance = np.average((v - average) ** 2, axis=0, weights=w)
    return float(np.sqrt(np.sum(variance)))


# =============================================================================== analysis behind the trace
def e_transmission(B_in: Bundle, B_out: Bundle) -> float:
    """ART/ModuleAnalysisAndPlots.py:62-77 (percent)."""
    return float(100 * np.sum(B_out.intensity) / np.sum(B_in.intensity))


def numerical_aperture(B: Bundle, refractive_index: float = 1) -> float:
    """ART/ModuleProcessing.py:536-566: sin of the largest angle between any ray and the central ray's vector."""
    _, cv = find_central_ray(B)
    return float(np.sin(np.amax(angle_between(np.broadcast_to(cv, B.vector.shape), B.vector))) * refractive_index)


def result_summary(D: Detector, B: Bundle):
    """ART/ModuleAnalysisAndPlots.py:81-129: (spot-size SD in mm, duration SD in fs) at the detector as it stands."""
    return standard_deviation(detector_points2dcentre(D, B)), standard_deviation(detector_delays(D, B))


def detector_shifted(D: Detector, shift: float) -> Detector:
    """ART/ModuleDetector.py:163-177 (shiftByDistance: centre - shift * normal; refpoint and normal stay)."""
    return Detector(D.centre - shift * D.normal, D.normal, D.refpoint)


def _find_optimal_distance_bis(D: Detector, amplitude, step, B: Bundle, opt_for, weighted):
    """ART/ModuleProcessing.py:317-366: one scan level, POSITION BY POSITION as the reference loops (no linear model):
    n = int(2 A / Step) positions from -A in steps of Step; returns the detector at the best position."""
    sizes, durations, fitness = [], [], []
    w = B.intensity if weighted else None
    D = detector_shifted(D, -amplitude)
    n = int(2 * amplitude / step)
    for _ in range(n):
        if opt_for in ("intensity", "spotsize"):
            P = detector_points2dcentre(D, B)
            size = weighted_standard_deviation(P, w) if weighted else standard_deviation(P)
            sizes.append(size)
        if opt_for in ("intensity", "duration"):
            dl = detector_delays(D, B)
            dur = weighted_standard_deviation(dl, w) if weighted else standard_deviation(dl)
            durations.append(dur)
        fitness.append(size ** 2 * dur if opt_for == "intensity" else (dur if opt_for == "duration" else size))
        D = detector_shifted(D, step)
    ind = fitness.index(min(fitness))
    D = detector_shifted(D, -(n - ind) * step)
    return (D, sizes[ind] if opt_for in ("intensity", "spotsize") else np.nan,
            durations[ind] if opt_for in ("intensity", "duration") else np.nan)


def find_optimal_distance(D: Detector, B: Bundle, opt_for="intensity", amplitude=None, precision=3, weighted=False):
    """ART/ModuleProcessing.py:369-460.  (Like the reference: "size" passes the name check but the scan only knows
    "spotsize" -- only "intensity" and "duration" work.)  -> (moved Detector, spot SD, duration SD)."""
    if opt_for not in ("intensity", "size", "duration"):
        raise NameError("OptFor must be either 'intensity', 'size' or 'duration'.")
    first = detector_distance(D)
    size = 2 * standard_deviation(detector_points2dcentre(D, B))
    na = numerical_aperture(B, 1)
    if amplitude is None:
        amplitude = min(4 * np.ceil(size / np.tan(np.arcsin(na))), first)
    step = amplitude / 10
    spot = dur = np.nan
    for k in range(precision + 1):
        D, spot, dur = _find_optimal_distance_bis(D, amplitude * 0.1 ** k, step * 0.1 ** k, B, opt_for, weighted)
    return D, spot, dur


def analysis_row(B: Bundle, D: Detector, co: float, axis=None):
    """What art_analyse_bundles (include/art_hip.h) reports for bundle B on detector D, restated from the REFERENCE's
    definitions only (no kernel code): the sums of FindCentralRay / getETransmission (ART/ModuleProcessing.py:464-482,
    ART/ModuleAnalysisAndPlots.py:62-77), the read-out of ART/ModuleDetector.py:191-279 at the detector and at the detector
    shifted by 1 mm (shiftByDistance, :163-177: every ray's read-out is linear in the shift, so the difference IS the
    slope), the largest Kahan angle to `axis` (default: the mean vector; ART/ModuleProcessing.py:536-566), and the shifts
    at which a hit point passes through its ray's origin (t(s) = n.(C - s n - A) / (u.n) = 0).  `co`: the provisional
    centre of the path moments -- a free parameter of the sums, taken from the caller.  -> dict of row slots."""
    w = B.intensity if B.intensity is not None else np.ones(len(B))
    P0 = detector_points2d(D, B)
    O0 = optical_paths(D, B)
    D1 = detector_shifted(D, 1.0)
    P1 = detector_points2d(D1, B)
    O1 = optical_paths(D1, B)
    q = [P0[:, 0], P0[:, 1], O0 - co]
    sq = [P1[:, 0] - P0[:, 0], P1[:, 1] - P0[:, 1], (O1 - O0) - 1.0]
    mom = np.zeros(32)
    for base, wt in ((0, np.ones(len(B))), (16, w)):
        mom[base] = wt.sum()
        for k in range(3):
            o = base + 1 + 5 * k
            mom[o:o + 5] = [(wt * q[k]).sum(), (wt * sq[k]).sum(), (wt * q[k] * q[k]).sum(), (wt * q[k] * sq[k]).sum(),
                            (wt * sq[k] * sq[k]).sum()]
    n = D.normal
    sk = ((D.centre - B.point) @ n) / (n @ n)
    if axis is None:
        axis = find_central_ray(B)[1]
    return {"count": float(len(B)), "sum_point": B.point.sum(axis=0), "sum_vector": B.vector.sum(axis=0), "sum_w": float(w.sum()),
            "sum_path": float(np.sum(B.path)), "moments": mom,
            "kink_below": sk[sk <= 0].max() if (sk <= 0).any() else -np.inf,
            "kink_above": sk[sk > 0].min() if (sk > 0).any() else np.inf,
            "max_angle": float(np.amax(angle_between(np.broadcast_to(axis, B.vector.shape), B.vector))),
            "bbox": np.array([P0[:, 0].min(), P0[:, 0].max(), P0[:, 1].min(), P0[:, 1].max(), O0.min(), O0.max()]),
            "mean_opl": float(O0.mean())}


# =============================================================================== ModuleSource
def spiral_vogel(n, radius):
    """ART/ModuleGeometry.py:61-76."""
    golden = np.pi * (3 - np.sqrt(5))
    r = np.sqrt(np.arange(n) / n) * radius
    theta = golden * np.arange(n)
    M = np.zeros((n, 2))
    M[:, 0] = np.cos(theta)
    M[:, 1] = np.sin(theta)
    return M * r.reshape((n, 1))


def point_source(S, axis, divergence, n, wavelength=None) -> Bundle:
    """ART/ModuleSource.py:23-81."""
    M = spiral_vogel(n, 1 * np.tan(divergence))
    vec = normalize_rows(np.stack([M[:, 0], M[:, 1], np.ones(n)], axis=1))
    pt = np.zeros((n, 3))
    pt, vec = rotation_rays(pt, vec, EZ, np.asarray(axis, float))
    pt = pt + np.asarray(S, float)
    return make_bundle(pt, vec, np.arange(n), None, wavelength)


def plane_wave_disk(centre, axis, radius, n, wavelength=None) -> Bundle:
    """ART/ModuleSource.py:135-169 (emits n-1 rays)."""
    M = spiral_vogel(n, radius)[: n - 1]
    pt = np.stack([M[:, 0], M[:, 1], np.zeros(n - 1)], axis=1)
    vec = np.tile(EZ, (n - 1, 1))
    pt, vec = rotation_rays(pt, vec, EZ, np.asarray(axis, float))
    pt = pt + np.asarray(centre, float)
    return make_bundle(pt, vec, np.arange(n - 1), None, wavelength)


def apply_gaussian_intensity(B: Bundle, fraction=1 / np.e ** 2) -> Bundle:
    """ART/ModuleSource.py:219-261."""
    _, axis = find_central_ray(B)
    ang = angle_between(axis[None, :], B.vector)
    div = max(0.0, float(np.max(ang)))
    if div > 1e-12:
        inten = np.exp(-2 * (np.tan(ang) / div) ** 2 * -0.5 * np.log(fraction))
    else:
        d = norm_rows(B.point)
        inten = np.exp(-2 * (d / np.max(d)) ** 2 * -0.5 * np.log(fraction))
    return Bundle(B.point, B.vector, B.number, B.path, B.incidence, inten, B.wavelength)


# =============================================================================== fixture glue
def optic_from_desc(d, arrays=None) -> Optic:
    S = Support(d["support"]["kind"], list(d["support"]["p"]))
    params = {k: d[k] for k in ("R", "r", "feff", "offaxis_rad", "p", "a", "b") if k in d}
    defects = [ZernikeDefect({(int(c[0]), int(c[1])): float(c[2]) for c in z["coeffs"]}, float(z["R"]))
               for z in d.get("defects", []) if z["kind"] == "zernike"]
    rect = [2 * S.p[0], 2 * S.p[0]] if S.kind in ("round", "roundhole") else [S.p[0], S.p[1]]   # _CircumRect
    grids = [GridDefect(arrays[z["map"]], rect) for z in d.get("defects", []) if z["kind"] == "fourrier"]
    return Optic(d["kind"], S, params, defects, d.get("type", ""), grids)


def elements_from_scene(scene, arrays=None) -> List[Element]:
    return [Element(optic_from_desc(e, arrays), np.array(e["position"], float), np.array(e["normal"], float),
                    np.array(e["majoraxis"], float)) for e in scene["elements"]]
