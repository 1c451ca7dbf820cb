"""Host-side geometry helpers with the reference's names and semantics (ART/ModuleGeometry.py).

Everything here is small host NumPy work on 3-vectors, point lists and element poses: scene construction,
detector placement and the constant 3x3 frame maps handed to the HIP kernels.  Nothing in this module
loops over a ray bundle -- bundles live on the GPU (see bundle.py, ModuleProcessing.RayTracingCalculation).
"""
import math

import numpy as np


def _norm(v):
    """np.linalg.norm of a real float array -- sqrt(x . x), the same two operations, so the same bits -- without its
    argument handling (5 us per call; placing a loop list normalises 300 vectors)."""
    v = v.ravel()
    if v.dtype != np.float64:
        v = v.astype(float)          # (as np.linalg.norm does: integer vectors must not overflow in the dot product)
    return math.sqrt(v.dot(v))


def Normalize(Vector):
    """Unit vector (ART/ModuleGeometry.py:17-19)."""
    Vector = np.asarray(Vector, dtype=float)
    return Vector / _norm(Vector)


def VectorPerpendicular(Vector):
    """Some vector perpendicular to `Vector` (ART/ModuleGeometry.py:23-36)."""
    for i, e in enumerate(np.eye(3, dtype=int)):
        if abs(Vector[i]) < 1e-15:
            return e
    return Normalize(np.array([1, 1, -1.0 * (Vector[0] + Vector[1]) / Vector[2]]))


def AngleBetweenTwoVectors(U, V):
    """Angle in rad, W. Kahan's formula (ART/ModuleGeometry.py:40-44)."""
    U = np.asarray(U, dtype=float)
    V = np.asarray(V, dtype=float)
    u = _norm(U)
    v = _norm(V)
    return 2 * np.arctan2(_norm(U * v - V * u), _norm(U * v + V * u))


def IntersectionLinePlane(A, u, P, n):
    """Point where the line A + t u pierces the plane through P with normal n (ART/ModuleGeometry.py:48-57)."""
    t = np.dot(n, -A + P) / np.dot(u, n)
    return u * t + A


def SpiralVogel(NbPoint, Radius):
    """NbPoint x 2 points of Vogel's spiral (ART/ModuleGeometry.py:61-76)."""
    k = np.arange(NbPoint)
    theta = np.pi * (3 - np.sqrt(5)) * k
    r = np.sqrt(k / NbPoint) * Radius
    M = np.zeros((NbPoint, 2))
    M[:, 0] = np.cos(theta)
    M[:, 1] = np.sin(theta)
    return M * r.reshape((NbPoint, 1))


def _real_roots(coeffs):
    return [s.real for s in np.roots(coeffs) if abs(s.imag) < 1e-15]


def SolverQuadratic(a, b, c):
    """Real solutions of a x^2 + b x + c = 0 (ART/ModuleGeometry.py:80-91). Host utility; the kernels use a closed form."""
    return _real_roots([a, b, c])


def SolverQuartic(a, b, c, d, e):
    """Real solutions of the quartic (ART/ModuleGeometry.py:95-106). Host utility only."""
    return _real_roots([a, b, c, d, e])


def KeepPositiveSolution(SolutionList):
    """ART/ModuleGeometry.py:110-120."""
    return [k for k in SolutionList if k > 1e-12]


def KeepNegativeSolution(SolutionList):
    """ART/ModuleGeometry.py:124-134."""
    return [k for k in SolutionList if k < -1e-12]


def ClosestPoint(A, I1, I2):
    """ART/ModuleGeometry.py:138-147."""
    return I1 if np.dot(I1 - A, I1 - A) < np.dot(I2 - A, I2 - A) else I2


def FarestPoint(A, I1, I2):
    """ART/ModuleGeometry.py:151-160."""
    return I1 if np.dot(I1 - A, I1 - A) > np.dot(I2 - A, I2 - A) else I2


def DiameterPointList(PointList):
    """Largest extent of the bounding box of 2D/3D points (ART/ModuleGeometry.py:164-218)."""
    if len(PointList) == 0:
        return None
    P = np.asarray(PointList, dtype=float)
    return float(np.max(np.abs(P.max(axis=0) - P.min(axis=0))))


def CentrePointList(PointList):
    """Shift 2D points so that their bounding box is centred on the origin (ART/ModuleGeometry.py:222-245)."""
    P = np.asarray(PointList, dtype=float)
    c = (P.max(axis=0) + P.min(axis=0)) * 0.5
    return list(P[:, :2] - c[:2])


def IncludeRectangle(X, Y, Point):
    """ART/ModuleGeometry.py:249-255."""
    return bool(abs(Point[0]) <= abs(X / 2) and abs(Point[1]) <= abs(Y / 2))


def IncludeDisk(R, Point):
    """ART/ModuleGeometry.py:259-268."""
    return bool((Point[0] ** 2 + Point[1] ** 2) <= R ** 2)


# ------------------------------------------------------------------------------------------------- hash epochs
# An OpticalElement's hash covers its pose and every parameter of its optic (like the reference's: it IS the cache key of
# OpticalChain.get_output_rays) and is recomputed from the contents on every call -- arrays may be modified in place.  One
# call of the host shell (trace_chain_list, analyse_chain_list, OEPlacement) asks for the hash of the same unchanged element
# five to ten times (cache keys, descriptors, lazy-history scene keys): inside `with frozen_hashes():` -- code of this
# package that does not modify an element between its first and last look at it -- the first result is reused.
_HASH_MEMO = None


class frozen_hashes:
    """Context manager: element hashes computed inside it are memoised by object identity until it exits (re-entrant)."""

    def __enter__(self):
        global _HASH_MEMO
        self._outer = _HASH_MEMO
        if _HASH_MEMO is None:
            _HASH_MEMO = {}
        return self

    def __exit__(self, *exc):
        global _HASH_MEMO
        _HASH_MEMO = self._outer
        return False


def memo_hash(obj, compute):
    """hash of `obj` through the current epoch's memo (compute() when there is none, or on first sight).  The memo holds
    the object itself: an id() cannot be recycled while the epoch lasts."""
    memo = _HASH_MEMO
    if memo is None:
        return compute()
    hit = memo.get(id(obj))
    if hit is None:
        hit = memo[id(obj)] = (compute(), obj)
    return hit[0]


# ------------------------------------------------------------------------------------------------- copies
_SCALARS = (int, float, str, bool, type(None), np.floating, np.integer)


def flat_deepcopy(obj, memo):
    """`__deepcopy__` of the small value classes of the host shell (supports, mirrors, masks, optical elements): scalars by
    value, arrays by `.copy()`, anything else through `copy.deepcopy` with the caller's memo -- the same result as the
    generic machinery (`__reduce_ex__` + `_reconstruct`) at a fifth of its cost; OpticalChain deep-copies its elements on
    construction, as the reference does, and a loop list builds ten chains."""
    import copy
    new = obj.__class__.__new__(obj.__class__)
    memo[id(obj)] = new
    d = new.__dict__
    for k, v in obj.__dict__.items():
        if isinstance(v, _SCALARS):
            d[k] = v
        elif type(v) is np.ndarray:
            d[k] = v.copy()
        else:
            d[k] = copy.deepcopy(v, memo)
    return new


# ------------------------------------------------------------------------------------------------- rotations
def _cross3(a, b):
    """np.cross for two 3-vectors, component by component (the same products and differences, so the same bits):
    np.cross spends ~40 us per call on axis bookkeeping, and building the frame maps of one optical element needs 28
    cross products -- that was 0.3 ms per element, i.e. the whole host cost of tracing a loop list."""
    return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]])


def RotationAroundAxis(Axis, Angle, Vector):
    """Rotate Vector by Angle (rad) about Axis (ART/ModuleGeometry.py:321-329, there via a unit quaternion).
    Written as the equivalent Rodrigues sum with half-angle terms: v + 2w(a x v) + 2 a x (a x v),
    a = sin(Angle/2) * axis, w = cos(Angle/2)."""
    k = Normalize(Axis)
    v = np.asarray(Vector, dtype=float)
    a = np.sin(0.5 * Angle) * k
    w = np.cos(0.5 * Angle)
    av = _cross3(a, v)
    return v + 2.0 * (w * av + _cross3(a, av))


def rotation_matrix(Axis1, Axis2):
    """3x3 matrix of the map RotationPoint(., Axis1, Axis2), including its two special cases
    (ART/ModuleGeometry.py:333-343): identity for parallel axes and the point inversion -I for antiparallel ones."""
    ang = AngleBetweenTwoVectors(Axis1, Axis2)
    if abs(ang) < 1e-10:
        return np.eye(3)
    if abs(ang - np.pi) < 1e-10:
        return -np.eye(3)
    N = _cross3(np.asarray(Axis1, dtype=float), np.asarray(Axis2, dtype=float))
    return np.stack([RotationAroundAxis(N, ang, e) for e in np.eye(3)], axis=1)


def RotationPoint(Point, Axis1, Axis2):
    """Map Point such that Axis1 becomes Axis2 (ART/ModuleGeometry.py:333-343)."""
    return rotation_matrix(Axis1, Axis2) @ np.asarray(Point, dtype=float)


def RotationPointList(PointList, Axis1, Axis2):
    """ART/ModuleGeometry.py:347-353."""
    M = rotation_matrix(Axis1, Axis2)
    return [M @ np.asarray(p, dtype=float) for p in PointList]


def SymmetricalVector(V, SymmetryAxis):
    """ART/ModuleGeometry.py:272-276."""
    return RotationAroundAxis(SymmetryAxis, np.pi, V)


def TranslationPoint(Point, T):
    return Point + T


def TranslationPointList(PointList, T):
    return [p + T for p in PointList]


_FRAME_MAPS = {}


def frame_maps(normal, majoraxis):
    """Constant lab->optic and optic->lab 3x3 maps of one optical element, composed exactly as
    RayTracingCalculation chains its per-ray rotations (ART/ModuleProcessing.py:289-294, :307-308):
        fwd = R(m' -> ex) . R(n -> ez),  m' = R(n -> ez) m        bwd = R(ez -> n) . R(ex -> m')
    Memoised on the exact bytes of the two axes: the chains of a loop list repeat most of their poses (read-only
    arrays are returned)."""
    n = np.asarray(normal, dtype=float)
    m = np.asarray(majoraxis, dtype=float)
    key = n.tobytes() + m.tobytes()
    hit = _FRAME_MAPS.get(key)
    if hit is not None:
        return hit
    ez = np.array([0.0, 0.0, 1.0])
    ex = np.array([1.0, 0.0, 0.0])
    R1 = rotation_matrix(n, ez)
    mPrime = R1 @ m
    R2 = rotation_matrix(mPrime, ex)
    B2 = rotation_matrix(ex, mPrime)
    B1 = rotation_matrix(ez, n)
    fwd, bwd = R2 @ R1, B1 @ B2
    fwd.setflags(write=False)
    bwd.setflags(write=False)
    if len(_FRAME_MAPS) > 4096:
        _FRAME_MAPS.clear()
    _FRAME_MAPS[key] = (fwd, bwd)
    return fwd, bwd


def normal_add(N1, N2):
    """Combine two surface normals by adding their slopes (ART/ModuleGeometry.py:394-407)."""
    n1 = Normalize(N1)
    n2 = Normalize(N2)
    gx = -n1[0] / n1[2] - n2[0] / n2[2]
    gy = -n1[1] / n1[2] - n2[1] / n2[2]
    return np.array([-gx, -gy, 1])


# ------------------------------------------------------------------------------------------------- ray lists
# The *RayList functions of the reference take lists of Ray objects.  Here they accept a RayBundle (device
# SoA) and return a new RayBundle; single Ray objects are handled on the host.
def TranslationRay(Ray, T):
    r = Ray.copy_ray()
    r.point = r.point + np.asarray(T, dtype=float)
    return r


def RotationRay(Ray, Axis1, Axis2):
    M = rotation_matrix(Axis1, Axis2)
    r = Ray.copy_ray()
    r.point = M @ Ray.point
    r.vector = M @ Ray.vector
    return r


def TranslationRayList(RayList, T):
    from .bundle import RayBundle
    if isinstance(RayList, RayBundle):
        return RayList.transformed(np.eye(3), np.asarray(T, dtype=float))
    return [TranslationRay(r, T) for r in RayList]


def RotationRayList(ListeRay, Axis1, Axis2):
    from .bundle import RayBundle
    M = rotation_matrix(Axis1, Axis2)
    if isinstance(ListeRay, RayBundle):
        return ListeRay.transformed(M, np.zeros(3))
    return [RotationRay(r, Axis1, Axis2) for r in ListeRay]


def RotationAroundAxisRayList(ListeRay, Axis, Angle):
    from .bundle import RayBundle
    M = np.stack([RotationAroundAxis(Axis, Angle, e) for e in np.eye(3)], axis=1)
    if isinstance(ListeRay, RayBundle):
        return ListeRay.transformed(M, np.zeros(3), rotate_points=False)
    out = []
    for r in ListeRay:
        q = r.copy_ray()
        q.vector = M @ r.vector
        out.append(q)
    return out
