"""Mirror surfaces, API of ART/ModuleMirror.py.

The classes are parameter holders for the HIP kernels: each exposes `_abi_kind` / `_abi_params()` (layout in
include/art_hip.h) plus the reference's host-side helpers `get_centre()` and `get_normal(Point)` for a single
point.  Ray/surface intersection, aperture test and reflection of a *bundle* never run in Python; see
ModuleProcessing.RayTracingCalculation -> libart_hip.so.  `get_grid3D` samples the surface for the 3-D render
(host, a few thousand points)."""
import math

import numpy as np

from . import _abi
from . import ModuleGeometry as mgeo


def _IntersectionRayMirror(PointMirror, ListPointIntersectionMirror):
    """Two candidate points -> the one closer to PointMirror, one -> it, otherwise None (ART/ModuleMirror.py:27-38).
    Host helper kept for API parity; the kernels apply the same rule per ray (art_device.h `consider`)."""
    if len(ListPointIntersectionMirror) == 2:
        return mgeo.ClosestPoint(PointMirror, ListPointIntersectionMirror[0], ListPointIntersectionMirror[1])
    if len(ListPointIntersectionMirror) == 1:
        return ListPointIntersectionMirror[0]
    return None


def _ReflectionMirrorRay(Mirror, PointMirror, Ray):
    """Reflect ONE ray at a given surface point (ART/ModuleMirror.py:878-906), on the host; bundles go through
    ReflectionMirrorRayList / RayTracingCalculation on the device."""
    NormalMirror = Mirror.get_normal(PointMirror)
    out = Ray.copy_ray()
    out.point = PointMirror
    out.vector = mgeo.SymmetricalVector(-Ray.vector, NormalMirror)
    out.incidence = mgeo.AngleBetweenTwoVectors(-Ray.vector, NormalMirror)
    out.path = Ray.path + (np.linalg.norm(PointMirror - Ray.point),)
    return out


class _Mirror:
    __deepcopy__ = mgeo.flat_deepcopy

    _abi_kind = None

    def _abi_params(self):
        raise NotImplementedError

    def _get_intersection(self, Ray):
        """Intersection point of ONE ray, given in the mirror's own frame, with the surface inside its support, or
        None (the `_get_intersection` of every mirror class of ART/ModuleMirror.py): a one-ray trace on the device."""
        hit = ReflectionMirrorRayList(self, [Ray], IgnoreDefects=True)
        return hit[0].point if len(hit) == 1 else None

    def _content_hash(self):
        return hash((self.type, hash(self.support)) + tuple(float(v) for v in self._abi_params()))

    def __hash__(self):
        return mgeo.memo_hash(self, self._content_hash)      # (recomputed from the contents, once per hash epoch)

    _cloud_has_outline = True      # the point cloud of get_grid3D starts with the outline of the support

    def _sag(self, x, y):
        """Surface height z(x, y) in the mirror's own frame (arrays; NaN where the surface does not exist)."""
        return np.zeros_like(x)

    def get_grid3D(self, NbPoint: int, **kwargs):
        """About NbPoint points of the surface inside its support, in the mirror's own frame: a tenth of them on the
        outline of the support (and of its hole), the rest on the support's grid (the `get_grid3D` of every optic
        class of ART/ModuleMirror.py, e.g. :93-113).  With edges=True also the closed index loops of the hole outlines.
        Points where the surface does not exist under the support are left out."""
        return _surface_cloud(self, self._sag, self._cloud_has_outline, NbPoint, bool(kwargs.get("edges")))


def _surface_cloud(optic, sag, with_outline, NbPoint, want_edges):
    n_outline = int(round(0.1 * NbPoint))
    outline, loops = optic.support._Contour_points(n_outline, edges=True)
    outline = np.asarray(outline, dtype=float).reshape(-1, 2)
    grid = optic.support._grid_xy(NbPoint - n_outline)
    centre = optic.get_centre()

    def lift(xy):
        x, y = xy[:, 0] + centre[0], xy[:, 1] + centre[1]
        with np.errstate(invalid="ignore"):
            z = sag(x, y)
        return np.column_stack((x, y, z)), np.isfinite(z)

    if with_outline:
        P, ok = lift(np.concatenate((outline, grid)))
        new_index = np.cumsum(ok) - 1
        loops = [[int(new_index[i]) for i in loop if ok[i]] for loop in loops]
        P = P[ok]
    else:
        # the ellipsoid's cloud (ART/ModuleMirror.py:716-751): grid points first, then only the outline points an edge
        # loop refers to (the hole), in loop order -- the loop's closing index is a point of its own
        G, okg = lift(grid)
        O, oko = lift(outline)
        parts, first, new_loops = [G[okg]], int(okg.sum()), []
        for loop in loops:
            keep = [i for i in loop if oko[i]]
            parts.append(O[keep])
            new_loops.append(list(range(first, first + len(keep))))
            first += len(keep)
        P, loops = np.concatenate(parts), new_loops
    cloud = list(P)
    return (cloud, loops) if want_edges else cloud


class MirrorPlane(_Mirror):
    """Plane mirror in the xy-plane of its own frame (ART/ModuleMirror.py:42-113)."""
    _abi_kind = _abi.ART_PLANE

    def __init__(self, Support):
        self.support = Support
        self.type = "Plane Mirror"

    def _abi_params(self):
        return []

    def get_normal(self, Point):
        return np.array([0, 0, 1])

    def get_centre(self):
        return np.array([0, 0, 0])


class MirrorSpherical(_Mirror):
    """Sphere x^2+y^2+z^2 = R^2; negative Radius = convex (ART/ModuleMirror.py:117-208)."""
    _abi_kind = _abi.ART_SPHERE

    def __init__(self, Radius, Support):
        if Radius < 0:
            self.type = "SphericalCX Mirror"
            self.radius = -Radius
        else:
            self.type = "SphericalCC Mirror"
            self.radius = Radius
        self.support = Support

    def _abi_params(self):
        return [self.radius]

    def get_normal(self, Point):
        return mgeo.Normalize(-np.asarray(Point, dtype=float))

    def get_centre(self):
        return np.array([0, 0, -self.radius])

    def _sag(self, x, y):
        return -np.sqrt(self.radius ** 2 - (x * x + y * y))


class MirrorParabolic(_Mirror):
    """Paraboloid x^2+y^2 = 2 p z with the support centre off-axis (ART/ModuleMirror.py:212-387).
    `offaxisangle` is given in degrees and stored/returned in radians, like the reference."""
    _abi_kind = _abi.ART_PARABOLA

    def __init__(self, FocalEffective: float, OffAxisAngle: float, Support):
        self._offaxisangle = np.deg2rad(OffAxisAngle)
        self.support = Support
        self.type = "Parabolic Mirror"
        self._feff = FocalEffective
        self._p = FocalEffective * (1 + np.cos(self._offaxisangle))

    @property
    def offaxisangle(self):
        return self._offaxisangle

    @offaxisangle.setter
    def offaxisangle(self, OffAxisAngle):
        self._offaxisangle = np.deg2rad(OffAxisAngle)
        self._p = self._feff * (1 + np.cos(self._offaxisangle))

    @property
    def feff(self):
        return self._feff

    @feff.setter
    def feff(self, FocalEffective):
        self._feff = FocalEffective
        self._p = self._feff * (1 + np.cos(self._offaxisangle))

    @property
    def p(self):
        return self._p

    @p.setter
    def p(self, SemiLatusRectum):
        self._p = SemiLatusRectum
        self._feff = self._p / (1 + np.cos(self._offaxisangle))

    def _abi_params(self):
        return [self._p]

    def get_normal(self, Point):
        return mgeo.Normalize(np.array([-Point[0], -Point[1], self._p]))

    def get_centre(self):
        return np.array([self.feff * np.sin(self.offaxisangle), 0,
                         self._p * 0.5 - self.feff * np.cos(self.offaxisangle)])

    def _sag(self, x, y):
        return (x * x + y * y) / (2 * self._p)


class MirrorToroidal(_Mirror):
    """Torus (sqrt(x^2+z^2) - R)^2 + y^2 = r^2 (ART/ModuleMirror.py:391-527)."""
    _abi_kind = _abi.ART_TORUS

    def __init__(self, MajorRadius, MinorRadius, Support):
        self.majorradius = MajorRadius
        self.minorradius = MinorRadius
        self.support = Support
        self.type = "Toroidal Mirror"

    def _abi_params(self):
        return [self.majorradius, self.minorradius]

    def get_normal(self, Point):
        x, y, z = Point
        R2, r2 = self.majorradius ** 2, self.minorradius ** 2
        S = x * x + y * y + z * z
        return mgeo.Normalize(-np.array([x * (S - R2 - r2), y * (S + R2 - r2), z * (S - R2 - r2)]))

    def get_centre(self):
        return np.array([0, 0, -self.majorradius - self.minorradius])

    def _sag(self, x, y):
        return -np.sqrt((np.sqrt(self.minorradius ** 2 - y * y) + self.majorradius) ** 2 - x * x)


def ReturnOptimalToroidalRadii(Focal: float, AngleIncidence: float):
    """Major/minor radii giving focal length `Focal` at `AngleIncidence` (deg) without astigmatism
    (ART/ModuleMirror.py:533-561)."""
    c = np.cos(AngleIncidence * np.pi / 180)
    return 2 * Focal * (1 / c - c), 2 * Focal * c


class MirrorEllipsoidal(_Mirror):
    """Ellipsoid (x/a)^2 + (y/b)^2 + (z/b)^2 = 1 (ART/ModuleMirror.py:565-751); same ctor variants."""
    _abi_kind = _abi.ART_ELLIPSOID

    def __init__(self, Support, SemiMajorAxis=None, SemiMinorAxis=None, OffAxisAngle=None, f_object=None,
                 f_image=None):
        self.type = "Ellipsoidal Mirror"
        self.support = Support
        self.a = None
        self.b = None
        self._offaxisangle = None
        if SemiMajorAxis is not None and SemiMinorAxis is not None:
            self.a = SemiMajorAxis
            self.b = SemiMinorAxis
        have_f = f_object is not None and f_image is not None
        if OffAxisAngle is not None:
            self._offaxisangle = np.deg2rad(OffAxisAngle)
            if have_f:
                foci_sq = f_object ** 2 + f_image ** 2 - 2 * f_object * f_image * np.cos(self._offaxisangle)
                self.a = (f_image + f_object) / 2
                self.b = np.sqrt(self.a ** 2 - foci_sq / 4)
        elif self.a is not None and self.b is not None:
            foci = 2 * np.sqrt(self.a ** 2 - self.b ** 2)
            if have_f:
                self._offaxisangle = np.arccos((f_image ** 2 + f_object ** 2 - foci ** 2) / (2 * f_image * f_object))
            else:
                self._offaxisangle = np.arccos(1 - foci ** 2 / (2 * self.a ** 2))
        if self.a is None or self.b is None or self._offaxisangle is None:
            raise ValueError("Invalid mirror parameters")

    def _abi_params(self):
        return [self.a, self.b]

    def get_normal(self, Point):
        return mgeo.Normalize(np.array([-Point[0] / self.a ** 2, -Point[1] / self.b ** 2, -Point[2] / self.b ** 2]))

    def get_centre(self):
        """Point of the surface at the centre of the support (ART/ModuleMirror.py:695-714)."""
        foci = 2 * np.sqrt(self.a ** 2 - self.b ** 2)
        h = -foci / 2 / np.tan(self._offaxisangle)
        R = np.sqrt(foci ** 2 / 4 + h ** 2)
        sign = 1
        if math.isclose(self._offaxisangle, np.pi / 2):
            h = 0
        elif self._offaxisangle > np.pi / 2:
            h = -h
            sign = -1
        qa = 1 - self.a ** 2 / self.b ** 2
        qb = -2 * h
        qc = self.a ** 2 + h ** 2 - R ** 2
        z = (-qb + sign * np.sqrt(qb ** 2 - 4 * qa * qc)) / (2 * qa)
        if math.isclose(z ** 2, self.b ** 2):
            return np.array([0, 0, -self.b])
        return np.array([self.a * np.sqrt(1 - z ** 2 / self.b ** 2), 0, sign * z])

    _cloud_has_outline = False

    def _sag(self, x, y):
        return -self.b * np.sqrt(1 - (x / self.a) ** 2 - (y / self.b) ** 2)


def ReturnOptimalEllipsoidalAxes(Focal: float, AngleIncidence: float):
    """ART/ModuleMirror.py:755-777."""
    return Focal, Focal * np.cos(np.deg2rad(AngleIncidence))


class MirrorCylindrical(_Mirror):
    """Cylinder y^2 + z^2 = R^2; negative Radius = convex (ART/ModuleMirror.py:781-874)."""
    _abi_kind = _abi.ART_CYLINDER

    def __init__(self, Radius, Support):
        if Radius < 0:
            self.type = "CylindricalCX Mirror"
            self.radius = -Radius
        else:
            self.type = "CylindricalCC Mirror"
            self.radius = Radius
        self.support = Support

    def _abi_params(self):
        return [self.radius]

    def get_normal(self, Point):
        return mgeo.Normalize(np.array([0, -Point[1], -Point[2]]))

    def get_centre(self):
        return np.array([0, 0, -self.radius])

    def _sag(self, x, y):
        return -np.sqrt(self.radius ** 2 - y * y) + 0.0 * x


class DeformedMirror(_Mirror):
    """A mirror with surface defects (ART/ModuleMirror.py:945-980).  The defect offset always shifts the hit
    point; the perturbed normal is used only when tracing with IgnoreDefects=False (reference default: True)."""

    def __init__(self, Mirror, DeformationList):
        self.Mirror = Mirror
        self.DeformationList = list(DeformationList)
        self.type = Mirror.type
        self.support = self.Mirror.support
        for d in self.DeformationList:
            if not (hasattr(d, "_abi_table") or hasattr(d, "_abi_grid")):
                raise NotImplementedError(f"defect type {type(d).__name__} has no device implementation yet")
        # Any number of Zernike defects: offsets and slopes of defects with the same normalisation radius add up
        # (ART/ModuleMirror.py:952-980), so they are merged into ONE table per radius (_zernike_groups).  What stays
        # limited is the number of tables per mirror: distinct radii, and gridded height maps.
        if len(self._zernike_groups()) > _abi.ART_MAX_DEFECTS:
            raise NotImplementedError(f"Zernike defects with more than {_abi.ART_MAX_DEFECTS} different normalisation radii on one mirror")
        if len(self._grid_defects()) > _abi.ART_MAX_DEFECTS:
            raise NotImplementedError(f"at most {_abi.ART_MAX_DEFECTS} gridded defects per mirror are supported")

    @property
    def _abi_kind(self):
        return self.Mirror._abi_kind

    def _abi_params(self):
        return self.Mirror._abi_params()

    def _zernike_defects(self):
        return [d for d in self.DeformationList if hasattr(d, "_abi_table")]

    def _grid_defects(self):
        return [d for d in self.DeformationList if hasattr(d, "_abi_grid")]

    def _zernike_groups(self):
        """{R: summed coefficient dict} of the Zernike defects, by normalisation radius, in order of first appearance."""
        groups = {}
        for d in self._zernike_defects():
            g = groups.setdefault(float(d.R), {})
            for k, c in d.coefficients.items():
                g[k] = g.get(k, 0.0) + c
        return groups

    def _abi_defect_table(self):
        """(device table of all Zernike groups, number of tables, recurrence layout?) -- see ModuleDefects.zernike_table.
        One group above order 16 puts all of the mirror's tables into the recurrence layout, padded to one order."""
        from .ModuleDefects import zernike_table
        groups = self._zernike_groups()
        top = max(max(k[0] for k in g) for g in groups.values())
        rec = top > _abi.ART_ZERN_MAX_ORDER
        return (np.concatenate([zernike_table(R, g, recurrence=rec, order=max(2, top)) for R, g in groups.items()]),
                len(groups), rec)

    def get_normal(self, PointMirror):
        """Normal of the deformed surface at ONE point, host side (ART/ModuleMirror.py:952-961): the base normal
        folded with each defect's normal by slope addition."""
        normal = np.asarray(self.Mirror.get_normal(PointMirror), dtype=float)
        C = self.get_centre()
        for d in self.DeformationList:
            normal = mgeo.normal_add(normal, d.get_normal(PointMirror - C))
            normal = normal / np.linalg.norm(normal)
        return normal

    def get_centre(self):
        return self.Mirror.get_centre()

    def get_grid3D(self, NbPoint, **kwargs):
        """The undeformed surface: the render does not show the defects (ART/ModuleMirror.py:966-967)."""
        return self.Mirror.get_grid3D(NbPoint, **kwargs)

    def __hash__(self):
        return hash((hash(self.Mirror),) + tuple(hash(d) for d in self.DeformationList))


def ReflectionMirrorRayList(Mirror, ListRay, IgnoreDefects=False):
    """Reflect a bundle given in the mirror's own frame (ART/ModuleMirror.py:912-939): one identity-pose
    element on the device."""
    from . import ModuleProcessing as mp
    from .ModuleOpticalElement import OpticalElement
    # position = get_centre() with identity axes makes lab frame == optic frame (P_opt = P - pos + centre)
    oe = OpticalElement(Mirror, np.asarray(Mirror.get_centre(), dtype=float), np.array([0.0, 0.0, 1.0]),
                        np.array([1.0, 0.0, 0.0]))
    return mp.RayTracingCalculation(ListRay, [oe], IgnoreDefects=IgnoreDefects)[0]
