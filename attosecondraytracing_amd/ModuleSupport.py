"""Aperture shapes of optics ("supports"), API of ART/ModuleSupport.py.

The five `_IncludeSupport` predicates (which the HIP kernels evaluate per ray from `_abi_kind` / `_abi_params`), the
circumscribed rectangle/circle used by defects and source sizing, and the sample points the plots draw an aperture
with: `_ContourSupport` (MirrorProjection), `_get_grid` / `_Contour_points` (RayRenderGraph)."""
import math
from abc import ABC, abstractmethod

import numpy as np

from . import _abi
from . import ModuleGeometry as mgeo


class Support(ABC):
    """Abstract base class for optics supports."""
    __deepcopy__ = mgeo.flat_deepcopy


    _abi_kind = None

    @abstractmethod
    def _IncludeSupport(self, Point):
        pass

    @abstractmethod
    def _abi_params(self):
        """Up to six doubles, layout documented in include/art_hip.h (enum ArtSupportKind)."""

    def _content_hash(self):
        return hash((type(self).__name__,) + tuple(float(v) for v in self._abi_params()))

    def __hash__(self):
        return mgeo.memo_hash(self, self._content_hash)      # (recomputed from the contents, once per hash epoch)

    def _ContourSupport(self, Figure):
        """Outline of the support (and of its hole) on a new equal-aspect axes of `Figure`; used by MirrorProjection
        (the `_ContourSupport` of every support class of ART/ModuleSupport.py).  Returns the axes."""
        from matplotlib import patches
        axe = Figure.add_subplot(111, aspect="equal")
        if hasattr(self, "radius"):
            axe.add_patch(patches.Circle((0, 0), self.radius, alpha=0.08))
        else:
            axe.add_patch(patches.Rectangle((-self.dimX * 0.5, -self.dimY * 0.5), self.dimX, self.dimY, alpha=0.08))
        if hasattr(self, "radiushole"):
            axe.add_patch(patches.Circle((self.centerholeX, self.centerholeY), self.radiushole, color="white", alpha=1))
        elif hasattr(self, "holeX"):
            axe.add_patch(patches.Rectangle((-self.holeX * 0.5 + self.centerholeX, -self.holeY * 0.5 + self.centerholeY),
                                            self.holeX, self.holeY, color="white", alpha=1))
        return axe

    # ---- sample points for the 3-D render (the `_get_grid` / `_Contour_points` of every support class of
    # ART/ModuleSupport.py; point counts and order as there, tests/golden/render_grids.npz) ----
    def _lattice_counts(self, NbPoint):
        """Columns and rows of the lattice on a rectangular support (ART/ModuleSupport.py:232-249)."""
        ratio, skew = self.dimX / self.dimY, (self.dimX - self.dimY) / self.dimY
        nbx = int(math.sqrt(ratio * NbPoint + 0.25 * skew ** 2) - 0.5 * skew)
        return nbx, int(NbPoint / nbx)

    def _grid_xy(self, NbPoint):
        """(m, 2) array: Vogel spiral on a round support, lattice (x-major) on a rectangular one, without the points
        that fall into the hole."""
        if hasattr(self, "radius"):
            xy = mgeo.SpiralVogel(NbPoint, self.radius)
        else:
            nbx, nby = self._lattice_counts(NbPoint)
            gx, gy = np.meshgrid(np.linspace(-self.dimX / 2, self.dimX / 2, nbx),
                                 np.linspace(-self.dimY / 2, self.dimY / 2, nby), indexing="ij")
            xy = np.column_stack((gx.ravel(), gy.ravel()))
        if hasattr(self, "radiushole"):
            keep = (xy[:, 0] - self.centerholeX) ** 2 + (xy[:, 1] - self.centerholeY) ** 2 > self.radiushole ** 2
        elif hasattr(self, "holeX"):
            keep = (np.abs(xy[:, 0] - self.centerholeX) > self.holeX / 2) | (np.abs(xy[:, 1] - self.centerholeY) > self.holeY / 2)
        else:
            return xy
        return xy[keep]

    def _get_grid(self, NbPoint: int, **kwargs):
        return list(self._grid_xy(NbPoint))

    def _contour_xy(self, NbPoint):
        """Outline points: (outer, hole or None), the hole in the order its closed edge loop runs."""
        round_outer = hasattr(self, "radius")
        outer_len = 2 * math.pi * self.radius if round_outer else 2 * (self.dimX + self.dimY)
        if hasattr(self, "radiushole"):
            if round_outer:     # shares by radius (ART/ModuleSupport.py:184-197)
                n_hole = NbPoint - int(round(NbPoint - NbPoint * self.radiushole / self.radius))
            else:               # shares by length (:355-369)
                hole_len = 2 * math.pi * self.radiushole
                n_hole = int(round(hole_len / (outer_len + hole_len) * NbPoint))
            hole = gen_circle_contour(self.radiushole, n_hole).reshape(-1, 2)
        elif hasattr(self, "holeX"):
            hole_len = 2 * (self.holeX + self.holeY)
            n_hole = int(round(hole_len / (outer_len + hole_len) * NbPoint))
            hole = gen_rectangle_contour(self.holeX, self.holeY, n_hole)[::-1]
        else:
            n_hole, hole = 0, None
        outer = (gen_circle_contour(self.radius, NbPoint - n_hole) if round_outer
                 else gen_rectangle_contour(self.dimX, self.dimY, NbPoint - n_hole))
        if hole is not None:
            hole = hole + np.array([self.centerholeX, self.centerholeY])
        return outer.reshape(-1, 2), hole

    def _Contour_points(self, NbPoint=100, edges=False):
        outer, hole = self._contour_xy(NbPoint)
        return flatten_point_arrays(outer, [] if hole is None else [hole], edges=edges)


def gen_circle_contour(radius, NbPoints):
    """NbPoints points on a circle, counter-clockwise from angle 0 (ART/ModuleSupport.py:566-574)."""
    if NbPoints == 0:
        return np.array([])
    phi = 2 * math.pi / NbPoints * np.arange(NbPoints)
    return np.column_stack((radius * np.cos(phi), radius * np.sin(phi)))


def gen_rectangle_contour(dimX, dimY, NbPoints):
    """Outline of a rectangle, clockwise from its top-left corner; NbPoints is shared between ONE horizontal and ONE
    vertical side, so about 2 NbPoints - 4 points come back (ART/ModuleSupport.py:546-563)."""
    along = max(math.ceil(dimX / (dimX + dimY) * NbPoints), 2)      # points on a horizontal side, corners included
    up = max(NbPoints - along, 2)          # ... on a vertical side (the reference divides by zero when a side gets fewer than two)
    xs = np.linspace(-dimX / 2, dimX / 2, along)
    ys = np.linspace(dimY / 2, -dimY / 2, up)
    sides = (np.column_stack((xs, np.full(along, dimY / 2))),                           # top, left to right
             np.column_stack((np.full(up - 1, dimX / 2), ys[1:])),                      # right, downwards
             np.column_stack((xs[::-1][1:], np.full(along - 1, -dimY / 2))),            # bottom, right to left
             np.column_stack((np.full(max(up - 2, 0), -dimX / 2), ys[::-1][1:up - 1])))  # left, upwards, both corners taken
    return np.concatenate(sides)


def flatten_point_arrays(outer, holes=(), edges=False):
    """Outer outline followed by the hole outlines as ONE list of points; with `edges`, also the closed index loop of
    every hole (ART/ModuleSupport.py:577-596)."""
    coords = list(outer)
    loops = []
    for arr in holes:
        first = len(coords)
        coords.extend(arr)
        loops.append(list(range(first, first + len(arr))) + [first])
    return (coords, loops) if edges else coords


class SupportRound(Support):
    """Disk of radius `radius` (ART/ModuleSupport.py:46-105)."""
    _abi_kind = _abi.ART_SUP_ROUND

    def __init__(self, Radius: float):
        self.radius = Radius

    def _IncludeSupport(self, Point):
        return mgeo.IncludeDisk(self.radius, Point)

    def _abi_params(self):
        return [self.radius]

    def _CircumRect(self):
        return np.array([self.radius * 2, self.radius * 2])

    def _CircumCirc(self):
        return self.radius


class SupportRoundHole(Support):
    """Disk with a round hole (ART/ModuleSupport.py:109-197)."""
    _abi_kind = _abi.ART_SUP_ROUNDHOLE

    def __init__(self, Radius: float, RadiusHole: float, CenterHoleX: float, CenterHoleY: float):
        self.radius = Radius
        self.radiushole = RadiusHole
        self.centerholeX = CenterHoleX
        self.centerholeY = CenterHoleY

    def _IncludeSupport(self, Point):
        hole = (Point[0] - self.centerholeX, Point[1] - self.centerholeY)
        return mgeo.IncludeDisk(self.radius, Point) and not mgeo.IncludeDisk(self.radiushole, hole)

    def _abi_params(self):
        return [self.radius, self.radiushole, self.centerholeX, self.centerholeY]

    def _CircumRect(self):
        return np.array([self.radius * 2, self.radius * 2])

    def _CircumCirc(self):
        return self.radius


class SupportRectangle(Support):
    """Rectangle dimX x dimY (ART/ModuleSupport.py:200-270)."""
    _abi_kind = _abi.ART_SUP_RECT

    def __init__(self, DimensionX: float, DimensionY: float):
        self.dimX = DimensionX
        self.dimY = DimensionY

    def _IncludeSupport(self, Point) -> bool:
        return mgeo.IncludeRectangle(self.dimX, self.dimY, Point)

    def _abi_params(self):
        return [self.dimX, self.dimY]

    def _CircumRect(self):
        return np.array([self.dimX, self.dimY])

    def _CircumCirc(self):
        return np.sqrt(self.dimX ** 2 + self.dimY ** 2) / 2


class SupportRectangleHole(Support):
    """Rectangle with a round hole (ART/ModuleSupport.py:273-370)."""
    _abi_kind = _abi.ART_SUP_RECTHOLE

    def __init__(self, DimensionX: float, DimensionY: float, RadiusHole: float, CenterHoleX: float,
                 CenterHoleY: float):
        self.dimX = DimensionX
        self.dimY = DimensionY
        self.radiushole = RadiusHole
        self.centerholeX = CenterHoleX
        self.centerholeY = CenterHoleY

    def _IncludeSupport(self, Point):
        hole = (Point[0] - self.centerholeX, Point[1] - self.centerholeY)
        return mgeo.IncludeRectangle(self.dimX, self.dimY, Point) and not mgeo.IncludeDisk(self.radiushole, hole)

    def _abi_params(self):
        return [self.dimX, self.dimY, self.radiushole, self.centerholeX, self.centerholeY]

    def _lattice_counts(self, NbPoint):
        """This class counts its lattice differently from the other rectangles (ART/ModuleSupport.py:328-335)."""
        return int(self.dimX / self.dimY * math.sqrt(NbPoint)), int(self.dimY / self.dimX * math.sqrt(NbPoint))

    def _CircumRect(self):
        return np.array([self.dimX, self.dimY])

    def _CircumCirc(self):
        return np.sqrt(self.dimX ** 2 + self.dimY ** 2) / 2


class SupportRectangleRectHole(Support):
    """Rectangle with a rectangular hole (ART/ModuleSupport.py:373-491)."""
    _abi_kind = _abi.ART_SUP_RECTRECTHOLE

    def __init__(self, DimensionX: float, DimensionY: float, HoleX: float, HoleY: float, CenterHoleX: float,
                 CenterHoleY: float):
        self.dimX = DimensionX
        self.dimY = DimensionY
        self.holeX = HoleX
        self.holeY = HoleY
        self.centerholeX = CenterHoleX
        self.centerholeY = CenterHoleY

    def _IncludeSupport(self, Point):
        hole = (Point[0] - self.centerholeX, Point[1] - self.centerholeY)
        return mgeo.IncludeRectangle(self.dimX, self.dimY, Point) and not mgeo.IncludeRectangle(
            self.holeX, self.holeY, hole)

    def _abi_params(self):
        return [self.dimX, self.dimY, self.holeX, self.holeY, self.centerholeX, self.centerholeY]

    def _CircumRect(self):
        return np.array([self.dimX, self.dimY])

    def _CircumCirc(self):
        return np.sqrt(self.dimX ** 2 + self.dimY ** 2) / 2
