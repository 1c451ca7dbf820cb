"""Aperture shapes of optics ("supports"), API of ART/ModuleSupport.py.

Only what the tracing path needs is built: the five `_IncludeSupport` predicates (which the HIP kernels
evaluate per ray from `_abi_kind` / `_abi_params`), the circumscribed rectangle/circle used by defects and
source sizing.  Plot meshes and contours of the reference (`_get_grid`, `_ContourSupport`, ...) are out of
scope (rendering)."""
from abc import ABC, abstractmethod

import numpy as np

from . import _abi
from . import ModuleGeometry as mgeo


class Support(ABC):
    """Abstract base class for optics supports."""

    _abi_kind = None

    @abstractmethod
    def _IncludeSupport(self, Point):
        pass

    @abstractmethod
    def _abi_params(self):
        """Up to six doubles, layout documented in include/art_hip.h (enum ArtSupportKind)."""

    def __hash__(self):
        return hash((type(self).__name__,) + tuple(float(v) for v in self._abi_params()))

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
