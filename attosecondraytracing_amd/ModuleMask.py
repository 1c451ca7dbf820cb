"""Mask: a plane that blocks the rays hitting its support (ART/ModuleMask.py:24-70).
The transmission test and path/incidence update run in the HIP kernel (kind ART_MASK)."""
import numpy as np

from . import _abi
from . import ModuleGeometry as mgeo


class Mask:
    __deepcopy__ = mgeo.flat_deepcopy

    _abi_kind = _abi.ART_MASK

    def __init__(self, Support):
        self.type = "Mask"
        self.support = Support

    def _abi_params(self):
        return []

    def get_normal(self, Point):
        return np.array([0, 0, 1])

    def get_centre(self):
        return np.array([0, 0, 0])

    def get_grid3D(self, NbPoint, **kwargs):
        """Sample points of the mask's plane inside its support, for the 3-D render (ART/ModuleMask.py:72-91)."""
        from .ModuleMirror import _surface_cloud
        return _surface_cloud(self, lambda x, y: np.zeros_like(x), True, NbPoint, bool(kwargs.get("edges")))

    def __hash__(self):
        return hash(("Mask", hash(self.support)))

    def _get_intersection(self, Ray):
        """Point where ONE ray (given in the mask's frame) crosses the mask plane if it passes, else None
        (ART/ModuleMask.py:51-61): a one-ray trace on the device."""
        hit = TransmitMaskRayList(self, [Ray])
        return hit[0].point if len(hit) == 1 else None


def _TransmitMaskRay(Mask, PointMask, Ray):
    """ONE ray continued from PointMask with its direction unchanged (ART/ModuleMask.py:93-108), on the host."""
    out = Ray.copy_ray()
    out.point = PointMask
    out.vector = Ray.vector
    out.incidence = mgeo.AngleBetweenTwoVectors(Ray.vector, Mask.get_normal(PointMask))
    out.path = Ray.path + (np.linalg.norm(PointMask - Ray.point),)
    return out


def TransmitMaskRayList(Mask, RayList):
    """The rays that pass the mask, given in the mask's own frame (ART/ModuleMask.py:112-136): one identity-pose
    element on the device."""
    from . import ModuleProcessing as mp
    from .ModuleOpticalElement import OpticalElement
    oe = OpticalElement(Mask, np.zeros(3), np.array([0.0, 0.0, 1.0]), np.array([1.0, 0.0, 0.0]))
    return mp.RayTracingCalculation(RayList, [oe])[0]
