"""Mask: a plane that blocks the rays hitting its support (ART/ModuleMask.py:24-70).
The transmission test and path/incidence update run in the HIP kernel (kind ART_MASK)."""
import numpy as np

from . import _abi


class Mask:
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

    def __hash__(self):
        return hash(("Mask", hash(self.support)))
