"""Surface defects of mirrors, API of ART/ModuleDefects.py.

`Zernike` is the defect the HIP kernels evaluate per ray (recurrences in csrc/art_device.h, coefficient
table staged in LDS): this class only packs the dense coefficient table.  `Fourrier` / `MeasuredMap`
(gridded maps, bilinear lookup) are not built yet -- constructing one raises NotImplementedError rather
than silently tracing without the defect."""
from abc import ABC, abstractmethod

import numpy as np

from . import _abi


class Defect(ABC):
    @abstractmethod
    def RMS(self):
        pass

    @abstractmethod
    def PV(self):
        pass


class Zernike(Defect):
    """Zernike-polynomial surface error (ART/ModuleDefects.py:149-180).

    coefficients : dict {(n, m): c}, m = 0..n, polynomials and normalisation as produced by
    ART/recursive_zernike_generator.py (Andersen 2018 Cartesian recurrences)."""

    def __init__(self, Support, coefficients):
        self.coefficients = dict(coefficients)
        self.max_order = int(np.max([k[0] for k in coefficients]))
        self.support = Support
        self.R = Support._CircumCirc()
        for (n, m) in self.coefficients:
            if not (0 <= m <= n):
                raise ValueError(f"Zernike index (n={n}, m={m}) must satisfy 0 <= m <= n")
        if self.max_order > _abi.ART_ZERN_MAX_ORDER:
            raise NotImplementedError(
                f"Zernike radial order {self.max_order} exceeds the kernels' maximum {_abi.ART_ZERN_MAX_ORDER}")

    def _abi_table(self):
        """ART_ZERN_STRIDE doubles: [R, max_order, dense coefficients] (include/art_hip.h)."""
        t = np.zeros(_abi.ART_ZERN_STRIDE)
        t[0] = self.R
        t[1] = max(2, self.max_order)  # recursive_zernike_generator.py:35-37
        for (n, m), c in self.coefficients.items():
            t[2 + n * (n + 1) // 2 + m] += c
        return t

    def RMS(self):
        return np.sqrt(np.sum([i ** 2 for i in self.coefficients.values()]))

    def PV(self):
        pass

    def __hash__(self):
        return hash((self.R,) + tuple(sorted(self.coefficients.items())))


class Fourrier(Defect):
    """Random FFT-generated surface map (ART/ModuleDefects.py:69-146): not built yet."""

    def __init__(self, *a, **k):
        raise NotImplementedError("Fourrier defect maps (gridded bilinear lookup on device) are not built yet")

    def RMS(self):
        pass

    def PV(self):
        pass


class MeasuredMap(Defect):
    """Measured surface map (ART/ModuleDefects.py:34-67): not built yet."""

    def __init__(self, *a, **k):
        raise NotImplementedError("MeasuredMap defects (gridded bilinear lookup on device) are not built yet")

    def RMS(self):
        pass

    def PV(self):
        pass
