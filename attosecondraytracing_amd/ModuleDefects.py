"""Surface defects of mirrors, API of ART/ModuleDefects.py.

`Zernike`: evaluated per ray in the HIP kernels (recurrences in csrc/art_device.h, coefficient table staged in
LDS); this class only packs the dense coefficient table.  `Fourrier`: height map synthesised on the host like the
reference, bilinear look-up per ray on the device.  `MeasuredMap`: cannot be constructed in the reference either."""
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
    """Random rough surface with a power-law spectrum between the spatial periods `smallest` and `biggest` (mm),
    scaled to the requested RMS (ART/ModuleDefects.py:69-146).  The map is synthesised on the host exactly as the
    reference does (same NumPy calls, including the single `np.random.uniform` draw, so a seeded run reproduces the
    reference's map bit for bit) and uploaded once; the per-ray bilinear look-up of the height runs in the kernels.

    As in the reference under NumPy >= 1.24, only the height offset is usable: `get_normal` raises."""

    def __init__(self, Support, RMS, slope=-2, smallest=0.1, biggest=None):
        rect = Support._CircumRect()
        if biggest is None:
            biggest = np.max(rect)
        k_max, k_min = 2 / smallest, 2 / biggest
        nkx = int(round(k_max * rect[0] / 2)) + 1
        nky = int(round(k_max * rect[1]))
        kx = np.linspace(0, k_max, num=nkx, dtype="float32", endpoint=False)[None, :]
        ky = np.linspace(-k_max, k_max, num=nky, dtype="float32", endpoint=False)[:, None]
        kr = np.sqrt(kx ** 2 + ky ** 2)
        band = (kr >= k_min) & (kr <= k_max)
        # np.ma's power() in the reference wraps the exponent in a 0-d array, which promotes float32 ** exponent
        # to float64: keep that (the amplitudes are float64 powers of float32 wavenumbers)
        amplitude = np.power(np.where(band, kr, np.float32(1)), np.asarray(slope))
        phase = np.random.uniform(0, 2 * np.pi, size=kr.shape).astype("float32")
        spectrum = (amplitude * np.exp(1j * phase)) * band.astype(np.int64)      # complex128 holding complex64 values
        self._spectrum, self._kx, self._ky = spectrum, kx, ky
        deformation = np.fft.irfft2(np.fft.ifftshift(spectrum, axes=0))
        self._scale = RMS / np.std(deformation)
        deformation *= self._scale
        self.deformation = deformation
        self.rms = np.std(deformation)
        self.support = Support
        self._X = np.linspace(-rect[0] / 2, rect[0] / 2, num=(nkx - 1) * 2)
        self._Y = np.linspace(-rect[1] / 2, rect[1] / 2, num=nky)
        self._device = None

    @property
    def DerivX(self):
        """Slope map d/dx (ART/ModuleDefects.py:101); computed on demand, not needed for tracing."""
        s = self._spectrum
        return np.fft.irfft2(np.fft.ifftshift(s * 1j * self._kx * self._scale, axes=0)) * np.pi / 2

    @property
    def DerivY(self):
        """Slope map d/dy (ART/ModuleDefects.py:102-103)."""
        s, n = self._spectrum, self._ky.shape[0]
        kY = np.concatenate((self._ky[n // 2:], self._ky[:n // 2]))
        return np.fft.irfft2(np.fft.ifftshift(s * 1j * self._scale, axes=0) * kY) * np.pi / 2

    def _abi_grid(self, backend):
        """(ArtGridDefect fields, device tensor of the transposed map [nx, ny]); uploaded once per defect object."""
        if self._device is None:
            self._device = backend.from_numpy(np.ascontiguousarray(self.deformation.T, dtype=np.float64))
        X, Y = self._X, self._Y
        return dict(h=self._device.data_ptr(), nx=len(X), ny=len(Y), x0=float(X[0]), y0=float(Y[0]),
                    dx=float((X[-1] - X[0]) / (len(X) - 1)), dy=float((Y[-1] - Y[0]) / (len(Y) - 1))), self._device

    def get_offset(self, Point):
        """Height at Point (host helper for a single point; bundles use the device look-up)."""
        from scipy.interpolate import RegularGridInterpolator
        if not hasattr(self, "_interp"):
            self._interp = RegularGridInterpolator((self._X, self._Y), np.transpose(self.deformation), method="linear")
        return self._interp(np.asarray(Point, dtype=float)[:2])

    def get_normal(self, Point):
        raise ValueError("Fourrier.get_normal is unusable in the reference under NumPy >= 1.24 "
                         "(ART/ModuleDefects.py:125-126 builds a ragged array); only the height offset is defined")

    def RMS(self):
        return self.rms

    def PV(self):
        pass

    def __hash__(self):
        return hash((id(self), float(self.rms)))


class MeasuredMap(Defect):
    """Measured surface map (ART/ModuleDefects.py:34-67).  Its constructor cannot run in the reference
    (`np.gradient(map, rect / map.shape)` at :43 passes one length-2 array as spacing for a 2-D map: TypeError),
    so there is no behaviour to reproduce; constructing one here fails loudly as well."""

    def __init__(self, *a, **k):
        raise NotImplementedError("MeasuredMap: the reference's constructor raises TypeError (ModuleDefects.py:43); "
                                  "nothing to reproduce")

    def RMS(self):
        pass

    def PV(self):
        pass
