"""Ray sources, API of ART/ModuleSource.py.  Every function returns a device-resident RayBundle.

Point sources and plane-wave disks are generated on the GPU from the ray index (Vogel spiral,
`art_make_source`) and so are the Gaussian intensity weights (`art_gaussian_intensity`: max-angle reduction +
weights, nothing returns to the host).  ExtendedSource is (vectorised) host NumPy, then uploaded."""
import numpy as np
import torch

from . import _lib
from . import ModuleGeometry as mgeo
from .bundle import RayBundle

_EZ = np.array([0.0, 0.0, 1.0])


def _device_source(kind, size, S, Axis, n, n_total, Wavelength):
    be = _lib.get_backend()
    b = RayBundle.allocate(n, backend=be)
    b.wavelength = Wavelength
    b.number = None  # slot index == Ray.number for these sources
    rot = mgeo.rotation_matrix(_EZ, np.asarray(Axis, dtype=float))
    be.make_source(kind, size, rot, np.asarray(S, dtype=float), 0, n, n_total, b.view())
    return b


def _Cone(Angle: float, NbRays: int, Wavelength=None):
    """Rays from the origin filling a cone about +z (ART/ModuleSource.py:23-50)."""
    return _device_source(0, Angle, np.zeros(3), _EZ, NbRays, NbRays, Wavelength)


def PointSource(S, Axis, Divergence: float, NbRays: int, Wavelength=None):
    """Point source at S, cone half-angle Divergence (rad) about Axis (ART/ModuleSource.py:54-81)."""
    return _device_source(0, Divergence, S, Axis, NbRays, NbRays, Wavelength)


def PlaneWaveDisk(Centre, Axis, Radius: float, NbRays: int, Wavelength=None):
    """Collimated round beam; like the reference it emits NbRays-1 rays (ART/ModuleSource.py:135-169, :162)."""
    return _device_source(1, Radius, Centre, Axis, NbRays - 1, NbRays, Wavelength)


def ExtendedSource(S, Axis, Diameter: float, Divergence: float, NbRays: int, Wavelength=None):
    """Disk of point sources (ART/ModuleSource.py:85-131), numbering included."""
    n_src = min(max(30, int(250 * Diameter)), int(NbRays / 300))
    XY = mgeo.SpiralVogel(n_src, Diameter / 2)
    per = max(300, int(NbRays / n_src))
    cone = mgeo.SpiralVogel(per, np.tan(Divergence))
    vec = np.concatenate([cone, np.ones((per, 1))], axis=1)
    vec /= np.linalg.norm(vec, axis=1)[:, None]
    pts = np.repeat(np.concatenate([XY, np.zeros((n_src, 1))], axis=1), per, axis=0)
    vecs = np.tile(vec, (n_src, 1))
    M = mgeo.rotation_matrix(_EZ, np.asarray(Axis, dtype=float))
    pts = pts @ M.T + np.asarray(S, dtype=float)
    vecs = vecs @ M.T
    return RayBundle.from_arrays(pts, vecs, np.arange(n_src * per), None, Wavelength)


def ApplyGaussianIntensityToRayList(RayList, IntensityFraction=1 / np.e ** 2):
    """Gaussian intensity profile, 1 on axis and IntensityFraction at the edge (ART/ModuleSource.py:219-261):
    in angle for diverging bundles, in distance from the origin for plane waves."""
    if IntensityFraction >= 1 or IntensityFraction <= 0:
        print("When applying a Gaussian intensity profile to a ray list, the IntensityFraction should be between "
              "0 and 1! I'm setting it to 1/e^2.")
        IntensityFraction = 1 / np.e ** 2
    from . import ModuleProcessing as mp
    B = RayList if isinstance(RayList, RayBundle) else RayBundle.from_ray_list(RayList)
    axis = mp.FindCentralRay(B).vector
    B.intensity = B.backend.gaussian_intensity(B.view(), axis, IntensityFraction, B.n_slots)
    B.touch()
    return B
