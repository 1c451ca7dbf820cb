"""Ray sources, API of ART/ModuleSource.py.  Every function returns a device-resident RayBundle.

Point sources and plane-wave disks are generated on the GPU from the ray index (Vogel spiral,
`art_make_source`) and so are the Gaussian intensity weights (`art_gaussian_intensity`: max-angle reduction +
weights, nothing returns to the host).  ExtendedSource: `art_make_extended_source`."""
import numpy as np


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
    # the bundle is a pure function of these arguments (the generator is deterministic)
    b.tag_content(("source", int(kind), float(size), np.asarray(S, dtype=float).tobytes(),
                   np.asarray(Axis, dtype=float).tobytes(), int(n), int(n_total), Wavelength))
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
    """Disk of point sources (ART/ModuleSource.py:85-131), numbering included: ray number = point source index *
    rays per point source + index within the cone, generated on the device from that number."""
    n_src = min(max(30, int(250 * Diameter)), int(NbRays / 300))
    per = max(300, int(NbRays / n_src))
    be = _lib.get_backend()
    b = RayBundle.allocate(n_src * per, backend=be)
    b.wavelength = Wavelength
    b.number = None
    rot = mgeo.rotation_matrix(_EZ, np.asarray(Axis, dtype=float))
    be.make_extended_source(Diameter / 2, Divergence, n_src, per, rot, np.asarray(S, dtype=float), 0, n_src * per,
                            b.view())
    b.tag_content(("extended source", float(Diameter), float(Divergence), int(n_src), int(per),
                   np.asarray(S, dtype=float).tobytes(), np.asarray(Axis, dtype=float).tobytes(), Wavelength))
    return b


def PlaneWaveSquare(Centre, Axis, SideLength: float, NbRays: int, Wavelength=None):
    """Collimated square beam (ART/ModuleSource.py:171-204).  As shipped, the reference only runs for NbRays < 4
    (one grid point per side): its test `abs(x) > 1e-4` is applied to the whole coordinate ARRAY, which NumPy refuses
    for more than one element.  Same here: the same ValueError for larger NbRays, the same rays below."""
    x = np.linspace(-SideLength / 2, SideLength / 2, int(np.sqrt(NbRays)))
    y = np.linspace(-SideLength / 2, SideLength / 2, int(np.sqrt(NbRays)))
    pts = [[0.0, 0.0, 0.0]]
    for i in x:
        for j in y:
            if abs(x) > 1e-4 and abs(y) > 1e-4:   # (sic) raises ValueError for arrays longer than 1
                pts.append([i, j, 0.0])
    pts = np.asarray(pts, dtype=float)
    M = mgeo.rotation_matrix(_EZ, np.asarray(Axis, dtype=float))
    vec = np.tile(M @ _EZ, (len(pts), 1))
    b = RayBundle.from_arrays(pts @ M.T + np.asarray(Centre, dtype=float), vec, np.arange(len(pts)), None, Wavelength)
    return b


def ApplyGaussianIntensityToRayList(RayList, IntensityFraction=1 / np.e ** 2):
    """Gaussian intensity profile, 1 on axis and IntensityFraction at the edge (ART/ModuleSource.py:219-261):
    in angle for diverging bundles, in distance from the origin for plane waves."""
    if IntensityFraction >= 1 or IntensityFraction <= 0:
        print("When applying a Gaussian intensity profile to a ray list, the IntensityFraction should be between "
              "0 and 1! I'm setting it to 1/e^2.")
        IntensityFraction = 1 / np.e ** 2
    from . import ModuleProcessing as mp
    B = RayList if isinstance(RayList, RayBundle) else RayBundle.from_ray_list(RayList)
    before = B.content_key()
    if hasattr(B.backend, "gaussian_intensity_central") and B.n_slots > 0:
        # the axis (FindCentralRay's mean vector) is formed on the device from the bundle's sums: no host round trip
        B.intensity = B.backend.gaussian_intensity_central(B.view(), IntensityFraction, B.n_slots)
    else:
        axis = mp.FindCentralRay(B).vector
        B.intensity = B.backend.gaussian_intensity(B.view(), axis, IntensityFraction, B.n_slots)
    B.touch()
    if before[0] != "bundle":       # weights of a tagged bundle are a pure function of its tag and the fraction
        B.tag_content((before, "gaussian intensity", float(IntensityFraction)))
    return B
