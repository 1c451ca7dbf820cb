"""Host-side `Ray` record with the constructor signature and attributes of the reference's Ray
(ART/ModuleOpticalRay.py:11-156).

Here a Ray is only a *view* of one slot of a device-resident RayBundle (bundle.py), or a single hand-made ray such as
the alignment ray of OEPlacement; bundles are never stored as lists of these objects.  The validation rules are the
reference's: 3-vectors must be numpy arrays, the direction is normalised on assignment and must be longer than 1e-9,
`number` is fixed at construction."""
import numpy as np

_NUMERIC = (int, float, np.float64)


def _is_vec3(v):
    return isinstance(v, np.ndarray) and len(v) == 3


def _checked(name, allowed, message):
    """Property whose setter accepts only the listed scalar types (TypeError otherwise)."""
    slot = "_" + name

    def fget(self):
        return getattr(self, slot)

    def fset(self, value):
        if type(value) not in allowed:
            raise TypeError(message)
        setattr(self, slot, value)

    return property(fget, fset)


class Ray:
    __slots__ = ("_point", "_vector", "_path", "_number", "_wavelength", "_incidence", "_intensity")

    def __init__(self, Point, Vector, Path=(0.0,), Number=None, Wavelength=None, Incidence=None, Intensity=None):
        if Number is not None and type(Number) is not int:      # a NumPy integer is refused too (:60-63)
            raise TypeError("Ray Number must be an integer.")
        self.point, self.vector = Point, Vector
        # the optional attributes bypass their setters at construction, so None is allowed (:57-60)
        self._path, self._wavelength, self._incidence, self._intensity = Path, Wavelength, Incidence, Intensity
        self._number = Number

    # origin of the ray
    def _get_point(self):
        return self._point

    def _set_point(self, Point):
        if not _is_vec3(Point):
            raise TypeError("Ray Point must be a 3D numpy.ndarray, but it is  %s." % type(Point))
        self._point = Point

    point = property(_get_point, _set_point)

    # unit direction; re-normalised whenever it is assigned
    def _get_vector(self):
        return self._vector

    def _set_vector(self, Vector):
        length = np.linalg.norm(Vector) if _is_vec3(Vector) else 0.0
        if not length > 1e-9:
            raise TypeError("Ray Vector must be a 3D numpy.ndarray with finite length.")
        self._vector = Vector / length

    vector = property(_get_vector, _set_vector)

    # tuple of the segment lengths travelled so far
    def _get_path(self):
        return self._path

    def _set_path(self, Path):
        self._path = Path

    path = property(_get_path, _set_path)

    number = property(lambda self: self._number)   # read-only: links a ray to its source ray
    wavelength = _checked("wavelength", _NUMERIC, "Ray Wavelength must be int or float or None.")
    incidence = _checked("incidence", (float, np.float64), "Ray Incidence must be a float or None.")
    intensity = _checked("intensity", _NUMERIC, "Ray Intensity must be int or float or None.")

    def copy_ray(self):
        """A new Ray carrying the same seven attributes (:145-149)."""
        return Ray(self._point, self._vector, self._path, self._number, self._wavelength, self._incidence,
                   self._intensity)

    def __hash__(self):
        return hash((*self._point, *self._vector, self._path, self._number, self._wavelength, self._incidence,
                     self._intensity))
