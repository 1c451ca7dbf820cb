"""Host-side `Ray` record with the reference's constructor and attributes (ART/ModuleOpticalRay.py:11-156).

A Ray is only a *view* of one slot of a device-resident RayBundle (or a single hand-made ray, e.g. the
alignment ray of OEPlacement); bundles are never stored as lists of these objects."""
import numpy as np


class Ray:
    __slots__ = ("_point", "_vector", "_path", "_number", "_wavelength", "_incidence", "_intensity")

    def __init__(self, Point, Vector, Path=(0.0,), Number=None, Wavelength=None, Incidence=None, Intensity=None):
        self.point = Point
        self.vector = Vector  # normalised by the setter, as in the reference (:85-90)
        self._path = Path
        self._wavelength = Wavelength
        self._incidence = Incidence
        self._intensity = Intensity
        if Number is not None and not isinstance(Number, (int, np.integer)):
            raise TypeError("Ray Number must be an integer.")
        self._number = None if Number is None else int(Number)

    @property
    def point(self):
        return self._point

    @point.setter
    def point(self, Point):
        if not (isinstance(Point, np.ndarray) and len(Point) == 3):
            raise TypeError("Ray Point must be a 3D numpy.ndarray, but it is  %s." % type(Point))
        self._point = Point

    @property
    def vector(self):
        return self._vector

    @vector.setter
    def vector(self, Vector):
        if not (isinstance(Vector, np.ndarray) and len(Vector) == 3 and np.linalg.norm(Vector) > 1e-9):
            raise TypeError("Ray Vector must be a 3D numpy.ndarray with finite length.")
        self._vector = Vector / np.linalg.norm(Vector)

    @property
    def path(self):
        return self._path

    @path.setter
    def path(self, Path):
        self._path = Path

    @property
    def number(self):
        return self._number

    @property
    def wavelength(self):
        return self._wavelength

    @wavelength.setter
    def wavelength(self, Wavelength):
        if type(Wavelength) not in (int, float, np.float64):
            raise TypeError("Ray Wavelength must be int or float or None.")
        self._wavelength = Wavelength

    @property
    def incidence(self):
        return self._incidence

    @incidence.setter
    def incidence(self, Incidence):
        if type(Incidence) not in (float, np.float64):
            raise TypeError("Ray Incidence must be a float or None.")
        self._incidence = Incidence

    @property
    def intensity(self):
        return self._intensity

    @intensity.setter
    def intensity(self, Intensity):
        if type(Intensity) not in (int, float, np.float64):
            raise TypeError("Ray Intensity must be int or float or None.")
        self._intensity = Intensity

    def copy_ray(self):
        """New Ray with the same properties (:145-149)."""
        return Ray(self.point, self.vector, self.path, self.number, self.wavelength, self.incidence, self.intensity)

    def __hash__(self):
        return hash(tuple(self.point) + tuple(self.vector)
                    + (self.path, self.number, self.wavelength, self.incidence, self.intensity))
