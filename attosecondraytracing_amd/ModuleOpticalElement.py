"""OpticalElement: an optic (mirror or mask) plus its pose in the lab frame, API of
ART/ModuleOpticalElement.py.  Its three vectors are what the kernels' constant frame maps are built from
(ModuleGeometry.frame_maps)."""
import numpy as np

from . import ModuleGeometry as mgeo


class OpticalElement:
    def __init__(self, Type, Position, Normal, MajorAxis):
        self._type = Type
        self.position = Position
        self.normal = mgeo.Normalize(Normal)
        self.majoraxis = mgeo.Normalize(MajorAxis)

    @property
    def position(self):
        return self._position

    @position.setter
    def position(self, NewPosition):
        if not (isinstance(NewPosition, np.ndarray) and len(NewPosition) == 3):
            raise TypeError("Position must be a 3D numpy.ndarray.")
        self._position = NewPosition

    @property
    def normal(self):
        return self._normal

    @normal.setter
    def normal(self, NewNormal):
        if not (isinstance(NewNormal, np.ndarray) and len(NewNormal) == 3 and np.linalg.norm(NewNormal) > 0):
            raise TypeError("Normal must be a 3D numpy.ndarray with finite length.")
        new = mgeo.Normalize(NewNormal)
        # keep the major axis perpendicular: carry it along with the rotation old normal -> new normal
        # (ART/ModuleOpticalElement.py:126-141; skipped during construction, when no major axis exists yet)
        if hasattr(self, "_majoraxis") and abs(np.dot(new, self._majoraxis)) > 1e-12:
            self._majoraxis = mgeo.RotationAroundAxis(np.cross(self._normal, NewNormal),
                                                      mgeo.AngleBetweenTwoVectors(self._normal, NewNormal),
                                                      self._majoraxis)
        self._normal = new

    @property
    def majoraxis(self):
        return self._majoraxis

    @majoraxis.setter
    def majoraxis(self, NewMajorAxis):
        if not (isinstance(NewMajorAxis, np.ndarray) and len(NewMajorAxis) == 3 and np.linalg.norm(NewMajorAxis) > 0):
            raise TypeError("MajorAxis must be a 3D numpy.ndarray with finite length.")
        if abs(np.dot(self.normal, mgeo.Normalize(NewMajorAxis))) > 1e-12:
            raise ValueError("The normal and major axis of optical elements need to be orthogonal!")
        self._majoraxis = mgeo.Normalize(NewMajorAxis)

    @property
    def type(self):
        return self._type

    def __hash__(self):
        return hash(tuple(self.position) + tuple(self.normal) + tuple(self.majoraxis)) + hash(self.type)

    # ------------------------------------------------------------------ (mis-)alignment, angles in degrees
    def rotate_pitch_by(self, angle):
        """About normal x majoraxis (ART/ModuleOpticalElement.py:169-185)."""
        axis = np.cross(self.normal, self.majoraxis)
        self.normal = mgeo.RotationAroundAxis(axis, np.deg2rad(angle), self.normal)

    def rotate_roll_by(self, angle):
        """About the major axis (:187-198)."""
        self.normal = mgeo.RotationAroundAxis(self.majoraxis, np.deg2rad(angle), self.normal)

    def rotate_yaw_by(self, angle):
        """About the normal (:200-209)."""
        self.majoraxis = mgeo.RotationAroundAxis(self.normal, np.deg2rad(angle), self.majoraxis)

    def rotate_random_by(self, angle):
        """About a random axis (:211-221)."""
        self.normal = mgeo.RotationAroundAxis(np.random.random(3), np.deg2rad(angle), self.normal)

    def shift_along_normal(self, distance):
        self.position = self.position + distance * self.normal

    def shift_along_major(self, distance):
        self.position = self.position + distance * self.majoraxis

    def shift_along_cross(self, distance):
        self.position = self.position + distance * mgeo.Normalize(np.cross(self.normal, self.majoraxis))

    def shift_along_random(self, distance):
        self.position = self.position + distance * mgeo.Normalize(np.random.random(3))
