"""OpticalElement: an optic (mirror or mask) plus its pose in the lab frame, API of
ART/ModuleOpticalElement.py.  Its three vectors are what the kernels' constant frame maps are built from
(ModuleGeometry.frame_maps)."""
import numpy as np

from . import ModuleGeometry as mgeo

_EPS_ORTHO = 1e-12


def _is_vec3(v, nonzero=False):
    return isinstance(v, np.ndarray) and len(v) == 3 and (not nonzero or mgeo._norm(v) > 0)


class OpticalElement:
    """Pose = position of the optic's centre point, unit normal, unit major axis (perpendicular to the normal).
    Assigning a new normal carries the major axis along so that the two stay perpendicular; assigning a major axis
    that is not perpendicular to the normal is an error (ART/ModuleOpticalElement.py:107-160)."""
    __deepcopy__ = mgeo.flat_deepcopy


    def __init__(self, Type, Position, Normal, MajorAxis):
        self._type = Type
        self.position = Position
        self.normal = mgeo.Normalize(Normal)
        self.majoraxis = mgeo.Normalize(MajorAxis)

    type = property(lambda self: self._type)

    @classmethod
    def _like(cls, other, Type):
        """A new element with `other`'s pose (own copies of its three vectors, already validated and normalised there) and
        the optic `Type`: what OEPlacement builds for the chains of a loop list that share a placement step."""
        new = cls.__new__(cls)
        new._type = Type
        new._position = other._position.copy()
        new._normal = other._normal.copy()
        new._majoraxis = other._majoraxis.copy()
        return new

    def _get_position(self):
        return self._position

    def _set_position(self, value):
        if not _is_vec3(value):
            raise TypeError("Position must be a 3D numpy.ndarray.")
        self._position = value

    position = property(_get_position, _set_position)

    def _get_normal(self):
        return self._normal

    def _set_normal(self, value):
        if not _is_vec3(value, nonzero=True):
            raise TypeError("Normal must be a 3D numpy.ndarray with finite length.")
        unit = mgeo.Normalize(value)
        # (during construction there is no major axis yet)
        if hasattr(self, "_majoraxis") and abs(np.dot(unit, self._majoraxis)) > _EPS_ORTHO:
            turn_axis = np.cross(self._normal, value)
            turn_angle = mgeo.AngleBetweenTwoVectors(self._normal, value)
            self._majoraxis = mgeo.RotationAroundAxis(turn_axis, turn_angle, self._majoraxis)
        self._normal = unit

    normal = property(_get_normal, _set_normal)

    def _get_majoraxis(self):
        return self._majoraxis

    def _set_majoraxis(self, value):
        if not _is_vec3(value, nonzero=True):
            raise TypeError("MajorAxis must be a 3D numpy.ndarray with finite length.")
        unit = mgeo.Normalize(value)
        if abs(np.dot(self.normal, unit)) > _EPS_ORTHO:
            raise ValueError("The normal and major axis of optical elements need to be orthogonal!")
        self._majoraxis = unit

    majoraxis = property(_get_majoraxis, _set_majoraxis)

    def _content_hash(self):
        # equal poses hash equal whatever their dtype and the sign of their zeros, like the reference's tuples of numbers
        # (ART/ModuleOpticalElement.py:107-112); through bytes, because this runs for every element of every trace call
        pose = (np.concatenate((self._position, self._normal, self._majoraxis)).astype(float) + 0.0).tobytes()
        return hash(pose) + hash(self.type)

    def __hash__(self):
        return mgeo.memo_hash(self, self._content_hash)

    # ------------------------------------------------------------------ (mis-)alignment, angles in degrees
    def _turn(self, which, axis, angle_deg):
        setattr(self, which, mgeo.RotationAroundAxis(axis, np.deg2rad(angle_deg), getattr(self, which)))

    def rotate_pitch_by(self, angle):
        """Normal turned about normal x majoraxis (:169-185)."""
        self._turn("normal", np.cross(self.normal, self.majoraxis), angle)

    def rotate_roll_by(self, angle):
        """Normal turned about the major axis (:187-198)."""
        self._turn("normal", self.majoraxis, angle)

    def rotate_yaw_by(self, angle):
        """Major axis turned about the normal (:200-209)."""
        self._turn("majoraxis", self.normal, angle)

    def rotate_random_by(self, angle):
        """Normal turned about a random axis, one np.random.random(3) draw (:211-221)."""
        self._turn("normal", np.random.random(3), angle)

    def _shift(self, direction, distance):
        self.position = self.position + distance * direction

    def shift_along_normal(self, distance):
        self._shift(self.normal, distance)

    def shift_along_major(self, distance):
        self._shift(self.majoraxis, distance)

    def shift_along_cross(self, distance):
        self._shift(mgeo.Normalize(np.cross(self.normal, self.majoraxis)), distance)

    def shift_along_random(self, distance):
        """Along a random direction, one np.random.random(3) draw (:252-265)."""
        self._shift(mgeo.Normalize(np.random.random(3)), distance)
