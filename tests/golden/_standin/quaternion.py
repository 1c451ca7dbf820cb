"""Stand-in for the third-party `numpy-quaternion` package (absent from this image, no network).

TEST INFRASTRUCTURE ONLY -- used solely by tests/golden/generate_goldens.py so that the
reference (/root/reference, read-only) can be imported in the build container to produce
golden vectors.  It is our own code (textbook Hamilton algebra), not reference source.

Only the surface the reference touches is provided (ART/ModuleGeometry.py:13, :321-329):
    quaternion(x, y, z)        -> pure-vector quaternion (w = 0), as numpy-quaternion does
    quaternion(w, x, y, z)
    np.exp(q)                  -> dispatched by NumPy's object ufunc loop to q.exp()
    np.conjugate(q)            -> dispatched to q.conjugate()
    q1 * q2                    -> Hamilton product
    q.imag                     -> np.array([x, y, z])
"""
import math
import numpy as np


class quaternion:
    __slots__ = ("w", "x", "y", "z")

    def __init__(self, *c):
        if len(c) == 3:
            self.w, self.x, self.y, self.z = 0.0, float(c[0]), float(c[1]), float(c[2])
        elif len(c) == 4:
            self.w, self.x, self.y, self.z = (float(v) for v in c)
        else:
            raise TypeError("quaternion takes 3 or 4 components")

    def exp(self):
        vn = math.sqrt(self.x * self.x + self.y * self.y + self.z * self.z)
        ew = math.exp(self.w)
        if vn == 0.0:
            return quaternion(ew, 0.0, 0.0, 0.0)
        s = ew * math.sin(vn) / vn
        return quaternion(ew * math.cos(vn), s * self.x, s * self.y, s * self.z)

    def conjugate(self):
        return quaternion(self.w, -self.x, -self.y, -self.z)

    def __mul__(self, o):
        return quaternion(
            self.w * o.w - self.x * o.x - self.y * o.y - self.z * o.z,
            self.w * o.x + self.x * o.w + self.y * o.z - self.z * o.y,
            self.w * o.y - self.x * o.z + self.y * o.w + self.z * o.x,
            self.w * o.z + self.x * o.y - self.y * o.x + self.z * o.w,
        )

    @property
    def imag(self):
        return np.array([self.x, self.y, self.z])

    @property
    def real(self):
        return self.w
