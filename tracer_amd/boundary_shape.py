"""
Bounding shapes attached to objects.  BoundaryBox (a local AABB whose global AABB is refreshed on
every transform_frame) is the input of the Kd-tree builder; BoundarySphere / BoundaryCylinder /
BoundaryPlane trim a CutSphereGM.  Interface of the reference's tracer/boundary_shape.py:7-162.
"""
import numpy as N
from .has_frame import HasFrame
from .vector_manipulations import AABB


class BoundaryShape(HasFrame):
    def __init__(self, location=None, rotation=None):
        HasFrame.__init__(self, location, rotation)

    def in_bounds(self, points):
        raise TypeError("Virtual function in_bounds() called. Implement this in a derived class")


class BoundaryBox(BoundaryShape):
    def __init__(self, aabb, location=None, rotation=None):
        """aabb: [[minx, miny, minz], [maxx, maxy, maxz]] in the owner's local frame."""
        BoundaryShape.__init__(self, location, rotation)
        self._aabb = N.array(aabb)
        self._AABB = N.array(aabb)

    def update_AABB(self):
        corners = getattr(self, '_corners', None)
        if corners is None or self._corners_of is not self._aabb:       # the eight corners of the local box, homogeneous: made once
            lo, hi = self._aabb
            k = N.arange(8)
            corners = N.ones((4, 8))
            corners[0] = N.where(k & 1, hi[0], lo[0])
            corners[1] = N.where(k & 2, hi[1], lo[1])
            corners[2] = N.where(k & 4, hi[2], lo[2])
            self._corners, self._corners_of = corners, self._aabb
        glob = N.dot(self._temp_frame[:3], corners)
        self._minpoint, self._maxpoint = glob.min(axis=1), glob.max(axis=1)
        self._AABB = N.array([self._minpoint, self._maxpoint])

    def in_bounds(self, bund_vertices):
        return N.logical_and(bund_vertices > self._minpoint, bund_vertices < self._maxpoint).all(axis=0)

    def transform_frame(self, transform):
        BoundaryShape.transform_frame(self, transform)
        self.update_AABB()


class BoundarySphere(BoundaryShape):
    """Points within `radius` of the shape's location are in bounds (boundary_shape.py:89-110)."""
    def __init__(self, location=None, radius=1.):
        BoundaryShape.__init__(self, location, None)
        self._radius = radius

    def in_bounds(self, bund_vertices):
        return self._radius ** 2 >= ((N.asarray(bund_vertices) - self._temp_frame[:3, 3]) ** 2).sum(axis=1)

    def _native(self):
        return 2, self._radius


class BoundaryCylinder(BoundaryShape):
    """An infinite cylinder along the shape's local Z axis (boundary_shape.py:130-149)."""
    def __init__(self, diameter=1., location=None, rotation=None):
        self._R = diameter / 2.
        BoundaryShape.__init__(self, location, rotation)

    def in_bounds(self, vertices):
        v = N.asarray(vertices)
        local_xy = N.dot(N.linalg.inv(self._temp_frame)[:2], N.vstack((v.T, N.ones(v.shape[0]))))
        return N.sum(local_xy ** 2, axis=0) <= self._R ** 2

    def _native(self):
        return 3, self._R


class BoundaryPlane(BoundaryShape):
    """The half space on the positive local Z side of the shape's XY plane (boundary_shape.py:151-162)."""
    def in_bounds(self, vertices):
        v = N.asarray(vertices)
        local_z = N.dot(N.linalg.inv(self._temp_frame)[2], N.vstack((v.T, N.ones(v.shape[0]))))
        return local_z >= 0

    def _native(self):
        return 1, 0.
