"""
Bounding shapes attached to objects.  BoundaryBox (a local AABB whose global AABB is refreshed on
every transform_frame) is the input of the Kd-tree builder; interface of the reference's
tracer/boundary_shape.py:7-87.
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
        lo, hi = self._aabb
        corners = N.ones((4, 8))
        for k in range(8):
            corners[0, k] = hi[0] if (k & 1) else lo[0]
            corners[1, k] = hi[1] if (k & 2) else lo[1]
            corners[2, k] = hi[2] if (k & 4) else lo[2]
        glob = N.dot(self._temp_frame, corners)[:3]
        self._minpoint, self._maxpoint = N.array(AABB(glob))
        self._AABB = N.array([self._minpoint, self._maxpoint])

    def in_bounds(self, bund_vertices):
        return N.logical_and(bund_vertices > self._minpoint, bund_vertices < self._maxpoint).all(axis=0)

    def transform_frame(self, transform):
        BoundaryShape.transform_frame(self, transform)
        self.update_AABB()
