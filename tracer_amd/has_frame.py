"""
Frame bookkeeping shared by surfaces, objects, assemblies and boundaries.
Same contract as the reference's tracer/has_frame.py:5-75: `_transform` is the frame relative to
the parent, `_temp_frame` the frame in global coordinates after transform_frame(parent_frame).
"""
import numpy as N


class HasFrame(object):
    def __init__(self, location=None, rotation=None):
        self._transform = N.eye(4)
        self.set_location(N.zeros(3) if location is None else location)
        self.set_rotation(N.eye(3) if rotation is None else rotation)
        self._temp_frame = self._transform

    def get_location(self):
        return self._loc

    def get_rotation(self):
        return self._rot

    def set_location(self, location):
        location = N.asarray(location)
        if location.shape != (3,):
            raise ValueError("location must be a 1D 3-component array")
        self._loc = location
        self._transform[:3, 3] = location

    def set_rotation(self, rotation):
        rotation = N.asarray(rotation)
        if rotation.shape != (3, 3):
            raise ValueError("rotation must be a 3x3 array")
        self._rot = rotation
        self._transform[:3, :3] = rotation

    def set_transform(self, transform):
        self._transform = transform
        self._loc = transform[:3, 3]
        self._rot = transform[:3, :3]

    def get_transform(self):
        return self._transform

    def transform_frame(self, transform):
        """Global frame = parent frame x own frame."""
        self._temp_frame = N.dot(transform, self._transform)
