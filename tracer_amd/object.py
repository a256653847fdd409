"""
AssembledObject: a rigid set of surfaces (and optional bounding shapes for the Kd-tree).
Interface of the reference's tracer/object.py:7-112.
"""
import numpy as N
from .assembly import Assembly


class AssembledObject(Assembly):
    def __init__(self, surfs=None, bounds=None, location=None, rotation=None, transform=None):
        self.surfaces = [] if surfs is None else surfs
        self.boundaries = [] if bounds is None else bounds
        if not hasattr(self.surfaces, '__len__'):
            self.surfaces = [self.surfaces]
        if not hasattr(self.boundaries, '__len__'):
            self.boundaries = [self.boundaries]
        if transform is None:
            transform = N.eye(4)
            if location is not None:
                transform[:3, 3] = location
            if rotation is not None:
                transform[:3, :3] = rotation
        self.set_transform(transform)

    def get_surfaces(self):
        return self.surfaces

    def get_objects(self):
        return [self]

    def add_surface(self, surface):
        self.surfaces.append(surface)
        self.transform_children()

    def add_boundary(self, boundary):
        self.boundaries.append(boundary)
        self.transform_children()

    def get_boundaries(self):
        if self.boundaries is None:
            self.boundaries = []
        return self.boundaries

    def transform_children(self, assembly_transform=N.eye(4)):
        mine = N.dot(assembly_transform, self.get_transform())
        for group in (self.surfaces, self.boundaries):
            if hasattr(group, 'transform_frames'):      # the faces of a mesh kept as arrays (face_set.py): moved in one go
                group.transform_frames(mine)
            else:
                for child in group:
                    child.transform_frame(mine)

    def own_rays(self, rays, surface_id):
        """Default: the object claims no ray (object.py:81-95)."""
        return N.zeros(rays.get_num_rays(), dtype=bool)

    def surfaces_for_next_iteration(self, rays, surface_id):
        """Default: every surface of the object is relevant for every ray (object.py:97-112)."""
        return N.ones((len(self.surfaces), rays.get_num_rays()), dtype=bool)
