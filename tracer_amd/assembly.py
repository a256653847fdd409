"""
Assembly: a tree of sub-assemblies and objects, each with a frame.  Same public interface and the
same ordering contract as the reference's tracer/assembly.py:9-151 -- objects of sub-assemblies
come first, own objects last (assembly.py:60-65); surfaces follow object order (:67-77).  That
order defines the surface indices of the flattened device scene and the tie rule of the engine.
"""
import numpy as N
from .has_frame import HasFrame
from .face_set import FaceSet, SurfaceSeq


class Assembly(HasFrame):
    def __init__(self, objects=None, subassemblies=None, location=None, rotation=None):
        self._objects = [] if objects is None else objects
        self._assemblies = [] if subassemblies is None else subassemblies
        HasFrame.__init__(self, location, rotation)

    def global_to_local(self, points):
        proj = N.round(N.linalg.inv(self._temp_frame), decimals=9)
        return N.dot(proj, N.vstack((points, N.ones(points.shape[1]))))

    def get_local_objects(self):
        return self._objects

    def get_assemblies(self):
        return self._assemblies

    def get_objects(self):
        found = []
        for sub in self._assemblies:
            found.extend(sub.get_objects())
        found.extend(self._objects)
        return found

    def get_surfaces(self):
        per_object = [obj.get_surfaces() for obj in self.get_objects()]
        if any(isinstance(p, FaceSet) for p in per_object):     # a mesh that keeps its faces as arrays: nothing is materialised
            return SurfaceSeq(per_object)
        return [s for p in per_object for s in p]

    def add_object(self, object, transform=None):
        self._objects.append(object)
        if transform is not None:
            object.set_transform(transform)
        self.transform_children()

    def add_assembly(self, assembly, transform=None):
        self._assemblies.append(assembly)
        if transform is not None:
            assembly.set_transform(transform)
        self.transform_children()

    def set_rotation(self, rotation):
        HasFrame.set_rotation(self, rotation)
        self.transform_children()

    def set_location(self, location):
        HasFrame.set_location(self, location)
        self.transform_children()

    def set_transform(self, transform):
        HasFrame.set_transform(self, transform)
        self.transform_children()

    def transform_children(self, assembly_transform=N.eye(4)):
        """Push `assembly_transform x own transform` down to every child (assembly.py:135-146)."""
        mine = N.dot(assembly_transform, self.get_transform())
        for child in list(getattr(self, '_assemblies', [])) + list(getattr(self, '_objects', [])):
            child.transform_children(mine)

    def reset_all_optics(self):
        surfaces = self.get_surfaces()
        managers = surfaces.distinct_optics() if isinstance(surfaces, SurfaceSeq) else [s.get_optics_manager() for s in surfaces]
        for opt in managers:
            if hasattr(opt, 'reset'):
                opt.reset()
