"""
A surface given as an indexed set of triangular faces (reference: tracer/models/triangulated_surface.py:7-52): one
TriangularFace Surface per non-degenerate face, in the face's own frame (origin at its first vertex, x along its first
edge, z along the face normal).  The faces are kept as arrays (face_set.FaceSet: origins, rotations, local edges, the shared
optics); `get_surfaces()[k]` makes the Surface of face k when a script asks for it.  For the device they are ordinary native
surfaces (TRC_GM_TRIANGLE), found through the engine's grid.
"""
import numpy as N

from ..object import AssembledObject
from ..face_set import FaceSet


class TriangulatedSurface(AssembledObject):
    def __init__(self, vertices, faces, optics, transform=None):
        """
        vertices: (n, 3) points in the object's frame; faces: (m, 3) integer indices into `vertices`; optics: the optics
        manager shared by the faces; transform: 4x4 frame of the object in its container.  Faces with an edge shorter than
        1e-8 or with collinear vertices are dropped, as in the reference.
        """
        vertices = N.asarray(vertices, dtype=float)
        faces = N.asarray(faces, dtype=int)
        origin = vertices[faces[:, 0]]
        edges = vertices[faces[:, 1:], :] - origin[:, None, :]           # (m, 2, 3)
        lengths = N.sqrt(N.sum(edges ** 2, axis=2))
        keep = N.all(abs(lengths) > 1e-8, axis=1)
        origin, edges, lengths = origin[keep], edges[keep], lengths[keep]

        x_axis = edges[:, 0] / lengths[:, 0, None]
        z_axis = N.cross(x_axis, edges[:, 1])
        with N.errstate(invalid='ignore', divide='ignore'):
            z_axis /= N.sqrt((z_axis ** 2).sum(-1))[:, None]
        keep = N.any(abs(z_axis) > 1e-6, axis=1)
        origin, edges, x_axis, z_axis = origin[keep], edges[keep], x_axis[keep], z_axis[keep]
        y_axis = N.cross(z_axis, x_axis)

        frames = N.concatenate((x_axis[..., None], y_axis[..., None], z_axis[..., None]), axis=2)      # columns = axes
        local_edges = N.einsum('fji,fej->fei', frames, edges)                                            # R^T e per face and edge
        surfs = FaceSet(origin, frames, local_edges, optics=optics)
        # (the reference passes `transform` third positionally, where object.py now has `location`: by keyword here)
        AssembledObject.__init__(self, surfs=surfs, bounds=None, transform=transform)
