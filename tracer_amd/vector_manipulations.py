"""
Host-side helpers with the names of the reference's ray_trace_utils/vector_manipulations.py that
scene construction needs (AABB :92-102).  The per-ray minimal rotation `rotate_z_to_normal`
(:56-74), a top CPU hotspot of the reference, lives on the device (trc_rotate_z_to_normal).
"""
import numpy as N


def AABB(vecs):
    """Axis-aligned bounding box of the columns of a (3,n) array: (min point, max point)."""
    return N.amin(vecs, axis=1), N.amax(vecs, axis=1)


def rotate_z_to_normal(vecs, normals):
    """
    Rotate the columns of vecs (3, n), given about +z, so that `normals` (3, n) or (3,) become their +z, each by the
    minimal rotation in the plane of z and its normal (reference :56-74; the device form is trc_rotate_z_to_normal).
    Host helper for direction samplers, one small rotation matrix per column as in the reference.
    """
    from .spatial_geometry import general_axis_rotation
    vecs = N.asarray(vecs, dtype=float)
    normals = N.asarray(normals, dtype=float)
    if normals.ndim == 1:
        normals = N.tile(normals[:, None], (1, vecs.shape[1]))
    out = N.empty_like(vecs)
    for i in range(vecs.shape[1]):
        n = normals[:, i] / N.sqrt(N.sum(normals[:, i] ** 2))
        axis = N.cross([0., 0., 1.], n)
        norm = N.sqrt(N.sum(axis ** 2))
        angle = N.arccos(N.clip(n[2], -1., 1.))
        if angle == 0.:
            out[:, i] = vecs[:, i]
            continue
        axis = axis / norm if norm > 0. else N.array([1., 0., 0.])      # antiparallel: any axis in the plane (:84-90)
        out[:, i] = N.dot(general_axis_rotation(axis, angle), vecs[:, i])
    return out
