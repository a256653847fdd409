"""
Host-side helpers with the names of the reference's ray_trace_utils/vector_manipulations.py that
scene construction needs (AABB :92-102).  The per-ray minimal rotation `rotate_z_to_normal`
(:56-74), a top CPU hotspot of the reference, lives on the device (trc_rotate_z_to_normal).
"""
import numpy as N


def AABB(vecs):
    """Axis-aligned bounding box of the columns of a (3,n) array: (min point, max point)."""
    return N.amin(vecs, axis=1), N.amax(vecs, axis=1)
