"""
QuadricGM: base of all second-order surfaces.  In the reference (tracer/quadric.py:7-187) a
subclass provides get_ABC / _normals / _select_coords in NumPy; here a subclass provides the kind
and parameters of its row in the native table and the solver (trc_intersect_quadric: discriminant
threshold, stable two-root solve, root-selection rules) runs on the GPU.
"""
from .geometry_manager import NativeGeometryManager


class QuadricGM(NativeGeometryManager):
    pass
