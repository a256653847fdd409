"""
Flat triangle with one vertex at the local origin (reference: tracer/triangular_face.py:11-74).
The barycentric trim runs on the GPU (trc_intersect_flat, TRC_GM_TRIANGLE).
"""
import numpy as np
from . import _cabi
from .flat_surface import FiniteFlatGM


class TriangularFace(FiniteFlatGM):
    def __init__(self, verts):
        """verts: 3x2 array, each column a vertex in the local XY plane, CCW from the origin."""
        FiniteFlatGM.__init__(self)
        self.set_vertices(verts)

    def set_vertices(self, verts):
        self._verts = verts

    def _native(self):
        v = np.asarray(self._verts, dtype=float)
        return _cabi.GM_TRIANGLE, [v[0, 0], v[1, 0], v[2, 0], v[0, 1], v[1, 1], v[2, 1]], []

    def mesh(self, resolution=None):
        if resolution is None:
            resolution = 10
        if resolution < 2:
            raise ValueError('Resolution must be >= 2')
        alpha, beta = np.meshgrid(np.linspace(0, 1, resolution), np.linspace(0, 1, resolution))
        v = np.asarray(self._verts, dtype=float)
        x, y, z = alpha * v[:, 1, None, None] * (1 - beta) + alpha * v[:, 0, None, None] * beta
        return x, y, z
