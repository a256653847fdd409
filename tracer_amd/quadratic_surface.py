"""
z = a x^2 + b y^2 + c x y + d x + e y + f patches (quad-fitted heliostat facets).
Reference: tracer/quadratic_surface.py:4-18, :64-71.
"""
import numpy as N
from . import _cabi
from .quadric import QuadricGM


class FlatQuadricSurfaceGM(QuadricGM):
    def __init__(self, a=1., b=1., c=1., d=0., e=0., f=0.):
        QuadricGM.__init__(self)
        self.a, self.b, self.c, self.d, self.e, self.f = a, b, c, d, e, f

    def _coeffs(self):
        return [self.a, self.b, self.c, self.d, self.e, self.f]

    def _native(self):
        return _cabi.GM_QUADRATIC, self._coeffs(), []


class RectFlatQuadricSurfaceGM(FlatQuadricSurfaceGM):
    def __init__(self, width, height, a=1., b=1., c=1., d=1., e=1., f=1.):
        FlatQuadricSurfaceGM.__init__(self, a, b, c, d, e, f)
        self._half_dims = N.c_[[width, height]] / 2.
        self._w, self._h = width / 2., height / 2.

    def _native(self):
        return _cabi.GM_QUADRATIC_RECT, self._coeffs() + [self._w, self._h], []
