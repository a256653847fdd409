"""
Cones x^2 + y^2 = (c (z - a))^2: infinite, finite, frustum and rectangular-cut frustum
(reference: tracer/cone.py:7-26, :74-87, :261-286, :356-364).  RectCutCone is accepted and, exactly
as in the reference (whose override is mis-named `select_coords`, cone.py:161, and never called),
behaves as a FiniteCone.
"""
import numpy as N
from . import _cabi
from .quadric import QuadricGM


class InfiniteCone(QuadricGM):
    def __init__(self, c, a=0):
        QuadricGM.__init__(self)
        self.c = float(c)
        self.a = float(a)

    def _native(self):
        return _cabi.GM_CONE_INF, [self.c, self.a], []


class FiniteCone(InfiniteCone):
    def __init__(self, r, h):
        if h <= 0. or r <= 0.:
            raise AttributeError
        self.h = float(h)
        self.r = float(r)
        InfiniteCone.__init__(self, c=self.r / self.h)

    def _native(self):
        return _cabi.GM_CONE_FINITE, [self.c, self.a, self.h], []


class RectCutCone(FiniteCone):
    def __init__(self, r, h, wf, hf):
        FiniteCone.__init__(self, r, h)
        self.half_dims = N.array([wf / 2., hf / 2.])


class ConicalFrustum(InfiniteCone):
    def __init__(self, z1, r1, z2, r2):
        r1, r2, z1, z2 = float(r1), float(r2), float(z1), float(z2)
        if r1 <= 0. or r2 <= 0.:
            raise AttributeError
        if r1 == r2 or z1 == z2:
            raise AttributeError
        InfiniteCone.__init__(self, c=float((r2 - r1) / (z2 - z1)), a=float((r2 * z1 - r1 * z2) / (r2 - r1)))
        self.r1, self.r2, self.z1, self.z2 = r1, r2, z1, z2
        self.zmin, self.zmax = N.sort([z1, z2])

    def _native(self):
        return _cabi.GM_FRUSTUM, [self.c, self.a, self.zmin, self.zmax], []


class RectCutConicalFrustum(ConicalFrustum):
    def __init__(self, z1, r1, z2, r2, w, h):
        ConicalFrustum.__init__(self, z1, r1, z2, r2)
        self.half_dims = N.array([w / 2., h / 2.])
        if N.sqrt(N.sum(self.half_dims ** 2)) <= N.amin([r1, r2]):
            raise ValueError('Bad rectangular cut frustum shape, width and height too small')

    def _native(self):
        return _cabi.GM_FRUSTUM_RECTCUT, [self.c, self.a, self.zmin, self.zmax, self.half_dims[0],
                                          self.half_dims[1]], []
