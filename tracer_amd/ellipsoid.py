"""
Ellipsoids (reference: tracer/ellipsoid.py:5-19, :63-76).  EllipsoidGM keeps the reference's
truncation rule as written: the limits are applied only when at least one of xlim/ylim/zlim is
None (then the given ones are used); when all three are given they are ignored (:71-76).
"""
import numpy as N
from . import _cabi
from .quadric import QuadricGM


class Ellipsoid(QuadricGM):
    def __init__(self, a, b, c):
        QuadricGM.__init__(self)
        self.a = 1. / (a ** 2)
        self.b = 1. / (b ** 2)
        self.c = 1. / (c ** 2)

    def _native(self):
        return _cabi.GM_ELLIPSOID, [self.a, self.b, self.c], []


class EllipsoidGM(Ellipsoid):
    def __init__(self, a, b, c, xlim=None, ylim=None, zlim=None):
        Ellipsoid.__init__(self, a, b, c)
        if xlim is None or ylim is None or zlim is None:
            self.xlim, self.ylim, self.zlim = xlim, ylim, zlim
            self.truncated = True
        else:
            self.xlim, self.ylim, self.zlim = [-a, a], [-b, b], [-c, c]
            self.truncated = False

    def _native(self):
        lims = []
        for lim in (self.xlim, self.ylim, self.zlim):
            if self.truncated and lim is not None:
                lims += [lim[0], lim[1]]
            else:
                lims += [-N.inf, N.inf]
        return _cabi.GM_ELLIPSOID_CUT, [self.a, self.b, self.c] + lims, []
