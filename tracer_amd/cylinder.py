"""
Cylinders about the local z axis: infinite, finite (height and angular range), rectangular-cut.
Constructors as in the reference's tracer/cylinder.py (:12-18, :59-68, :161-168).
"""
import numpy as N
from . import _cabi
from .quadric import QuadricGM


class InfiniteCylinder(QuadricGM):
    def __init__(self, diameter):
        self._R = diameter / 2.
        QuadricGM.__init__(self)

    def _native(self):
        return _cabi.GM_CYL_INF, [self._R], []


class FiniteCylinder(InfiniteCylinder):
    def __init__(self, diameter, height, ang_range=[0., 2. * N.pi]):
        self._half_h = height / 2.
        assert len(ang_range) == 2
        self._ang_range = ang_range
        InfiniteCylinder.__init__(self, diameter)

    def _native(self):
        return _cabi.GM_CYL_FINITE, [self._R, self._half_h, self._ang_range[0], self._ang_range[1]], []

    def get_fluxmap(self, eners, local_coords, resolution):
        """Energy per area on a (z, azimuth) grid of the wall."""
        zs = N.linspace(-self._half_h, self._half_h, resolution + 1)
        angs = N.linspace(self._ang_range[0], self._ang_range[1], resolution + 1)
        az = N.arctan2(local_coords[1], local_coords[0])
        az[az < 0.] += 2. * N.pi
        h = N.histogram2d(local_coords[2], az, bins=[zs, angs], weights=eners)[0]
        areas = N.diff(zs)[:, None] * (self._R * N.diff(angs))[None, :]
        return N.hstack(h / areas)


class RectCutCylinder(FiniteCylinder):
    def __init__(self, diameter, height, w, h):
        FiniteCylinder.__init__(self, diameter, height)
        self.half_dims = N.array([w / 2., h / 2.])
        if N.sqrt(N.sum(self.half_dims ** 2)) <= self._R:
            raise ValueError('Bad rectangular cut cylinder shape, width and height too small')

    def _native(self):
        return _cabi.GM_CYL_RECTCUT, [self._R, self._half_h, self.half_dims[0], self.half_dims[1]], []
