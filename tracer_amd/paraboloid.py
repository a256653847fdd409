"""
Paraboloids: z = a x^2 + b y^2 with circular / hexagonal / rectangular (optionally off-axis)
apertures, and the parabolic cylinder / trough.  Constructors follow the reference's
tracer/paraboloid.py (:11-28, :71-87, :174-190, :225-255, :328-342, :386-403).
"""
import numpy as N
from . import _cabi
from .quadric import QuadricGM
from .spatial_geometry import roty, rotz


class Paraboloid(QuadricGM):
    def __init__(self, a=1., b=None):
        """z = (x/a)^2 + (y/b)^2 (the arguments are the reference's legacy form)."""
        if b is None:
            b = a
        QuadricGM.__init__(self)
        self.a = 1. / (a ** 2)
        self.b = 1. / (b ** 2)

    def _native(self):
        return _cabi.GM_PARABOLOID, [self.a, self.b], []


class ParabolicDishGM(Paraboloid):
    def __init__(self, diameter, focal_length):
        par_param = 2. * N.sqrt(focal_length)
        Paraboloid.__init__(self, par_param, par_param)
        self._R = float(diameter / 2.)
        self._h = float((diameter / 2. / par_param) ** 2)

    def _native(self):
        return _cabi.GM_PARAB_DISH, [self.a, self.b, self._h], []

    def mesh(self, resolution=None):
        if resolution is None:
            resolution = 40.
        rs = N.r_[0:self._R * (1 + 1. / resolution):self._R / resolution]
        angs = N.r_[0:2 * N.pi * (1. + 1. / resolution):2 * N.pi / resolution][:int(resolution) + 1]
        x = N.outer(rs, N.cos(angs))
        y = N.outer(rs, N.sin(angs))
        return x, y, self.a * x ** 2 + self.b * y ** 2

    def get_fluxmap(self, eners, local_coords, resolution):
        """Polar-bin flux map with paraboloid surface areas (paraboloid.py:151-172)."""
        rads = N.sqrt(N.sum(local_coords[:2] ** 2., axis=0))
        angs = N.arctan2(local_coords[1], local_coords[0])
        angs[angs < 0.] += 2. * N.pi
        r = N.r_[0:self._R * (1. + 1. / resolution):self._R / resolution]
        ang = N.r_[0:2. * N.pi * (1 + 1. / resolution):2 * N.pi / resolution][:resolution + 1]
        h = N.histogram2d(rads, angs, bins=[r, ang], weights=eners)[0]
        grow = (4. * self.a ** 2 * r ** 2 + 1.) ** 1.5
        areas = (grow[1:] - grow[:-1])[:, None] * (ang[1:] - ang[:-1])[None, :]
        return N.hstack(h / areas)


class HexagonalParabolicDishGM(Paraboloid):
    def __init__(self, diameter, focal_length):
        par_param = 2 * N.sqrt(focal_length)
        Paraboloid.__init__(self, par_param, par_param)
        self._R = diameter / 2.

    def _native(self):
        return _cabi.GM_PARAB_HEX, [self.a, self.b, self._R], []


class RectangularParabolicDishGM(Paraboloid):
    def __init__(self, width, height, focal_length, off_axis_normal=None):
        self._off_axis_normal = off_axis_normal
        if off_axis_normal is not None:
            # same construction as paraboloid.py:239-251
            d = focal_length
            theta_n = N.arccos(off_axis_normal[2])
            theta_r = 2. * theta_n
            r = d * N.sin(theta_r)
            zc = d * (1. - N.cos(theta_r)) / 2.
            focal_length = d * N.cos(theta_r) + zc
            phi = N.arctan2(off_axis_normal[1], off_axis_normal[0])
            xc, yc = -r * N.cos(phi), -r * N.sin(phi)
            self._rect_center = -N.array([xc, yc, zc])
            rect_rot = N.dot(roty(-theta_n), rotz(-phi))[:3, :3]
            self._rect_rot = N.dot(rotz(phi)[:3, :3], rect_rot)
        self._w, self._h = width / 2., height / 2.
        self._half_dims = N.c_[[width, height]] / 2
        par_param = 2. * N.sqrt(focal_length)
        Paraboloid.__init__(self, par_param, par_param)

    def _native(self):
        if self._off_axis_normal is None:
            return _cabi.GM_PARAB_RECT, [self.a, self.b, self._w, self._h], []
        p = [self.a, self.b, self._w, self._h] + list(N.ravel(self._rect_rot)) + list(N.ravel(self._rect_center))
        return _cabi.GM_PARAB_RECT_OFFAXIS, p, []


class ParabolicCylinder(QuadricGM):
    def __init__(self, a=1.):
        QuadricGM.__init__(self)
        self.a = 1. / (a ** 2)

    def _native(self):
        return _cabi.GM_PARAB_CYL, [self.a], []


class ParabolicTroughGM(ParabolicCylinder):
    def __init__(self, aperture, focal_length, length):
        par_param = 2. * N.sqrt(focal_length)
        ParabolicCylinder.__init__(self, par_param)
        self._l = length
        self._w = float(aperture)
        self._h = float((aperture / 2. / par_param) ** 2)

    def _native(self):
        return _cabi.GM_PARAB_TROUGH, [self.a, self._l / 2., self._h], []
