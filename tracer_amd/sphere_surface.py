"""
Spheres: full sphere, lower hemisphere, rectangular spherical facet (reference:
tracer/sphere_surface.py:9-68, :117-139, :168-204, :206-228).  CutSphereGM trims the sphere by a
BoundaryPlane / BoundarySphere / BoundaryCylinder given in the surface's own frame.
"""
import numpy as N
from . import _cabi
from .quadric import QuadricGM


class SphericalGM(QuadricGM):
    def __init__(self, radius=1.):
        QuadricGM.__init__(self)
        self.set_radius(radius)

    def get_radius(self):
        return self._rad

    def set_radius(self, rad):
        if rad <= 0:
            raise ValueError("Radius must be positive")
        self._rad = rad

    def _native(self):
        return _cabi.GM_SPHERE, [self._rad], []

    def get_fluxmap(self, eners, local_coords, resolution):
        """Energy per area on a (polar angle, azimuth) grid of resolution x 2*resolution bins (sphere_surface.py:100-115)."""
        ths_bin = N.linspace(0., N.pi, resolution + 1)
        phis_bin = N.linspace(0., 2. * N.pi, resolution * 2 + 1)
        ths = N.arccos(local_coords[2] / self._rad)
        phis = N.arctan2(local_coords[1], local_coords[0])
        phis[phis < 0.] += 2. * N.pi
        h = N.histogram2d(ths, phis, bins=[ths_bin, phis_bin], weights=eners)[0]
        areas = self._rad ** 2. * N.diff(phis_bin)[None, :] * (N.cos(ths_bin[:-1]) - N.cos(ths_bin[1:]))[:, None]
        return N.hstack(h / areas)


class HemisphereGM(SphericalGM):
    """The half of the sphere with local z <= 0."""
    def _native(self):
        return _cabi.GM_HEMISPHERE, [self._rad], []


class SphericalRectFacet(SphericalGM):
    def __init__(self, radius, lx, ly):
        SphericalGM.__init__(self, radius)
        self.lx = lx
        self.ly = ly

    def _native(self):
        return _cabi.GM_SPHERE_RECT, [self._rad, self.lx / 2., self.ly / 2.], []


class CutSphereGM(SphericalGM):
    """
    The part of the sphere inside a bounding volume (a BoundaryShape of boundary_shape.py, placed in the surface's
    frame).  Of two candidate hits the one inside the volume is kept; with both inside, the base-class choice.
    """
    def __init__(self, radius=1., bounding_volume=None):
        SphericalGM.__init__(self, radius)
        self._bound = bounding_volume

    def _native(self):
        if self._bound is None:
            return _cabi.GM_SPHERE, [self._rad], []
        if not hasattr(self._bound, '_native'):
            raise NotImplementedError("CutSphereGM: bounding volumes are BoundaryPlane, BoundarySphere or BoundaryCylinder")
        kind, param = self._bound._native()
        t = self._bound.get_transform()
        return _cabi.GM_SPHERE_CUT, [self._rad, float(kind)] + list(t[:3, :3].ravel()) + list(t[:3, 3]) + [float(param)], []
