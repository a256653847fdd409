"""
Spheres: full sphere, lower hemisphere, rectangular spherical facet (reference:
tracer/sphere_surface.py:9-68, :117-139, :206-228).  CutSphereGM (a sphere trimmed by an arbitrary
BoundaryShape, :168-204) is not in the native table -- the reference's own version cannot run on
Python 3 (xrange, :198).
"""
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
