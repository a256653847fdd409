"""
Spheres: full sphere, lower hemisphere, rectangular spherical facet (reference:
tracer/sphere_surface.py:9-68, :117-139, :168-204, :206-228).  CutSphereGM trims the sphere by a
BoundaryPlane / BoundarySphere / BoundaryCylinder given in the surface's own frame.
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
