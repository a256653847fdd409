"""
Flat geometries: infinite plane and the finite plates trimmed from it.
Classes, constructor arguments and argument checks follow the reference's tracer/flat_surface.py
(:11-113 plane, :181-211 rect, :253-274 extruded rect, :357-377 perforated rect, :457-492 round,
:548-560 straight-cut round); the intersection rules themselves live in csrc/trc_core.h
(trc_intersect_flat) and run on the GPU.
"""
import numpy as N
from . import _cabi
from .geometry_manager import NativeGeometryManager


class FlatGeometryManager(NativeGeometryManager):
    """Infinite plane z=0 of the local frame."""
    def _native(self):
        return _cabi.GM_FLAT_INF, [], []


class FiniteFlatGM(FlatGeometryManager):
    """Common base of the trimmed plates."""
    def __init__(self):
        FlatGeometryManager.__init__(self)


def _histogram_flux(xs, ys, coords_x, coords_y, eners):
    """Energy per unit area on a rectilinear grid (flat_surface.py:237-251)."""
    h = N.histogram2d(coords_x, coords_y, bins=[xs, ys], weights=eners)[0]
    areas = N.abs(N.diff(xs))[:, None] * N.abs(N.diff(ys))[None, :]
    return N.hstack(h / areas)


class RectPlateGM(FiniteFlatGM):
    """Rectangle of `width` (local x) by `height` (local y) centred on the origin."""
    def __init__(self, width, height):
        if width <= 0:
            raise ValueError("Width must be positive")
        if height <= 0:
            raise ValueError("Height must be positive")
        self.width = width
        self.height = height
        self._half_dims = N.c_[[width, height]] / 2.
        FiniteFlatGM.__init__(self)

    def _native(self):
        return _cabi.GM_RECT, [self._half_dims[0, 0], self._half_dims[1, 0]], []

    def mesh(self, resolution):
        if resolution is None:
            resolution = 40
        xs = N.linspace(-self._half_dims[0, 0], self._half_dims[0, 0], resolution + 1)
        ys = N.linspace(-self._half_dims[1, 0], self._half_dims[1, 0], resolution + 1)
        x, y = N.broadcast_arrays(xs[:, None], ys)
        return x, y, N.zeros_like(x)

    def get_fluxmap(self, eners, local_coords, resolution):
        xs = N.linspace(-self._half_dims[0, 0], self._half_dims[0, 0], resolution + 1)
        ys = N.linspace(-self._half_dims[1, 0], self._half_dims[1, 0], resolution + 1)
        return _histogram_flux(xs, ys, local_coords[0], local_coords[1], eners)


class ExtrudedRectPlateGM(RectPlateGM):
    """Rectangular plate with a rectangular hole."""
    def __init__(self, width, height, extr_center, extr_width, extr_height):
        RectPlateGM.__init__(self, width, height)
        self.extr_center = extr_center
        self.extr_half_dims = N.c_[[extr_width, extr_height]] / 2.
        assert ((extr_center + self.extr_half_dims) < self._half_dims).all()

    def _native(self):
        c = N.ravel(self.extr_center)
        return _cabi.GM_RECT_EXTRUDED, [self._half_dims[0, 0], self._half_dims[1, 0], c[0], c[1],
                                         self.extr_half_dims[0, 0], self.extr_half_dims[1, 0]], []


class PerforatedRectPlateGM(RectPlateGM):
    """Rectangular plate with circular perforations: extr_centers (n,2), extr_radii (n,)."""
    def __init__(self, width, height, extr_centers, extr_radii):
        RectPlateGM.__init__(self, width, height)
        self.extr_centers = extr_centers
        self.extr_radii = extr_radii

    def _native(self):
        c = N.asarray(self.extr_centers, dtype=float).reshape(-1, 2)
        r = N.ravel(N.asarray(self.extr_radii, dtype=float))
        extra = N.column_stack((c[:, 0], c[:, 1], r)).ravel().tolist()
        return _cabi.GM_RECT_PERFORATED, [self._half_dims[0, 0], self._half_dims[1, 0]], extra


class RoundPlateGM(FiniteFlatGM):
    """Disc of radius Re, optionally an annulus from Ri."""
    def __init__(self, Re, Ri=None):
        if Re <= 0.:
            raise ValueError("Radius must be positive")
        if Ri is not None:
            if Ri >= Re:
                raise ValueError("Inner Radius must be lower than the outer one")
            if Ri <= 0.:
                raise ValueError("Radius must be positive")
        self._Ri = Ri
        self._Re = Re
        FiniteFlatGM.__init__(self)

    def _native(self):
        return _cabi.GM_ROUND, [self._Re, -1. if self._Ri is None else self._Ri], []

    def _polar_grid(self, resolution):
        angs = N.r_[0.:2. * N.pi + 2. * N.pi / resolution:2. * N.pi / resolution]
        r0 = 0. if self._Ri is None else self._Ri
        rs = r0 + (self._Re - r0) / resolution * N.arange(0, resolution + 1)
        return rs, angs

    def mesh(self, resolution=None):
        if resolution is None:
            resolution = 40
        rs, angs = self._polar_grid(resolution)
        x = N.outer(rs, N.cos(angs))
        y = N.outer(rs, N.sin(angs))
        return x, y, N.zeros_like(x)

    def get_fluxmap(self, eners, local_coords, resolution):
        """Polar-bin flux map (flat_surface.py:524-545; note the reference's arctan2(x, y) order)."""
        if resolution is None:
            resolution = 40
        rads = N.sqrt(N.sum(local_coords[:2] ** 2, axis=0))
        azim = N.arctan2(local_coords[0], local_coords[1])
        azim[azim < 0.] += 2. * N.pi
        rs, angs = self._polar_grid(resolution)
        h = N.histogram2d(rads, azim, bins=[rs, angs], weights=eners)[0]
        areas = N.diff(rs)[:, None] * ((rs[1:] + rs[:-1]) / 2.)[:, None] * N.abs(N.diff(angs))[None, :]
        return N.hstack(h / areas)


class StraightCutRoundPlateGM(RoundPlateGM):
    """Disc with the part x > x_cut removed."""
    def __init__(self, Re, x_cut=None):
        RoundPlateGM.__init__(self, Re)
        self.x_cut = x_cut

    def _native(self):
        return _cabi.GM_ROUND_CUT, [self._Re, -1. if self._Ri is None else self._Ri,
                                    N.inf if self.x_cut is None else self.x_cut], []
