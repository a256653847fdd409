"""
GeometryManager: the geometry half of the trace protocol.

`GeometryManager` is the abstract interface of the reference (tracer/geometry_manager.py:8-71): a
user subclass that implements find_intersections / select_rays / get_normals /
get_intersection_points_global in Python keeps working with engine='protocol'.

`NativeGeometryManager` is the base of every geometry in the native kind table.  A native GM only
holds parameters; `_native()` returns (gm_kind, params, extra) for the device scene table, and the
protocol methods run the HIP kernels of that kind through the C-ABI (trc_gm_find_intersections,
trc_gm_get_normals), so that unit-level use goes through the same device code as the fused engines.
"""
import ctypes as C
import numpy as N
from . import _cabi


class GeometryManager(object):
    def find_intersections(self, frame, ray_bundle):
        self._working_frame = frame
        self._working_bundle = ray_bundle
        if type(self) is GeometryManager:
            raise TypeError("Find intersections must be extended by a base class")

    def up(self):
        """The surface's local z axis in global coordinates."""
        return self._working_frame[:3, 2]

    def done(self):
        if hasattr(self, '_working_frame'):
            del self._working_frame
            del self._working_bundle

    def select_rays(self, idxs):
        pass

    def get_normals(self):
        pass

    def get_intersection_points_global(self):
        pass

    def get_scene_graph(self, resolution=None):
        return self.mesh(resolution)


def fill_desc(desc, frame, gm_kind, gm_params, optics_kind=_cabi.OPT_TRANSPARENT, opt_params=(), flags=0,
              extra_off=-1, extra_len=0):
    """Write one trc_surface_desc."""
    desc.gm_kind = int(gm_kind)
    desc.optics_kind = int(optics_kind)
    desc.flags = int(flags)
    desc.extra_off = int(extra_off)
    desc.extra_len = int(extra_len)
    desc.reserved = 0
    fr = N.asarray(frame, dtype=float)
    for r in range(3):
        for k in range(4):
            desc.frame[4 * r + k] = fr[r, k]
    for i in range(16):
        desc.gm[i] = float(gm_params[i]) if i < len(gm_params) else 0.0
    for i in range(8):
        desc.opt[i] = float(opt_params[i]) if i < len(opt_params) else 0.0
    return desc


class NativeGeometryManager(GeometryManager):
    """Geometry whose intersection, aperture and normal rules are implemented in trc_core.h."""

    def _native(self):
        """(gm_kind, list of float params, list of extra floats)"""
        raise NotImplementedError

    def _desc(self, frame):
        kind, params, extra = self._native()
        desc = _cabi.SurfaceDesc()
        fill_desc(desc, frame, kind, params, extra_off=0 if len(extra) else -1, extra_len=len(extra))
        return desc, _cabi.f64(extra)

    def find_intersections(self, frame, ray_bundle):
        GeometryManager.find_intersections(self, frame, ray_bundle)
        ctx = _cabi.get_context()
        desc, extra = self._desc(frame)
        cols = ray_bundle.columns_soa(need_energy=False)
        n = cols['x'].shape[0]
        rays = _cabi.make_rays(n, cols['x'], cols['y'], cols['z'], cols['dx'], cols['dy'], cols['dz'])
        t = N.empty(n)
        hits = N.empty((3, n))
        _cabi.check(ctx.lib.trc_gm_find_intersections(
            ctx.handle, C.byref(desc), len(extra), _cabi.ptr(extra) if len(extra) else None, C.byref(rays),
            _cabi.ptr(t), _cabi.ptr(hits[0]), _cabi.ptr(hits[1]), _cabi.ptr(hits[2])))
        self._params = t
        self._global_all = hits
        return t

    def select_rays(self, idxs):
        self._idxs = idxs
        self._global = self._global_all[:, idxs].copy()

    def get_normals(self):
        ctx = _cabi.get_context()
        desc, _ = self._desc(self._working_frame)
        d = _cabi.f64(self._working_bundle.get_directions()[:, self._idxs])
        h = _cabi.f64(self._global)
        n = h.shape[1]
        out = N.empty((3, n))
        _cabi.check(ctx.lib.trc_gm_get_normals(
            ctx.handle, C.byref(desc), n, _cabi.ptr(h[0]), _cabi.ptr(h[1]), _cabi.ptr(h[2]), _cabi.ptr(d[0]),
            _cabi.ptr(d[1]), _cabi.ptr(d[2]), _cabi.ptr(out[0]), _cabi.ptr(out[1]), _cabi.ptr(out[2])))
        return out

    def get_intersection_points_global(self):
        return self._global

    def done(self):
        for a in ('_global', '_global_all', '_idxs', '_params'):
            if hasattr(self, a):
                delattr(self, a)
        GeometryManager.done(self)
