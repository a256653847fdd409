"""
Surface: frame + geometry manager + optics callable, and the four-step trace protocol
(register_incoming / select_rays / get_outgoing / done) of the reference's tracer/surface.py:55-112
and user-doc/trace_protocol.rst.  The fused engines do not call these methods: they read the
surface's parameters through scene.compile_scene().  The protocol stays for unit-level use and for
user-defined geometry/optics plug-ins (engine='protocol').
"""
import numpy as N
from .has_frame import HasFrame


class Surface(HasFrame):
    def __init__(self, geometry, optics, location=None, rotation=None, fixed_color=False):
        HasFrame.__init__(self, location, rotation)
        self._geom = geometry
        self._opt = optics
        self._fixed_color = fixed_color
        self._transparency = 0
        if fixed_color:
            self._fixed_color = fixed_color[:3]
            self._transparency = fixed_color[-1] if len(fixed_color) == 4 else 0

    def get_optics_manager(self):
        return self._opt

    def get_geometry_manager(self):
        return self._geom

    def register_incoming(self, ray_bundle):
        """Keep the bundle, return the parametric hit distance of each ray (+inf = miss)."""
        self._current_bundle = ray_bundle
        return self._geom.find_intersections(self._temp_frame, ray_bundle)

    def select_rays(self, idxs):
        self._selected = idxs
        self._geom.select_rays(idxs)

    def get_outgoing(self):
        return self._opt(self._geom, self._current_bundle, self._selected)

    def update_current_bundle(self, bundle):
        self._current_bundle = bundle

    def done(self):
        if hasattr(self, '_current_bundle'):
            del self._current_bundle
        self._geom.done()

    def global_to_local(self, points):
        """Global -> local with the inverse frame rounded to 9 decimals (surface.py:114-126)."""
        proj = N.round(N.linalg.inv(self._temp_frame), decimals=9)
        return N.dot(proj, N.vstack((points, N.ones(points.shape[1]))))

    def mesh(self, resolution):
        x, y, z = self._geom.mesh(resolution)
        local = N.array((x, y, z, N.ones_like(x)))
        return N.tensordot(self._temp_frame, local, axes=([1], [0]))[:3]
