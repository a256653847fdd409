"""
A concentrator with a square homogenizer behind its focus and a receiver (PV panel) closing the homogenizer
(reference: tracer/models/homogenized_local_receiver.py:13-83).  Scene construction on the public API; the flux histogram
of the receiver is the caller-side numpy.histogram2d of the reference -- TracerEngine.set_fluxmap gives the same map
accumulated on the device when the hits are not wanted on the host.
"""
import numpy as N

from .. import spatial_geometry as sp
from ..assembly import Assembly
from ..object import AssembledObject
from .one_sided_mirror import one_sided_receiver
from .homogenizer import rect_homogenizer


class HomogenizedLocalReceiver(Assembly):
    def __init__(self, main_reflector, receiver_pos, receiver_dims, homogenizer_depth, homog_opt_eff):
        """
        main_reflector: the Surface that focuses the rays; receiver_pos: distance along the optical axis (+z) from the
        reflector to the receiver's end surface; receiver_dims: side of the square receiver, or (x, y) sides; homogenizer_depth:
        height of the mirror duct standing on the receiver; homog_opt_eff: reflectivity of each of its mirrors.
        """
        self._sides = receiver_dims if isinstance(receiver_dims, tuple) else (receiver_dims, receiver_dims)
        self._rec_pos = receiver_pos
        # receiver and duct share a frame: on the axis at receiver_pos, turned to face the reflector
        facing_back = N.dot(sp.translate(0, 0, receiver_pos), sp.rotx(N.pi))
        self._rec = one_sided_receiver(*self._sides)
        self._rec.set_transform(facing_back)
        self._hom = rect_homogenizer(self._sides[0], self._sides[1], homogenizer_depth, homog_opt_eff)
        self._hom.set_transform(facing_back)
        self._mr = main_reflector
        Assembly.__init__(self, objects=[self._rec, AssembledObject(surfs=[main_reflector])], subassemblies=[self._hom])

    def get_receiver_surf(self):
        return self._rec

    def get_homogenizer(self):
        return self._hom

    def get_main_reflector(self):
        return self._mr

    def histogram_hits(self, bins=50):
        """
        2-D histogram of the energy absorbed on the receiver in its local x, y after a trace: (H, xbins, ybins), x along the
        first axis, over the receiver's own extent.
        """
        surface = self._rec.get_surfaces()[0]
        energy, points = surface.get_optics_manager().get_all_hits()
        x, y = surface.global_to_local(points)[:2]
        half_x, half_y = self._sides[0] / 2., self._sides[1] / 2.
        return N.histogram2d(x, y, bins, range=([-half_x, half_x], [-half_y, half_y]), weights=energy)
