"""
A concentrator with a rectangular homogenizer duct in front of a flat receiver.  The plate sits on the optical axis
(+z of the main reflector) at `receiver_pos`, turned over to look back at the reflector; the duct stands on it and opens
towards the reflector.  Class, arguments and accessors are those of the reference's
tracer/models/homogenized_local_receiver.py:14-83, which tau_minidish.MiniDish and PETAL_dish.PETAL derive from.
"""
import numpy as N

from .. import optics_callables as opt
from ..assembly import Assembly
from ..object import AssembledObject
from ..surface import Surface
from ..spatial_geometry import generate_transform
from .homogenizer import rect_homogenizer
from .one_sided_mirror import one_sided_receiver


def _as_pair(dims):
    return tuple(dims) if isinstance(dims, tuple) else (dims, dims)


class HomogenizedLocalReceiver(Assembly):
    def __init__(self, main_reflector, receiver_pos, receiver_dims, homogenizer_depth, homog_opt_eff):
        """
        main_reflector: the Surface concentrating light towards the receiver.  receiver_pos: axial distance from the
        reflector's vertex to the plate.  receiver_dims: side of a square plate, or the tuple (x side, y side).
        homogenizer_depth, homog_opt_eff: length of the duct and reflectivity of its four walls.
        """
        lx, ly = self._sides = _as_pair(receiver_dims)
        self._rec_pos = receiver_pos
        self._mr = main_reflector
        # half a turn about x, then up the axis: both parts are modelled looking up (+z) and mounted looking down
        mount = generate_transform(N.r_[1., 0., 0.], N.pi, N.c_[[0., 0., receiver_pos]])
        self._rec = one_sided_receiver(lx, ly)
        self._hom = rect_homogenizer(lx, ly, homogenizer_depth, homog_opt_eff)
        for part in (self._rec, self._hom):
            part.set_transform(mount)
        Assembly.__init__(self, objects=[self._rec, AssembledObject(surfs=[main_reflector])], subassemblies=[self._hom])

    def get_receiver_surf(self):
        """the receiver (an object of one surface, which holds the hits)"""
        return self._rec

    def get_homogenizer(self):
        """the duct: an assembly of four one-sided mirrors"""
        return self._hom

    def get_main_reflector(self):
        return self._mr

    def histogram_hits(self, bins=50):
        """
        Energy absorbed on the plate during the traces run so far, binned over the plate: (H, x edges, y edges) in the
        convention of numpy.histogram2d, x along the first axis of H, `bins` cells each way.
        """
        plate, = self._rec.get_surfaces()
        absorbed, where = plate.get_optics_manager().get_all_hits()[:2]
        local = plate.global_to_local(where)
        extent = [(-side / 2., side / 2.) for side in self._sides]
        return N.histogram2d(local[0], local[1], bins, range=extent, weights=absorbed)


class DishOnHomogenizedReceiver(HomogenizedLocalReceiver):
    """
    What the dish collectors of this package share (tau_minidish.MiniDish, PETAL_dish.PETAL): a paraboloidal mirror of the
    aperture shape `aperture` (a geometry manager taking diameter and focal length) as main reflector, and a receiver plate of
    receiver_side (x) by receiver_side * receiver_aspect (y).
    """
    aperture = None

    def __init__(self, diameter, focal_length, dish_opt_eff, receiver_pos, receiver_side, homogenizer_depth, homog_opt_eff,
                 receiver_aspect=1., **surface_options):
        mirror = Surface(self.aperture(diameter, focal_length), opt.Reflective(1 - dish_opt_eff), **surface_options)
        HomogenizedLocalReceiver.__init__(self, mirror, receiver_pos, (receiver_side, receiver_side * receiver_aspect),
                                          homogenizer_depth, homog_opt_eff)
        self._ext_dims = (diameter, receiver_pos)

    def get_external_dimensions(self):
        """(diameter of the dish, height from its vertex to the receiver plate): the cylinder the collector fits in"""
        return self._ext_dims

