"""
The ANU SG4 big dish as two nested parabolic dishes with different slope errors (inner zone lifted by 0.1 mm so that
it is met first); same class, arguments and attributes as the reference's tracer/models/SG4.py:14-61.
"""
import numpy as N

from ..assembly import Assembly
from ..object import AssembledObject
from ..surface import Surface
from ..paraboloid import ParabolicDishGM
from ..optics_callables import RealReflectiveReceiver
from ..spatial_geometry import translate

EFFECTIVE_MIRROR_AREA = 489.          # m2 of glass on the SG4 frame


class SG4(Assembly):
    def __init__(self, dishDiameter, dishFocus, absMirrors, sigma, dishDiameter_in=20., sigma_in=1.95e-3):
        """
        dishDiameter, dishFocus - the outer dish (m); absMirrors - absorptivity of the glass, spread over the round
        aperture by the ratio of mirrored to aperture area; sigma - slope error of the outer zone (rad);
        dishDiameter_in, sigma_in - diameter and slope error of the inner zone.
        """
        aperture_area = N.pi * (dishDiameter / 2.) ** 2
        self.dishDiameter = dishDiameter
        self.dishFocus = dishFocus
        self.absDish = 1. - (1. - absMirrors) * EFFECTIVE_MIRROR_AREA / aperture_area
        self.sigma = sigma
        Assembly.__init__(self)
        zones = ((dishDiameter, sigma, None), (dishDiameter_in, sigma_in, translate(z=0.0001)))
        for diameter, slope_error, lift in zones:
            mirror = Surface(ParabolicDishGM(diameter, dishFocus), RealReflectiveReceiver(self.absDish, slope_error))
            self.add_object(AssembledObject(surfs=[mirror], transform=lift))

    def get_all_hits(self):
        """hit points (3, n) and absorbed energies (n,) of both zones; also kept as .hits, .abs and .total_abs"""
        per_surface = [s.get_optics_manager().get_all_hits() for s in self.get_surfaces()]
        self.abs = N.hstack([h[0] for h in per_surface])
        self.hits = N.hstack([h[1] for h in per_surface])
        self.total_abs = N.sum(self.abs)
        return self.hits, self.abs
