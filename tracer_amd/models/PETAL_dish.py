"""
The PETAL dish of Sede Boqer (hexagonal aperture) in front of a homogenized receiver; same class and arguments as the
reference's tracer/models/PETAL_dish.py:12-50.
"""
from .. import optics_callables as opt
from ..surface import Surface
from ..paraboloid import HexagonalParabolicDishGM
from .homogenized_local_receiver import HomogenizedLocalReceiver


class PETAL(HomogenizedLocalReceiver):
    def __init__(self, diameter, focal_length, dish_opt_eff, receiver_pos, receiver_side, homogenizer_depth,
                 homog_opt_eff, receiver_aspect=1.):
        """
        diameter - of the circle around the hexagonal aperture; the other arguments as in tau_minidish.MiniDish
        (receiver_aspect scales the plate's second side).
        """
        dish = Surface(HexagonalParabolicDishGM(diameter, focal_length), opt.Reflective(1 - dish_opt_eff))
        HomogenizedLocalReceiver.__init__(self, dish, receiver_pos, (receiver_side, receiver_side * receiver_aspect),
                                          homogenizer_depth, homog_opt_eff)
        self._ext_dims = (diameter, receiver_pos)

    def get_external_dimensions(self):
        """(diameter, height from the dish vertex to the receiver plate)"""
        return self._ext_dims
