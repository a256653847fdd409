"""
The PETAL dish of Sede Boqer (hexagonal aperture) in front of a homogenized receiver; same class and arguments as the
reference's tracer/models/PETAL_dish.py:12-50.
"""
from ..paraboloid import HexagonalParabolicDishGM
from .homogenized_local_receiver import DishOnHomogenizedReceiver


class PETAL(DishOnHomogenizedReceiver):
    """
    PETAL(diameter, focal_length, dish_opt_eff, receiver_pos, receiver_side, homogenizer_depth, homog_opt_eff,
    receiver_aspect=1.): diameter of the circle around the hexagonal aperture; the other arguments as tau_minidish.MiniDish.
    """
    aperture = HexagonalParabolicDishGM
