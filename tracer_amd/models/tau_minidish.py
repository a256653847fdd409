"""
The Tel Aviv University miniature dish: a round parabolic dish in front of a homogenized square receiver.
Same names and arguments as the reference's tracer/models/tau_minidish.py:22-103 (sizing rules of Ries et al. 1997).
"""
from math import sqrt, pi

from ..paraboloid import ParabolicDishGM
from .homogenized_local_receiver import DishOnHomogenizedReceiver


class MiniDish(DishOnHomogenizedReceiver):
    """
    MiniDish(diameter, focal_length, dish_opt_eff, receiver_pos, receiver_side, homogenizer_depth, homog_opt_eff,
    receiver_aspect=1.): the dish and its reflectivity; the axial distance from the dish vertex to the receiver plate; the
    plate (receiver_side along x, receiver_side * receiver_aspect along y); depth and wall reflectivity of the duct before it.
    """
    aperture = ParabolicDishGM

    def __init__(self, *args, **kwargs):
        DishOnHomogenizedReceiver.__init__(self, *args, fixed_color=(1., 0., 0.), **kwargs)


def standard_minidish_measures(diameter, concentration, virt_sources):
    """
    Dimensions of a 45 degree rim-angle dish whose homogenizer shows `virt_sources` virtual sources besides the real
    one: returns f (focal length), W (receiver side for the aperture-to-receiver area ratio `concentration`) and
    H (duct depth = receiver distance behind the focus).
    """
    f = diameter / 4. / (sqrt(2) - 1)
    W = diameter / 2. * sqrt(pi / concentration)
    n = virt_sources + 1
    H = n * W * f / (diameter - n * W)
    return f, W, H


def standard_minidish(diameter, concentration, virt_sources, dish_opt_eff=0.9, homog_opt_eff=0.9):
    """A MiniDish sized by standard_minidish_measures(); returns (minidish, f, W, H)."""
    f, W, H = standard_minidish_measures(diameter, concentration, virt_sources)
    return MiniDish(diameter, f, dish_opt_eff, f + H, W, H, homog_opt_eff), f, W, H
