"""
One-sided mirrors and receivers built on the public API; same factory functions and arguments as
the reference's tracer/models/one_sided_mirror.py:33-107.
"""
from ..object import AssembledObject
from ..surface import Surface
from ..flat_surface import RectPlateGM
from ..paraboloid import RectangularParabolicDishGM
from ..quadratic_surface import RectFlatQuadricSurfaceGM
from .. import optics_callables as opt


def _mirror_optics(option, absorptivity, sigma, bi_var):
    if option == 'fast':
        return opt.OneSidedRealReflective(absorptivity, sigma, bi_var)
    if option == 'receiver':
        return opt.OneSidedRealReflectiveReceiver(absorptivity, sigma, bi_var)
    return opt.OneSidedRealReflectiveDetector(absorptivity, sigma, bi_var)


def rect_one_sided_mirror(width, height, absorptivity=0, sigma=0., bi_var=True, option=None, location=None,
                          rotation=None, bounds=None):
    """Flat rectangular mirror reflecting on its +z side, absorbing everything on the other."""
    optics = _mirror_optics('fast' if option == 'fast' else None, absorptivity, sigma, bi_var)
    surf = Surface(RectPlateGM(width, height), optics)
    return AssembledObject(surfs=[surf], location=location, rotation=rotation, bounds=bounds)


def rect_para_one_sided_mirror(width, height, focal_length, absorptivity=0., sigma=0., bi_var=True, option=None,
                               location=None, rotation=None, bounds=None):
    """Rectangular paraboloidal mirror of the given focal length."""
    optics = _mirror_optics('fast' if option == 'fast' else None, absorptivity, sigma, bi_var)
    surf = Surface(RectangularParabolicDishGM(width, height, focal_length), optics)
    return AssembledObject(surfs=[surf], location=location, rotation=rotation, bounds=bounds)


def flat_quad_one_sided_mirror(width, height, quad_params, absorptivity=0., sigma=0., bi_var=True, option=None,
                               location=None, rotation=None, bounds=None):
    """Rectangular mirror whose sag is z = a x^2 + b y^2 + c x y + d x + e y + f."""
    a, b, c, d, e, f = quad_params
    optics = _mirror_optics(option, absorptivity, sigma, bi_var)
    surf = Surface(RectFlatQuadricSurfaceGM(width, height, a, b, c, d, e, f), optics)
    return AssembledObject(surfs=[surf], location=location, rotation=rotation, bounds=bounds)


def one_sided_receiver(width, height, absorptivity=1, location=None, rotation=None):
    """Rectangular plate absorbing on its +z side, recording absorbed energy and hit points."""
    front = Surface(RectPlateGM(width, height), opt.OneSidedReflectiveReceiver(absorptivity))
    return AssembledObject(surfs=[front], location=location, rotation=rotation)
