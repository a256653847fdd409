"""
Rectangular flux homogenizer: a duct of four inward-facing one-sided mirrors standing on z = 0
(same factory and arguments as the reference's tracer/models/homogenizer.py:13-43).
"""
import numpy as N

from ..assembly import Assembly
from .one_sided_mirror import rect_one_sided_mirror
from .. import spatial_geometry as sp


def rect_homogenizer(aperture_xdim, aperture_ydim, height, opt_eff):
    """
    aperture_xdim, aperture_ydim - inner size of the duct along x and y; height - the walls span z = 0..height;
    opt_eff - reflectivity of each wall.  Returns an Assembly of four objects ordered +x, -x, +y, -y.
    """
    absorb = 1 - opt_eff
    mid = height / 2.
    # (mirror width, mirror height, position of its centre, rotation turning its +z towards the duct axis)
    walls = ((height, aperture_ydim, (aperture_xdim / 2., 0, mid), sp.roty(-N.pi / 2.)),
             (height, aperture_ydim, (-aperture_xdim / 2., 0, mid), sp.roty(N.pi / 2.)),
             (aperture_xdim, height, (0, aperture_ydim / 2., mid), sp.rotx(N.pi / 2.)),
             (aperture_xdim, height, (0, -aperture_ydim / 2., mid), sp.rotx(-N.pi / 2.)))
    objects = []
    for w, h, centre, turn in walls:
        wall = rect_one_sided_mirror(w, h, absorb)
        wall.set_transform(N.dot(sp.translate(*centre), turn))
        objects.append(wall)
    return Assembly(objects=objects)
