"""
Thick spherical lens as an AssembledObject: two refracting faces (spherical caps cut from CutSphereGM by a
BoundaryPlane, or flat discs) and, where the faces do not meet at the rim, a cylindrical edge.  Same
constructor and placement rule as the reference's tracer/models/spherical_lens.py:19-124: the back principal
point sits at z = 0 and light travels towards -z, so the back focal point is at z = -f.
"""
import numpy as N

from ..object import AssembledObject
from ..surface import Surface
from ..flat_surface import RoundPlateGM
from ..sphere_surface import CutSphereGM
from ..boundary_shape import BoundaryPlane
from ..cylinder import FiniteCylinder
from ..optics_callables import RefractiveHomogenous
from ..spatial_geometry import rotx


def _is_flat(radius):
    return radius is None or radius == 0 or N.isinf(radius)


class SphericalLens(AssembledObject):
    def __init__(self, diameter, depth, R1, R2, refr_idx, transform=None):
        """
        diameter - of the aperture; depth - axial distance between the two faces; R1, R2 - radii of the front
        face (the one met first by rays coming down the Z axis) and of the back face, positive when the centre
        of curvature lies further down the axis, 0 / None / inf for a flat face; refr_idx - index of the glass.
        """
        flip = rotx(N.pi)[:3, :3]

        def face(radius, flat_rotation):
            """(surface, cut plane z in the face's frame or None, radius as a float)"""
            if _is_flat(radius):
                return Surface(RoundPlateGM(diameter / 2.), RefractiveHomogenous(1., refr_idx), rotation=flat_rotation), None, N.inf
            z_cut = N.sqrt(radius ** 2 - diameter ** 2 / 4.)
            if radius > 0:
                plane = BoundaryPlane(location=N.r_[0, 0, z_cut])
            else:
                plane = BoundaryPlane(location=N.r_[0, 0, -z_cut], rotation=flip)
            cap = CutSphereGM(radius=abs(radius), bounding_volume=plane)
            return Surface(geometry=cap, optics=RefractiveHomogenous(1., refr_idx)), plane.get_location()[2], float(radius)

        self._front, cut1, R1 = face(R1, None)
        self._back, cut2, R2 = face(R2, flip)

        # thick-lens power (lensmaker's equation with the depth term)
        power = (refr_idx - 1) * (1. / R1 - 1. / R2 + depth * (refr_idx - 1) / R1 / R2 / refr_idx)
        self._f = 1. / power
        # the back vertex lies this far above the back principal point (z = 0)
        back_vertex = self._f * depth * (refr_idx - 1) / refr_idx / R1

        edge_height = 0.
        edge_centre = 0.
        if cut2 is not None:
            centre_b = back_vertex - R2            # centre of curvature of the back face
            self._back.set_location(N.r_[0., 0., centre_b])
            edge_centre += (centre_b + cut2) / 2.
            edge_height -= centre_b + cut2
        if cut1 is not None:
            centre_f = back_vertex + depth - R1
            self._front.set_location(N.r_[0., 0., centre_f])
            edge_centre += (centre_f + cut1) / 2.
            edge_height += centre_f + cut1

        surfs = [self._front, self._back]
        if edge_height > 0:
            self._cyl = Surface(FiniteCylinder(diameter, edge_height), RefractiveHomogenous(refr_idx, 1.),
                                location=N.r_[0., 0., edge_centre])
            surfs.append(self._cyl)
        AssembledObject.__init__(self, surfs=surfs, transform=transform)

    def focal_length(self):
        """Effective focal length: distance from the back principal point (z = 0) to the back focal point."""
        return self._f
