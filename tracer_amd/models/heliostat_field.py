"""
Heliostat field: an Assembly of two-axis tracking mirrors.  Same class, methods and arguments as
the reference's tracer/models/heliostat_field.py:20-251 (HeliostatField, track_sun with
azimuth-elevation or tilt-roll tracking, solar_vector, radial_stagger); pure scene construction on
the public API, O(#heliostats) host work.
"""
import numpy as N

from ..assembly import Assembly
from .one_sided_mirror import rect_one_sided_mirror, rect_para_one_sided_mirror, flat_quad_one_sided_mirror
from ..spatial_geometry import rotx, roty, general_axis_rotation
from ..object import AssembledObject
from ..boundary_shape import BoundaryBox


class RotationAxis(AssembledObject):
    """A surface-less object that carries a rotation axis in its frame."""
    def __init__(self, axis=None):
        self.axis = axis
        AssembledObject.__init__(self)

    def get_rotation_axis(self):
        return N.dot(self.get_rotation()[:3, :3], self.axis)


def _unit_rows(vectors):
    """rows scaled to unit length, in place (the reference normalises its aim arrays in place too)"""
    vectors /= N.sqrt(N.sum(vectors ** 2, axis=1)[:, None])
    return vectors


def _in_limits(angle, limits):
    return limits[0] <= angle <= limits[1]


class HeliostatField(Assembly):
    def __init__(self, positions, width, height, absorptivity, sigma, bi_var=True, focal_lengths=None,
                 quad_params=None, MCRT_option='fast',
                 rotation_axes_pos=N.array([[0., 0., 0.], [0., 0., 0.]]),
                 rotation_axes_vec=N.array([[0., 0., 1.], [1., 0., 0.]])):
        """
        positions: (n,3) pedestal positions; width/height: mirror size; absorptivity: scalar or (n,);
        sigma: slope error (rad); focal_lengths / quad_params: optional per-heliostat curvature.
        Each heliostat is Assembly[primary axis, facet Assembly[mirror, secondary axis]] and carries a
        thin BoundaryBox for the Kd-tree (heliostat_field.py:61-77).
        """
        count = positions.shape[0]
        self._pos = positions
        self.rotation_axes_pos = rotation_axes_pos
        alphas = absorptivity if hasattr(absorptivity, '__len__') else N.ones(count) * absorptivity
        focals = focal_lengths if focal_lengths is not None else [None] * count
        quads = quad_params if quad_params is not None else [None] * count
        half = N.array([width, height]) / 2.

        def mirror_for(k):
            """flat, paraboloidal or quadric facet of heliostat k, with its thin bounding box"""
            assert focals[k] is None or quads[k] is None
            box = BoundaryBox(N.array([[-half[0], -half[1], -1e-6], [half[0], half[1], 1e-6]]))
            common = (alphas[k], sigma, bi_var, MCRT_option)
            if focals[k] is not None:
                return rect_para_one_sided_mirror(width, height, focals[k], *common, bounds=box)
            if quads[k] is not None:
                return flat_quad_one_sided_mirror(width, height, quads[k], *common, bounds=box)
            return rect_one_sided_mirror(width, height, *common, bounds=box)

        self._heliostats = []
        for k in range(count):
            axis_1, axis_2 = RotationAxis(axis=rotation_axes_vec[0]), RotationAxis(axis=rotation_axes_vec[1])
            mirror = mirror_for(k)
            mirror.set_location(rotation_axes_pos[1] - rotation_axes_pos[0])
            facet = Assembly(objects=[mirror, axis_2], location=rotation_axes_pos[0])
            self._heliostats.append(Assembly(objects=[axis_1], subassemblies=[facet], location=positions[k]))
        Assembly.__init__(self, subassemblies=self._heliostats)

    def get_heliostats(self):
        return self._heliostats

    def set_aim_height(self, h):
        self._th = h

    def track_sun(self, azimuth, zenith, aim_points=None, aim_vectors=None, tracking='azimuth_elevation',
                  tracking_error=None, tracking_limits_primary_axis=None, tracking_limits_secondary_axis=None):
        """
        Aim every heliostat so that the sun at (azimuth from North towards East, zenith), in radians,
        is reflected towards its aim point (or along its aim vector).  NB: like the reference
        (heliostat_field.py:114-115) `aim_points` is modified in place.
        """
        if aim_points is not None:
            aim_points -= self._pos + N.sum(self.rotation_axes_pos, axis=0)
            aim = _unit_rows(aim_points)
        elif aim_vectors is not None:
            aim = _unit_rows(aim_vectors)
        else:
            raise ValueError('aim-points or aiming vectors have to be set')
        # mirror normals: bisectors of the sun and aim directions
        normals = _unit_rows(solar_vector(azimuth, zenith) + aim)
        lim_1 = tracking_limits_primary_axis if tracking_limits_primary_axis is not None else [-N.pi, N.pi]
        lim_2 = tracking_limits_secondary_axis if tracking_limits_secondary_axis is not None else [-N.pi, N.pi]
        count = self._pos.shape[0]
        # pointing errors, two draws per heliostat in heliostat order (as the reference draws them)
        errors = N.zeros((count, 2)) if tracking_error is None else N.random.normal(scale=tracking_error, size=(count, 2))

        if tracking == 'azimuth_elevation':
            first = N.arctan2(normals[:, 1], normals[:, 0]) + errors[:, 0]
            second = N.arccos(normals[:, 2]) + errors[:, 1]
            for k, heliostat in enumerate(self._heliostats):
                a_az, a_ze = first[k], second[k]
                if a_az < -N.pi:          # as in the reference (:147-150): half a turn, not a full one
                    a_az += N.pi
                if a_az > N.pi:
                    a_az -= N.pi
                for angle, limits in ((a_az, lim_1), (a_ze, lim_2)):
                    if not _in_limits(angle, limits):
                        print(angle, 'is outside of tracking limits')
                        break
                else:
                    facet = heliostat.get_assemblies()[0]
                    axis_1 = heliostat.get_local_objects()[0]
                    facet.set_rotation(general_axis_rotation(axis_1.get_rotation_axis(), N.pi / 2. + a_az))
                    mirror, axis_2 = facet.get_objects()
                    mirror.set_rotation(general_axis_rotation(axis_2.get_rotation_axis(), a_ze))
        elif tracking == 'tilt_roll':
            first = N.arctan2(normals[:, 1], normals[:, 2]) + errors[:, 0]
            second = N.arcsin(normals[:, 0]) + errors[:, 1]
            for k, heliostat in enumerate(self._heliostats):
                if _in_limits(first[k], lim_1) and _in_limits(second[k], lim_2):
                    heliostat.set_rotation(N.dot(rotx(-first[k])[:3, :3], roty(second[k])[:3, :3]))
        # re-run the frame propagation from the root (the reference re-initialises for the same reason, :192)
        Assembly.__init__(self, subassemblies=self._heliostats)

    def get_tracking_vectors(self):
        return [N.dot(h.get_rotation(), N.vstack([0., 0., 1.])) for h in self.get_heliostats()]


def solar_vector(azimuth, zenith):
    """Unit vector towards the sun; azimuth from North (+y) towards East (+x), zenith from +z; radians."""
    az = N.pi / 2. - azimuth
    if az < 0.:
        az += 2 * N.pi
    return N.r_[N.sin(zenith) * N.cos(az), N.sin(zenith) * N.sin(az), N.cos(zenith)]


def radial_stagger(start_ang, end_ang, az_space, rmin, rmax, r_space):
    """(n,2) x,y positions of a radially staggered field (heliostat_field.py:222-251)."""
    rs = N.r_[rmin:rmax:r_space]
    angs = N.r_[start_ang:end_ang:az_space / 2]
    xs = N.r_[N.outer(rs[::2], N.cos(angs[::2])).flatten(), N.outer(rs[1::2], N.cos(angs[1::2])).flatten()]
    ys = N.r_[N.outer(rs[::2], N.sin(angs[::2])).flatten(), N.outer(rs[1::2], N.sin(angs[1::2])).flatten()]
    return N.vstack((xs, ys)).T


def field_losses(transfer, heliostat_surfaces, receiver_surfaces=(), flux=None, projected_areas=None):
    """
    The per-heliostat results of the reference's NSTTF example (examples/Sandia_NSTTF_field example.py:229-290) read off
    the surface-to-surface transfer matrix of TracerEngine.get_transfer_matrix() (last row = the source):
      incoming[h]    energy of the source rays whose first hit is heliostat h (:283),
      blocking[h]    energy of the rays reflected by heliostat h that land on another heliostat (:277),
      to_receiver[h] energy heliostat h delivers to the receiver surfaces,
      shading[h]     flux * projected_areas[h] - incoming[h] when both are given (:288).
    heliostat_surfaces / receiver_surfaces: surface indices in Assembly.get_surfaces() order.
    """
    T = N.asarray(transfer)
    hel = N.asarray(heliostat_surfaces, dtype=int)
    rec = N.asarray(receiver_surfaces, dtype=int)
    res = dict(incoming=T[-1, hel].copy(), blocking=T[N.ix_(hel, hel)].sum(axis=1),
               to_receiver=T[N.ix_(hel, rec)].sum(axis=1) if len(rec) else N.zeros(len(hel)))
    if flux is not None and projected_areas is not None:
        res['shading'] = flux * N.asarray(projected_areas, dtype=float) - res['incoming']
    return res
