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
        n = positions.shape[0]
        self._pos = positions
        if focal_lengths is None:
            focal_lengths = [None] * n
        if quad_params is None:
            quad_params = [None] * n
        if not hasattr(absorptivity, '__len__'):
            absorptivity = N.ones(n) * absorptivity
        self._heliostats = []
        self.rotation_axes_pos = rotation_axes_pos
        offset = rotation_axes_pos[1] - rotation_axes_pos[0]
        for p in range(n):
            primary = RotationAxis(axis=rotation_axes_vec[0])
            secondary = RotationAxis(axis=rotation_axes_vec[1])
            assert not ((focal_lengths[p] is not None) and (quad_params[p] is not None))
            box = BoundaryBox(N.array([[-width / 2., width / 2.], [-height / 2., height / 2.], [-1e-6, 1e-6]]).T)
            if focal_lengths[p] is None and quad_params[p] is None:
                mirror = rect_one_sided_mirror(width, height, absorptivity[p], sigma, bi_var, MCRT_option, bounds=box)
            elif focal_lengths[p] is not None:
                mirror = rect_para_one_sided_mirror(width, height, focal_lengths[p], absorptivity[p], sigma, bi_var,
                                                    MCRT_option, bounds=box)
            else:
                mirror = flat_quad_one_sided_mirror(width, height, quad_params[p], absorptivity[p], sigma, bi_var,
                                                    MCRT_option, bounds=box)
            mirror.set_location(offset)
            facet = Assembly(objects=[mirror, secondary], location=rotation_axes_pos[0])
            self._heliostats.append(Assembly(objects=[primary], subassemblies=[facet], location=positions[p]))
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
        sun_vec = solar_vector(azimuth, zenith)
        if aim_points is None:
            if aim_vectors is None:
                raise ValueError('aim-points or aiming vectors have to be set')
            aim = aim_vectors
            aim /= N.sqrt(N.sum(aim ** 2, axis=1)[:, None])
        else:
            aim_points -= self._pos + N.sum(self.rotation_axes_pos, axis=0)
            aim_points /= N.sqrt(N.sum(aim_points ** 2, axis=1)[:, None])
            aim = aim_points
        trac = sun_vec + aim
        trac /= N.sqrt(N.sum(trac ** 2, axis=1)[:, None])

        if tracking_limits_primary_axis is None:
            tracking_limits_primary_axis = [-N.pi, N.pi]
        if tracking_limits_secondary_axis is None:
            tracking_limits_secondary_axis = [-N.pi, N.pi]
        err1 = err2 = 0.
        if tracking == 'azimuth_elevation':
            az = N.arctan2(trac[:, 1], trac[:, 0])
            ze = N.arccos(trac[:, 2])
            for h in range(self._pos.shape[0]):
                if tracking_error is not None:
                    err1 = N.random.normal(scale=tracking_error)
                    err2 = N.random.normal(scale=tracking_error)
                ang_az = az[h] + err1
                ang_ze = ze[h] + err2
                if ang_az < -N.pi:
                    ang_az += N.pi
                if ang_az > N.pi:
                    ang_az -= N.pi
                if not (tracking_limits_primary_axis[0] <= ang_az <= tracking_limits_primary_axis[1]):
                    print(ang_az, 'is outside of tracking limits')
                    continue
                if not (tracking_limits_secondary_axis[0] <= ang_ze <= tracking_limits_secondary_axis[1]):
                    print(ang_ze, 'is outside of tracking limits')
                    continue
                facet = self._heliostats[h].get_assemblies()[0]
                primary = self._heliostats[h].get_local_objects()[0]
                facet.set_rotation(general_axis_rotation(primary.get_rotation_axis(), N.pi / 2. + ang_az))
                mirror, secondary = facet.get_objects()
                mirror.set_rotation(general_axis_rotation(secondary.get_rotation_axis(), ang_ze))
        elif tracking == 'tilt_roll':
            tilt = N.arctan2(trac[:, 1], trac[:, 2])
            roll = N.arcsin(trac[:, 0])
            for h in range(self._pos.shape[0]):
                if tracking_error is not None:
                    err1 = N.random.normal(scale=tracking_error)
                    err2 = N.random.normal(scale=tracking_error)
                a_t, a_r = tilt[h] + err1, roll[h] + err2
                if not (tracking_limits_primary_axis[0] <= a_t <= tracking_limits_primary_axis[1]):
                    continue
                if not (tracking_limits_secondary_axis[0] <= a_r <= tracking_limits_secondary_axis[1]):
                    continue
                self._heliostats[h].set_rotation(N.dot(rotx(-a_t)[:3, :3], roty(a_r)[:3, :3]))
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
