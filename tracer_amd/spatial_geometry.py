"""
Rotation and translation helpers (host side, O(#surfaces) work done once per scene).
Mirrors the public functions of the reference's tracer/spatial_geometry.py:8-98 (same names,
arguments and conventions) so scene scripts run unchanged.
"""
import math
import numpy as N


def general_axis_rotation(axis, ang):
    """
    3x3 rotation by `ang` radians about the unit vector `axis` (Rodrigues).  sin/cos are rounded to
    14 decimals like the reference does (spatial_geometry.py:18) so frames built here are bit-equal.
    """
    axis = N.asarray(axis, dtype=float)
    s = N.round(math.sin(ang), decimals=14)
    c = N.round(math.cos(ang), decimals=14)
    x, y, z = axis
    cross = N.array([[0., -z, y], [z, 0., -x], [-y, x, 0.]])
    return N.multiply.outer(axis, axis) * (1 - c) + N.eye(3) * c + cross * s


def rotation_to_z(vecs):
    """
    For each unit vector v: the matrix whose columns are (perp, v x perp, v), with
    perp = unit(v_y, -v_x, 0) or x-hat when v is along z (spatial_geometry.py:24-48).
    Accepts one vector (returns 3x3) or an (n,3) array (returns (n,3,3)).
    """
    v = N.asarray(vecs, dtype=float)
    if v.ndim == 1:         # one vector: the same arithmetic without the array machinery (this runs once per source bundle)
        x, y, z = float(v[0]), float(v[1]), float(v[2])
        px, py = y, -x
        if px == 0. and py == 0.:
            px, py = 1., 0.
        norm = math.sqrt(px * px + py * py + 0.)
        px, py = px / norm, py / norm
        return N.array([[px, y * 0. - z * py, x], [py, z * px - x * 0., y], [0., x * py - y * px, z]])
    v = N.atleast_2d(v)
    perp = N.zeros_like(v)
    perp[:, 0] = v[:, 1]
    perp[:, 1] = -v[:, 0]
    degenerate = N.all(perp == 0., axis=1)
    perp[degenerate] = (1., 0., 0.)
    perp /= N.sqrt(N.sum(perp ** 2., axis=1))[:, None]
    out = N.stack((perp, N.cross(v, perp), v), axis=2)
    return N.squeeze(out)


def generate_transform(axis, angle, translation):
    """4x4 homogeneous transform: rotation about `axis` by `angle`, then `translation` (3x1)."""
    top = N.hstack((general_axis_rotation(axis, angle), translation))
    return N.vstack((top, [0., 0., 0., 1.]))


def _rot(ang, i, j):
    m = N.eye(4)
    s, c = N.sin(ang), N.cos(ang)
    m[i, i] = c
    m[j, j] = c
    m[i, j] = -s
    m[j, i] = s
    return m


def rotx(ang):
    """4x4 rotation about x."""
    return _rot(ang, 1, 2)


def roty(ang):
    """4x4 rotation about y."""
    return _rot(ang, 2, 0)


def rotz(ang):
    """4x4 rotation about z."""
    return _rot(ang, 0, 1)


def translate(x=0, y=0, z=0):
    """4x4 translation."""
    m = N.eye(4)
    m[:3, 3] = (x, y, z)
    return m
