"""
Physical laws of the optics callables as stand-alone functions, with the signatures of the
reference's tracer/optics.py (:13-39 fresnel, :41-81 fresnel_conductor / fresnel_to_attenuating,
:145-157 reflections, :159-192 refractions).
They are evaluated by the same device code the engines use (trc_shade via trc_optics_apply),
not by a NumPy re-implementation.
"""
import ctypes as C
import numpy as N
from . import _cabi
from .geometry_manager import fill_desc


def _apply(optics_kind, opt_params, dirs, normals, energy, ref_index=None, seed=1):
    ctx = _cabi.get_context()
    d = _cabi.f64(dirs)
    nn = _cabi.f64(N.broadcast_to(normals, d.shape))
    n = d.shape[1]
    e = _cabi.f64(N.broadcast_to(energy, (n,)))
    ri = None if ref_index is None else _cabi.f64(N.broadcast_to(ref_index, (n,)))
    zeros = N.zeros((3, n))
    desc = _cabi.SurfaceDesc()
    fill_desc(desc, N.eye(4), _cabi.GM_FLAT_INF, [], optics_kind, opt_params)
    rin = _cabi.make_rays(n, dx=d[0], dy=d[1], dz=d[2], e=e, ref_index=ri)
    m = 2 * n
    o = dict((k, N.empty(m)) for k in ('x', 'y', 'z', 'dx', 'dy', 'dz', 'e', 'ref'))
    par = N.empty(m, dtype=N.int64)
    rout = _cabi.make_rays(m, o['x'], o['y'], o['z'], o['dx'], o['dy'], o['dz'], o['e'], parent=par,
                           ref_index=o['ref'])
    _cabi.check(ctx.lib.trc_optics_apply(
        ctx.handle, C.byref(desc), 0, None, C.byref(rin), _cabi.ptr(zeros[0]), _cabi.ptr(zeros[1]),
        _cabi.ptr(zeros[2]), _cabi.ptr(nn[0]), _cabi.ptr(nn[1]), _cabi.ptr(nn[2]), seed, 1, C.byref(rout)))
    k = rout.n
    return N.vstack((o['dx'][:k], o['dy'][:k], o['dz'][:k])), o['e'][:k], par[:k], o['ref'][:k]


def reflections(ray_dirs, normals):
    """Mirror-law directions; ray_dirs and normals are 3 x n (normals may be 3 x 1)."""
    dirs, _, _, _ = _apply(_cabi.OPT_REFLECTIVE, [0.], ray_dirs, normals, 1.)
    return dirs


def _index_pairs(n1, n2, n):
    a = N.broadcast_to(N.asarray(n1, dtype=float), (n,))
    b = N.broadcast_to(N.asarray(n2, dtype=float), (n,))
    pairs = N.unique(N.vstack((a, b)), axis=1).T
    return a, b, pairs


def refractions(n1, n2, ray_dirs, normals):
    """
    Snell refraction.  Returns (refracted, refr_dirs): a boolean array marking the rays that are not
    totally internally reflected, and the 3 x k directions of those rays.
    """
    ray_dirs = N.asarray(ray_dirs, dtype=float)
    n = ray_dirs.shape[1]
    normals = N.broadcast_to(normals, ray_dirs.shape)
    a, b, pairs = _index_pairs(n1, n2, n)
    refracted = N.zeros(n, dtype=bool)
    out = N.zeros((3, n))
    for pa, pb in pairs:
        sel = N.nonzero((a == pa) & (b == pb))[0]
        dirs, _, par, _ = _apply(_cabi.OPT_REFRACTIVE_HOMOGENOUS, [pa, pb, 0., -1.], ray_dirs[:, sel],
                                 normals[:, sel], 1., ref_index=pa)
        second = slice(len(sel), None)      # refracted block follows the reflected block
        refracted[sel[par[second]]] = True
        out[:, sel[par[second]]] = dirs[:, second]
    return refracted, out[:, refracted]


def fresnel(ray_dirs, normals, n1, n2):
    """Unpolarised Fresnel reflectance of each ray (1 where totally internally reflected)."""
    ray_dirs = N.asarray(ray_dirs, dtype=float)
    n = ray_dirs.shape[1]
    normals = N.broadcast_to(normals, ray_dirs.shape)
    a, b, pairs = _index_pairs(n1, n2, n)
    R = N.ones(n)
    for pa, pb in pairs:
        sel = N.nonzero((a == pa) & (b == pb))[0]
        _, e, par, _ = _apply(_cabi.OPT_REFRACTIVE_HOMOGENOUS, [pa, pb, 0., -1.], ray_dirs[:, sel],
                              normals[:, sel], 1., ref_index=pa)
        R[sel] = e[:len(sel)]               # reflected block carries E*R with E = 1
    return R


def fresnel_to_attenuating(n1, m2, theta1):
    """
    Interface between a perfect dielectric (index n1) and an absorbing medium of complex index m2, incidence
    angles theta1 (radians).  Returns R_p, R_s, theta2 (the parallel / perpendicular reflectances and the
    refraction angle), as optics.py:63-81.
    """
    th = _cabi.f64(N.atleast_1d(theta1))
    n = len(th)
    m = N.broadcast_to(N.asarray(m2, dtype=complex), (n,))
    mre, mim = _cabi.f64(m.real), _cabi.f64(m.imag)
    rp, rs, t2 = N.empty(n), N.empty(n), N.empty(n)
    ctx = _cabi.get_context()
    _cabi.check(ctx.lib.trc_optics_fresnel_attenuating(ctx.handle, n, float(n1), _cabi.ptr(mre), _cabi.ptr(mim),
                                                       _cabi.ptr(th), _cabi.ptr(rp), _cabi.ptr(rs), _cabi.ptr(t2)))
    return rp, rs, t2


def fresnel_conductor(ray_dirs, normals, lambdas, material, n1=1., m2=None):
    """Fresnel reflection from a dielectric onto a conductor (optics.py:41-61); material.m(lambdas) gives the complex index."""
    if m2 is None:
        m2 = material.m(lambdas)
    theta_in = N.arccos(N.abs((N.asarray(normals) * N.asarray(ray_dirs)).sum(axis=0)))
    return fresnel_to_attenuating(n1, m2, theta_in)
