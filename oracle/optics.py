"""
Optics laws and callables, restated.  Each law takes its random variates as arguments so that it can be
pinned against the reference by replaying numpy.random draws; `shade()` draws them from Philox in the
device's order.
"""
import numpy as N
from .kinds import *
from . import philox


def reflections(d, n):
    """optics.py:145-157"""
    vertical = N.sum(d * n, axis=0) * n
    return d - 2. * vertical


def general_axis_rotation(axis, ang):
    """spatial_geometry.py:8-22 (sin/cos rounded to 14 decimals)"""
    s = N.round(N.sin(ang), decimals=14)
    c = N.round(N.cos(ang), decimals=14)
    v = 1 - c
    add = N.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return N.multiply.outer(axis, axis) * v + N.eye(3) * c + add * s


def rotate_z_to_normal(vecs, normals):
    """ray_trace_utils/vector_manipulations.py:56-90: minimal rotation z -> normal, applied per ray"""
    z = N.zeros(vecs.shape)
    z[2] = 1.
    with N.errstate(invalid='ignore', divide='ignore'):
        axes = N.cross(z.T, normals.T).T
        axes = axes / N.sqrt(N.sum(axes ** 2, axis=0))
    nans = N.isnan(axes[0])
    axes[:, nans] = N.array([[1., 0., 0.]]).T
    angles = N.arccos(normals[2])
    out = N.empty_like(vecs)
    for i in range(vecs.shape[1]):
        if angles[i] != 0.:
            out[:, i] = N.dot(general_axis_rotation(axes[:, i], angles[i]), vecs[:, i])
        else:
            out[:, i] = vecs[:, i]
    return out


def rotation_to_z(vecs):
    """spatial_geometry.py:24-48: (n,3) unit vectors -> (n,3,3) frames with columns (perp, v x perp, v)"""
    vecs = N.atleast_2d(vecs)
    perp = N.hstack((vecs[:, 1][:, None], -vecs[:, 0][:, None], N.zeros((vecs.shape[0], 1))))
    perp[N.all(perp == 0., axis=1)] = N.r_[1., 0., 0.]
    perp /= N.sqrt(N.sum(perp ** 2., axis=1))[:, None]
    return N.concatenate((perp[..., None], N.cross(vecs, perp)[..., None], vecs[..., None]), axis=2)


def pillbox_directions(xi1, xi2, ang_range):
    """sources.py:91-98 with the uniforms given: xi1 in [0,2pi), xi2 in [0,1)"""
    if ang_range == 0.:
        dirs = N.zeros((3, len(xi1)))
        dirs[2] = 1.
        return dirs
    sinsqrt = N.sin(ang_range) * N.sqrt(xi2)
    return N.vstack((N.cos(xi1) * sinsqrt, N.sin(xi1) * sinsqrt, N.sqrt(1. - sinsqrt ** 2.)))


def slope_error_normals(ideal_normals, sigma, bi_var, g0, g1, u):
    """optics_callables.py:234-257; g0,g1 standard normal variates, u uniform [0,1)"""
    if bi_var:
        tanx = N.tan(sigma * g0)
        tany = N.tan(sigma * g1)
        ez = (1. / (1. + tanx ** 2. + tany ** 2.)) ** 0.5
        ex = tanx * ez
        ey = tany * ez
    else:
        th = sigma * g0
        phi = 2. * N.pi * u
        ez = N.cos(th)
        ex = N.sin(th) * N.cos(phi)
        ey = N.sin(th) * N.sin(phi)
    real = rotate_z_to_normal(N.vstack((ex, ey, ez)), ideal_normals)
    return real / N.sqrt(N.sum(real ** 2, axis=0))


def lambertian_directions(normals, xi1, xi2, ang_range):
    """optics_callables.py:163-165"""
    directs = pillbox_directions(xi1, xi2, ang_range)
    return N.sum(rotation_to_z(normals.T) * directs.T[:, None, :], axis=2).T


def refractions(n1, n2, d, n):
    """optics.py:159-192"""
    eta = N.broadcast_arrays(n2 / n1, d[0])[0]
    n = N.broadcast_arrays(n, d)[0]
    cos1 = (n * d).sum(axis=0)
    refracted = cos1 ** 2 >= 1. - eta ** 2
    cos1 = cos1[refracted]
    dd = d[:, refracted]
    nn = n[:, refracted]
    eta = eta[refracted]
    refr = (dd - cos1 * nn) / eta
    cos2 = N.sqrt(1 - 1. / eta ** 2 * (1. - cos1 ** 2))
    refr = refr + nn * cos2 * N.where(cos1 < 0., -1, 1)
    return refracted, refr


def fresnel(d, n, n1, n2):
    """optics.py:13-39"""
    theta_in = N.arccos(N.abs((n * d).sum(axis=0)))
    foo = N.cos(theta_in)
    bar = N.sqrt(1 - (n1 / n2 * N.sin(theta_in)) ** 2)
    Rs = ((n1 * foo - n2 * bar) / (n1 * foo + n2 * bar)) ** 2
    Rp = ((n1 * bar - n2 * foo) / (n1 * bar + n2 * foo)) ** 2
    return (Rs + Rp) / 2


def fresnel_to_attenuating(n1, m2, theta1):
    """optics.py:63-81 as written: R_p, R_s, theta2"""
    b = (m2.real ** 2 - m2.imag ** 2 - (n1 * N.sin(theta1)) ** 2)
    a = N.sqrt(b ** 2 + 4. * (m2.real * m2.imag) ** 2)
    p = N.sqrt(0.5 * (a + b))
    q = N.sqrt(0.5 * (a - b))
    theta2 = N.arctan(n1 * N.sin(theta1) / p)
    R_s = ((n1 * N.cos(theta1) - p) ** 2 + q ** 2) / ((n1 * N.cos(theta1) + p) ** 2 + q ** 2)
    R_p = ((p - n1 * N.sin(theta1) * N.tan(theta1)) ** 2 + q ** 2) / ((p + n1 * N.sin(theta1) * N.tan(theta1)) ** 2 + q ** 2) * R_s
    return R_p, R_s, theta2


def scattering(sigma, path, R):
    """tracer/optics.py:214-239 with the draw given: free path -ln(R) / sigma (sigma == 0: the way to the surface itself, i.e.
    no scattering); scattered where it is shorter than the way to the surface.  Returns (scattered, free paths)."""
    with N.errstate(divide='ignore', invalid='ignore'):
        lengths = -N.log(R) / sigma
    lengths = N.where(sigma == 0., path, lengths)
    return lengths < path, lengths


def hg_theta(g, R):
    """ray_trace_utils/sampling.py:160-168: polar angle of a Henyey-Greenstein event from its uniform (g may be an array)"""
    g = N.asarray(g, dtype=float) * N.ones_like(R)
    s = 2. * R - 1.
    gs = N.where(g == 0., 1., g)
    c = 1. / (2. * gs) * (1. + gs ** 2 - ((1. - gs ** 2) / (1. + gs * s)) ** 2)
    return N.arccos(N.clip(N.where(g == 0., s, c), -1., 1.))


def iam(opt, ia, ic, d, nrm):
    """IAM.__call__ (optics_callables.py:276-281) as a factor on e (1 - abs); a_r = opt[ia] (0 or absent: 1), c = opt[ic]"""
    if len(opt) <= ic or opt[ia] == 0.:
        return 1.
    vertical = N.sum(d * nrm, axis=0) * nrm
    cos_aoi = N.sqrt(N.sum(vertical ** 2, axis=0))
    return (1. - N.exp(-cos_aoi ** opt[ic] / opt[ia])) / (1. - N.exp(-1. / opt[ia]))


def attenuations(path_lengths, k, lambda_0, energy):
    """optics.py:205-212"""
    return N.exp(-4. * N.pi * path_lengths * k / lambda_0) * energy


def interp2(tab, th, lam):
    """RegularGridInterpolator (linear) over (theta, lambda), arguments clamped to the grid; tab as csrc/trc_core.h trc_interp2"""
    nt, nl = int(tab[0]), int(tab[1])
    ts, ls = tab[2:2 + nt], tab[2 + nt:2 + nt + nl]
    v = N.asarray(tab[2 + nt + nl:]).reshape(nt, nl)

    def cell(xs, x):
        x = N.clip(x, xs[0], xs[-1])
        i = N.clip(N.searchsorted(xs, x, side='right') - 1, 0, len(xs) - 2)
        return i, (x - xs[i]) / (xs[i + 1] - xs[i])
    it, wt = cell(ts, th)
    il, wl = cell(ls, lam)
    return (1. - wt) * ((1. - wl) * v[it, il] + wl * v[it, il + 1]) + wt * ((1. - wl) * v[it + 1, il] + wl * v[it + 1, il + 1])


def spectral_factor(opt_kind, opt, extra, d, nrm):
    """What the optics classes do to the spectrum a polychromatic ray carries (`outg._spectra *= ...`): 1 - absorptivity for
    Reflective :137-138, Lambertian :173-174 (LambertianAbsorbant builds its Lambertian with 0, :897) and RealReflective :266-267
    with their one-sided and IAM children; 1 - absorptance(theta) for Lambertian_directional_axisymmetric_piecewise :358-359;
    every other class hands the spectrum on unchanged (RayBundle.inherit)."""
    H = d.shape[1]
    if opt_kind in (OPT_REFLECTIVE, OPT_ONE_SIDED_REFLECTIVE, OPT_REAL_REFLECTIVE, OPT_ONE_SIDED_REAL_REFLECTIVE):
        return N.full(H, 1. - opt[0])
    if opt_kind == OPT_LAMBERTIAN:
        return N.full(H, 1. - opt[0]) if not (len(opt) > 2 and opt[2] != 0.) else N.ones(H)
    if opt_kind == OPT_LAMBERTIAN_DIRECTIONAL and int(opt[0]) == 0:
        vert = N.sum(d * nrm, axis=0) * nrm
        th = N.arccos(N.sqrt(N.sum(vert ** 2, axis=0)))
        k = len(extra) // 2
        return 1. - N.interp(th, extra[:k], extra[k:2 * k])
    return N.ones(H)


def shade(opt_kind, opt, extra, up, d, e, ref, wl, nrm, seed, rid, event, path=None, ext=None):
    """
    shade_core plus what the rays of the ordered engine carry (csrc/trc_core.h trc_shade_x).  ext: dict with
    mat (K, H) complex: the scene's materials at each ray's wavelength; spec, swl (W, H): spectrum and its wavelength grid.
    Blocks then carry `spectra` (W, k) too; `ref` is complex when the rays' is.
    """
    H = d.shape[1]
    allsel = N.arange(H)
    ext = ext or {}
    spec, swl = ext.get('spec'), ext.get('swl')
    if opt_kind == OPT_REFRACTIVE_MATERIAL:                      # Refractive :726-858, RefractiveAbsorbant :908-944
        single, sigma, atten, scaling, k0, k1 = opt[0] != 0., opt[1], opt[2] != 0., opt[3], int(opt[4]), int(opt[5])
        mat0, mat1 = ext['mat'][k0], ext['mat'][k1]
        u0, u1 = philox.uniform_pair(seed, rid, event, 0)
        u2, u3 = philox.uniform_pair(seed, rid, event, 1)
        nrm = nrm.copy()
        if sigma >= 0.:                                          # :767-781
            g0, _ = philox.normal_pair(u0, u1)
            th = sigma * g0
            phi = 2. * N.pi * u2
            err = N.vstack((N.sin(th) * N.cos(phi), N.sin(th) * N.sin(phi), N.cos(th)))
            rots = rotation_to_z(nrm.T)
            for i in range(H):
                nrm[:, i] = N.dot(rots[i], err[:, i])
        m1 = N.asarray(ref, dtype=complex)
        m2 = N.where(m1 == mat0, mat1, mat0)                     # :750-751
        with N.errstate(all='ignore'):
            refr, out_dirs = refractions(m1.real, m2.real, d, nrm)   # :786
            R = N.ones(H)
            R[refr] = N.real(fresnel(d[:, refr], nrm[:, refr], m1[refr], m2[refr]))     # :838-840 (the cast keeps the real part)
        refl_dirs = reflections(d, nrm)
        if single:                                               # :796-823
            refl = u3 <= R
            dirs_refr = N.zeros((3, H))
            dirs_refr[:, refr] = out_dirs
            blocks = []
            if refl.any():
                blocks.append(dict(sel=allsel[refl], directions=refl_dirs[:, refl], energy=e[refl], ref=m1[refl], rid=rid[refl]))
            if (~refl).any():
                blocks.append(dict(sel=allsel[~refl], directions=dirs_refr[:, ~refl], energy=e[~refl], ref=m2[~refl], rid=rid[~refl]))
        else:                                                    # :825-835
            blocks = [dict(sel=allsel, directions=refl_dirs, energy=e * R, ref=m1.copy(), rid=rid)]
            if refr.any():
                blocks.append(dict(sel=allsel[refr], directions=out_dirs, energy=e[refr] * (1. - R[refr]), ref=m2[refr],
                                   rid=philox.child_rid(rid[refr], event)))
        for b in blocks:
            if atten:                                            # Absorbant.attenuate :874-882: k and lambda of the new bundle
                b['energy'] = attenuations(path[b['sel']] * scaling, b['ref'].imag, wl[b['sel']], b['energy'])
            if spec is not None:
                b['spectra'] = spec[:, b['sel']].copy()
        return blocks
    if opt_kind == OPT_LAMBERTIAN_POLYCHROMATIC:                 # :406-425
        vert = N.sum(d * nrm, axis=0) * nrm
        th = N.arccos(N.sqrt(N.sum(vert ** 2, axis=0)))
        ab = interp2(extra, N.tile(th, (spec.shape[0], 1)), swl)
        spectra = spec * (1. - ab)
        energy = N.sum((swl[1:] - swl[:-1]) * (spectra[1:] + spectra[:-1]) / 2., axis=0)      # N.trapz(spectra, wavelengths, axis=0)
        u0, u1 = philox.uniform_pair(seed, rid, event, 0)
        dirs = lambertian_directions(nrm, 2. * N.pi * u0, u1, N.pi / 2.)
        return [dict(sel=allsel, directions=dirs, energy=energy, ref=N.asarray(ref).copy(), rid=rid, spectra=spectra)]
    cplx = N.iscomplexobj(ref)
    blocks = shade_core(opt_kind, opt, extra, up, d, e, N.real(ref) if cplx else ref, wl, nrm, seed, rid, event, path=path)
    if cplx:        # the other optics hand the index on unchanged, or set a real one (RefractiveHomogenous toggles real indices)
        for b in blocks:
            same = b['ref'] == N.real(ref)[b['sel']]
            b['ref'] = N.where(same, N.asarray(ref)[b['sel']], b['ref'].astype(complex))
    if spec is not None:
        f = spectral_factor(opt_kind, opt, extra, d, nrm)
        for b in blocks:
            b['spectra'] = spec[:, b['sel']] * f[b['sel']]
        if opt_kind == OPT_PERIODIC_BOUNDARY:                    # the stub's spectrum is cancelled with its energy (:710-713)
            blocks[0]['spectra'] = N.zeros_like(blocks[0]['spectra'])
    return blocks


def shade_core(opt_kind, opt, extra, up, d, e, ref, wl, nrm, seed, rid, event, path=None):
    """
    One optics call on H hits.  Returns a list of blocks (reflected block first, refracted second), each a dict
    with sel (indices into the H hits), directions (3,k), energy (k,), ref (k,), rid (k,).
    Draw order = csrc/trc_core.h trc_shade.  path: distance travelled to the hit (Absorbant.attenuate, :874-889), or None.
    """
    H = d.shape[1]
    allsel = N.arange(H)
    if opt_kind == OPT_TRANSPARENT:                              # optics_callables.py:106-113
        return [dict(sel=allsel, directions=d.copy(), energy=e.copy(), ref=ref.copy(), rid=rid)]
    if opt_kind == OPT_PERIODIC_BOUNDARY:                        # :703-723: the stub of energy 0, then the ray one period along the normal
        return [dict(sel=allsel, directions=d.copy(), energy=N.zeros(H), ref=ref.copy(), rid=philox.child_rid(rid, event)),
                dict(sel=allsel, directions=d.copy(), energy=e.copy(), ref=ref.copy(), rid=rid, shift=N.full(H, float(opt[0])))]
    if opt_kind in (OPT_REFLECTIVE, OPT_ONE_SIDED_REFLECTIVE):   # :130-140, :201-212
        eo = e * (1. - opt[0]) * iam(opt, 1, 2, d, nrm)
        if opt_kind == OPT_ONE_SIDED_REFLECTIVE:
            eo = eo.copy()
            eo[N.sum(d * up[:, None], axis=0) > 0] = 0
        return [dict(sel=allsel, directions=reflections(d, nrm), energy=eo, ref=ref.copy(), rid=rid)]
    if opt_kind == OPT_REFLECTIVE_SPECTRAL:                      # :183-193
        k = len(extra) // 2
        eo = e * (1. - N.interp(wl, extra[:k], extra[k:]))
        return [dict(sel=allsel, directions=reflections(d, nrm), energy=eo, ref=ref.copy(), rid=rid)]
    if opt_kind in (OPT_REAL_REFLECTIVE, OPT_ONE_SIDED_REAL_REFLECTIVE):   # :231-269, :498-504
        sigma, bi = opt[1], opt[2] != 0.
        real = nrm
        if sigma > 0.:
            u0, u1 = philox.uniform_pair(seed, rid, event, 0)
            g0, g1 = philox.normal_pair(u0, u1)
            u2 = philox.uniform_pair(seed, rid, event, 1)[0] if not bi else N.zeros(H)
            real = slope_error_normals(nrm, sigma, bi, g0, g1, u2)
        eo = e * (1 - opt[0]) * iam(opt, 3, 4, d, nrm)
        if opt_kind == OPT_ONE_SIDED_REAL_REFLECTIVE:
            eo = eo.copy()
            eo[N.sum(d * up[:, None], axis=0) > 0] = 0
        return [dict(sel=allsel, directions=reflections(d, real), energy=eo, ref=ref.copy(), rid=rid)]
    if opt_kind == OPT_LAMBERTIAN:                               # :154-176
        u0, u1 = philox.uniform_pair(seed, rid, event, 0)
        dirs = lambertian_directions(nrm, 2. * N.pi * u0, u1, opt[1])
        eo = e * (1. - opt[0]) * iam(opt, 4, 5, d, nrm)
        if len(opt) > 2 and opt[2] != 0.:                        # LambertianAbsorbant :898-906: attenuate, then absorb
            eo = e * N.exp(-opt[2] * (path * opt[3])) * (1. - opt[0])
        return [dict(sel=allsel, directions=dirs, energy=eo, ref=ref.copy(), rid=rid)]
    if opt_kind == OPT_SEMI_LAMBERTIAN:
        # optics_callables.py:506-531 as the class describes itself (:507-509): mirror above `angular_range` of incidence,
        # Lambertian below, specular block first (:531).  The reference's __call__ cannot run (:525 indexes the (3, n)
        # direction array by ray number): parity unpinned, this is the evident intent.
        angs = N.arccos(-N.sum(d * nrm, axis=0))
        gl = angs > opt[1]
        u0, u1 = philox.uniform_pair(seed, rid, event, 0)
        ng = ~gl
        return [dict(sel=allsel[gl], directions=reflections(d[:, gl], nrm[:, gl]), energy=e[gl] * (1. - opt[0]), ref=ref[gl].copy(), rid=rid[gl]),
                dict(sel=allsel[ng], directions=lambertian_directions(nrm[:, ng], 2. * N.pi * u0[ng], u1[ng], opt[1]),
                     energy=e[ng] * (1. - opt[0]), ref=ref[ng].copy(), rid=rid[ng])]
    if opt_kind == OPT_LAMBERTIAN_SPECULAR:                      # :561-585
        u0, u1 = philox.uniform_pair(seed, rid, event, 0)
        u2, _ = philox.uniform_pair(seed, rid, event, 1)
        specular = u0 < opt[1]
        dirs = N.zeros(d.shape)
        dirs[:, specular] = reflections(d[:, specular], nrm[:, specular])
        ns = ~specular
        dirs[:, ns] = lambertian_directions(nrm[:, ns], 2. * N.pi * u1[ns], u2[ns], N.pi / 2.)
        return [dict(sel=allsel, directions=dirs, energy=e * (1. - opt[0]), ref=ref.copy(), rid=rid)]
    if opt_kind in (OPT_LAMBERTIAN_DIRECTIONAL, OPT_LAMBERTIAN_DIRECTIONAL_SPECTRAL):   # :340-361, :373-391
        vertical = N.sum(d * nrm, axis=0) * nrm
        thetas_in = N.arccos(N.sqrt(N.sum(vertical ** 2, axis=0)))
        mode = int(opt[0]) if (opt_kind == OPT_LAMBERTIAN_DIRECTIONAL and len(opt)) else 0
        if opt_kind == OPT_LAMBERTIAN_DIRECTIONAL:
            k = len(extra) // (3 if mode == 2 else 2)
            ang_abss = N.interp(thetas_in, extra[:k], extra[k:2 * k])
        else:
            from scipy.interpolate import RegularGridInterpolator
            nt, nl = int(extra[0]), int(extra[1])
            ts, ls = extra[2:2 + nt], extra[2 + nt:2 + nt + nl]
            grid = N.reshape(extra[2 + nt + nl:], (nt, nl))
            pts = N.array([N.clip(thetas_in, ts[0], ts[-1]), N.clip(wl, ls[0], ls[-1])]).T
            ang_abss = RegularGridInterpolator((ts, ls), grid)(pts)
        u0, u1 = philox.uniform_pair(seed, rid, event, 0)
        if mode == 0:
            dirs = lambertian_directions(nrm, 2. * N.pi * u0, u1, N.pi / 2.)
        else:       # :427-455 constant specularity, :457-487 specularity interpolated on the incidence angle
            spec = opt[1] if mode == 1 else N.interp(thetas_in, extra[:k], extra[2 * k:])
            u2, _ = philox.uniform_pair(seed, rid, event, 1)
            sp = u0 < spec
            dirs = N.zeros(d.shape)
            dirs[:, sp] = reflections(d[:, sp], nrm[:, sp])
            dirs[:, ~sp] = lambertian_directions(nrm[:, ~sp], 2. * N.pi * u1[~sp], u2[~sp], N.pi / 2.)
        return [dict(sel=allsel, directions=dirs, energy=e * (1. - ang_abss), ref=ref.copy(), rid=rid)]
    if opt_kind == OPT_FRESNEL_CONDUCTOR:                        # :1536-1558 with optics.py:41-81
        k = len(extra) // 3
        m2 = N.interp(wl, extra[:k], extra[k:2 * k]) + 1j * N.interp(wl, extra[:k], extra[2 * k:])
        theta1 = N.arccos(N.abs((nrm * d).sum(axis=0)))
        R_p, R_s, _ = fresnel_to_attenuating(opt[0], m2, theta1)
        return [dict(sel=allsel, directions=reflections(d, nrm), energy=e * (R_p + R_s) / 2., ref=ref.copy(), rid=rid)]
    if opt_kind == OPT_REFRACTIVE_SCATTERING:
        # optics_callables.py:946-1036 / :1350-1376 as their docstrings and the body kept in comments at :1385-1470 describe them
        # (the classes do not run in the reference): a free path -ln(R) / s_c (optics.py:214-239) shorter than the way to the
        # surface scatters the ray there into a Henyey-Greenstein direction about its own (sampling.py:150-168), the others are
        # refracted as by RefractiveHomogenous.  Blocks: scattered, reflected, refracted.  Draws: block 2 = (R, R_hg), block 3 = (azimuth, -).
        s_c = N.where(ref == opt[0], extra[0], extra[1])
        g = N.where(ref == opt[0], extra[2], extra[3])
        r0, r1 = philox.uniform_pair(seed, rid, event, 2)
        r2, _ = philox.uniform_pair(seed, rid, event, 3)
        scat, lengths = scattering(s_c, path, r0)
        blocks = []
        if scat.any():
            th = hg_theta(g[scat], r1[scat])
            ph = 2. * N.pi * r2[scat]
            loc = N.vstack((N.sin(th) * N.cos(ph), N.sin(th) * N.sin(ph), N.cos(th)))
            dirs = rotate_z_to_normal(loc, d[:, scat])
            blocks.append(dict(sel=allsel[scat], directions=dirs, energy=e[scat].copy(), ref=ref[scat].copy(), rid=rid[scat],
                               back=(path - lengths)[scat]))
        keep = ~scat
        if keep.any():
            sub = shade(OPT_REFRACTIVE_HOMOGENOUS, opt, extra, up, d[:, keep], e[keep], ref[keep], wl[keep], nrm[:, keep], seed, rid[keep],
                        event, path=path[keep])
            idx = allsel[keep]
            for b in sub:
                b['sel'] = idx[b['sel']]
                blocks.append(b)
        return blocks
    if opt_kind == OPT_REFRACTIVE_HOMOGENOUS:                    # :1226-1296 on :836-858
        na, nb, single, sigma = opt[0], opt[1], opt[2] != 0., opt[3]
        u0, u1 = philox.uniform_pair(seed, rid, event, 0)
        u2, u3 = philox.uniform_pair(seed, rid, event, 1)
        nrm = nrm.copy()
        if sigma >= 0.:                                          # :1227-1239
            g0, _ = philox.normal_pair(u0, u1)
            th = sigma * g0
            phi = 2. * N.pi * u2
            err = N.vstack((N.sin(th) * N.cos(phi), N.sin(th) * N.sin(phi), N.cos(th)))
            rots = rotation_to_z(nrm.T)
            for i in range(H):
                nrm[:, i] = N.dot(rots[i], err[:, i])
        if len(opt) > 7 and opt[7] != 0.:                        # RefractiveTransmissiveHomogenous :1345-1348, Absorbant :881-886
            e = e * N.exp(-N.where(ref == nb, opt[5], opt[4]) * (path * opt[6]))
        n1 = ref
        n2 = N.where(n1 == na, nb, na)                           # :1217-1218
        refr, out_dirs = refractions(n1, n2, d, nrm)
        R = N.ones(H)
        R[refr] = fresnel(d[:, refr], nrm[:, refr], n1[refr], n2[refr])
        refl_dirs = reflections(d, nrm)
        if single:                                               # :1254-1280
            refl = u3 <= R
            dirs_refr = N.zeros((3, H))
            dirs_refr[:, refr] = out_dirs
            blocks = []
            if refl.any():
                blocks.append(dict(sel=allsel[refl], directions=refl_dirs[:, refl], energy=e[refl], ref=ref[refl],
                                   rid=rid[refl]))
            if (~refl).any():
                blocks.append(dict(sel=allsel[~refl], directions=dirs_refr[:, ~refl], energy=e[~refl], ref=n2[~refl],
                                   rid=rid[~refl]))
            return blocks
        blocks = [dict(sel=allsel, directions=refl_dirs, energy=e * R, ref=ref.copy(), rid=rid)]      # :1284-1294
        if refr.any():
            blocks.append(dict(sel=allsel[refr], directions=out_dirs, energy=e[refr] * (1. - R[refr]), ref=n2[refr],
                               rid=philox.child_rid(rid[refr], event)))
        return blocks
    raise ValueError(opt_kind)
