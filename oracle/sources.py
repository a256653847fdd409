"""Source bundles restated (tracer/sources.py), with the uniforms given or drawn from Philox in device order."""
import numpy as N
from .kinds import *
from . import philox
from .optics import pillbox_directions

NELEM = 210


def buie_tables(CSR, pre_process_CSR=True):
    """sources.py:333-361.  Returns dict(theta, phi, integ, cdf, gamma, kappa, theta_dni, theta_tot, CSR)"""
    theta_dni = 4.65e-3
    theta_tot = 43.6e-3
    theta_int = N.linspace(0., theta_dni, NELEM + 1)
    phi = N.cos(0.326 * theta_int * 1e3) / N.cos(0.308 * theta_int * 1e3)
    integ = 0.5 * (phi[:-1] * N.cos(theta_int[:-1]) * N.sin(theta_int[:-1]) + phi[1:] * N.cos(theta_int[1:]) * N.sin(theta_int[1:])) * (theta_int[1:] - theta_int[:-1])
    gamma = kappa = 0.
    if CSR == 0.:
        integ_phi = N.sum(integ)
    else:
        if pre_process_CSR:
            if CSR <= 0.1:
                CSR = -2.245e+03 * CSR ** 4. + 5.207e+02 * CSR ** 3. - 3.939e+01 * CSR ** 2. + 1.891e+00 * CSR + 8e-03
            else:
                CSR = 1.973 * CSR ** 4. - 2.481 * CSR ** 3. + 0.607 * CSR ** 2. + 1.151 * CSR - 0.020
        kappa = 0.9 * N.log(13.5 * CSR) * CSR ** (-0.3)
        gamma = 2.2 * N.log(0.52 * CSR) * CSR ** (0.43) - 0.1
        integ_csr = 1e-6 * N.exp(kappa) / (gamma + 2.) * ((theta_tot * 1000.) ** (gamma + 2.) - (theta_dni * 1000.) ** (gamma + 2.))
        integ_phi = N.sum(integ) + integ_csr
    cdf = N.add.accumulate(N.hstack(([0], integ / integ_phi)))
    return dict(theta=theta_int, phi=phi, integ=integ, cdf=cdf, gamma=gamma, kappa=kappa, theta_dni=theta_dni,
                theta_tot=theta_tot, CSR=CSR)


def buie_thetas(R_thetas, tab):
    """sources.py:364-377 -- the loop over the 210 intervals, as written"""
    theta_int, phi, integ, cdf = tab['theta'], tab['phi'], tab['integ'], tab['cdf']
    thetas = N.zeros(len(R_thetas))
    for i in range(len(cdf) - 1):
        sl = N.logical_and(R_thetas >= cdf[i], R_thetas < cdf[i + 1])
        A = phi[i] * N.cos(theta_int[i]) * N.sin(theta_int[i])
        B = phi[i + 1] * N.cos(theta_int[i + 1]) * N.sin(theta_int[i + 1])
        C = 2. * N.sum(integ) * (R_thetas[sl] - cdf[i]) * (theta_int[i + 1] - theta_int[i])
        thetas[sl] = -(-A * theta_int[i + 1] + B * theta_int[i] + N.sqrt(((theta_int[i] - theta_int[i + 1]) * A) ** 2. + C * (B - A))) / (A - B)
    aureole = R_thetas >= cdf[-1]
    if tab['CSR'] > 0.:
        g, k = tab['gamma'], tab['kappa']
        Ra = R_thetas[aureole]
        thetas[aureole] = ((Ra - 1.) * ((g + 2.) / (10. ** (3. * g) * N.exp(k)) * N.sum(integ) - tab['theta_dni'] ** (g + 2.)) + Ra * tab['theta_tot'] ** (g + 2.)) ** (1. / (g + 2.))
    return thetas


def table_from_desc_buie(buie):
    """Rebuild the table dict from trc_source_desc.buie (theta | g | cdf | scalars)."""
    b = N.asarray(buie, dtype=float)
    n = NELEM + 1
    return dict(theta=b[:n], g=b[n:2 * n], cdf=b[2 * n:3 * n], I_dni=b[3 * n], gamma=b[3 * n + 1], kappa=b[3 * n + 2],
                theta_dni=b[3 * n + 3], theta_tot=b[3 * n + 4], csr_pos=b[3 * n + 5] != 0.)


def buie_thetas_packed(R, t):
    """Same map on the packed table that crosses the C-ABI (g = phi*cos*sin precomputed by the host)."""
    theta, g, cdf = t['theta'], t['g'], t['cdf']
    thetas = N.zeros(len(R))
    for i in range(NELEM):
        sl = N.logical_and(R >= cdf[i], R < cdf[i + 1])
        A, B = g[i], g[i + 1]
        C = 2. * t['I_dni'] * (R[sl] - cdf[i]) * (theta[i + 1] - theta[i])
        thetas[sl] = -(-A * theta[i + 1] + B * theta[i] + N.sqrt(((theta[i] - theta[i + 1]) * A) ** 2. + C * (B - A))) / (A - B)
    aur = R >= cdf[-1]
    if t['csr_pos']:
        gm, k = t['gamma'], t['kappa']
        Ra = R[aur]
        thetas[aur] = ((Ra - 1.) * ((gm + 2.) / (10. ** (3. * gm) * N.exp(k)) * t['I_dni'] - t['theta_dni'] ** (gm + 2.)) + Ra * t['theta_tot'] ** (gm + 2.)) ** (1. / (gm + 2.))
    return thetas


def generate(src, n, seed, offset, uniforms=None):
    """
    src: dict(kind, center(3), rot_pos(3,3), rot_dir(3,3), p, energy, buie) mirroring trc_source_desc.
    uniforms: optional (u0,u1,u2,u3) in [0,1) replacing the Philox draws (variate replay against the reference).
    Returns vertices (3,n), directions (3,n), energy (n,), rid (n,)
    """
    rid = N.arange(n, dtype=N.uint64) + N.uint64(offset)
    if uniforms is None:
        u0, u1, u2, u3 = philox.uniform_quad(seed, rid, 0, 0)
    else:
        u0, u1, u2, u3 = uniforms
    p = src['p']
    kind = src['kind']
    if kind == SRC_PILLBOX_DISK:        # sources.py:200-216
        a = pillbox_directions(2. * N.pi * u0, u1, p[4])
        rs = N.sqrt(p[1] ** 2. + u2 * (p[0] ** 2. - p[1] ** 2.))
        th = p[2] + (p[3] - p[2]) * u3
        xs, ys = rs * N.cos(th), rs * N.sin(th)
        if p[5] != 0. and uniforms is None:     # x_cut (sources.py:216-228): rejected positions are redrawn; per-ray stream here
            blk = 2
            redo = N.nonzero(~(xs < p[6]))[0]
            while len(redo) and blk < 2 + 4096:
                a2, a3 = philox.uniform_pair(seed, rid[redo], 0, blk)
                r2 = N.sqrt(p[1] ** 2. + a2 * (p[0] ** 2. - p[1] ** 2.))
                t2 = p[2] + (p[3] - p[2]) * a3
                xs[redo], ys[redo] = r2 * N.cos(t2), r2 * N.sin(t2)
                redo = redo[~(xs[redo] < p[6])]
                blk += 1
        loc = N.vstack((xs, ys, N.zeros(n)))
    elif kind == SRC_PILLBOX_RECT:      # sources.py:243-256
        a = pillbox_directions(2. * N.pi * u0, u1, p[2])
        xs = -p[0] / 2. + p[0] * u2
        ys = -p[1] / 2. + p[1] * u3
        if p[3] != 0.:
            xs, ys = ys, xs
        loc = N.vstack((ys, xs, N.zeros(n)))
    elif kind == SRC_PILLBOX_TRIANGLE:  # sources.py:559-568
        sq = N.sqrt(u0)
        loc = N.vstack((sq * (1. - u1), u1 * sq, N.zeros(n)))
        a = pillbox_directions(2. * N.pi * u2, u3, p[0])
    elif kind in (SRC_VF_CYLINDER, SRC_VF_FRUSTUM):
        if kind == SRC_VF_CYLINDER:     # sources.py:737-746
            zs = p[1] * u0 - p[1] / 2.
            phi = p[2] + (p[3] - p[2]) * u1
            loc = N.vstack((p[0] * N.cos(phi), p[0] * N.sin(phi), zs))
            flat = pillbox_directions(2. * N.pi * u2, u3, p[4])
            slope, sign = 0., p[5]
        else:                           # sources.py:670-685
            flat = pillbox_directions(2. * N.pi * u0, u1, p[5])
            slope = (p[1] - p[0]) / p[2]
            rs = N.sqrt((p[1] ** 2. - p[0] ** 2.) * u2 + p[0] ** 2.)
            zs = (rs - p[0]) / ((p[1] - p[0]) / p[2])
            phi = p[3] + (p[4] - p[3]) * u3
            loc = N.vstack((rs * N.cos(phi), rs * N.sin(phi), zs))
            sign = p[6]
        trot = -N.pi / 2. + N.arctan(slope)     # sources.py:687-695, :748-753: rotz(phi) . roty(trot) . dir_flat
        cy, sy = N.cos(trot), N.sin(trot)
        rx, ry, rz = cy * flat[0] + sy * flat[2], flat[1], -sy * flat[0] + cy * flat[2]
        a = sign * N.vstack((N.cos(phi) * rx - N.sin(phi) * ry, N.sin(phi) * rx + N.cos(phi) * ry, rz))
    else:
        tab = table_from_desc_buie(src['buie'])
        if kind == SRC_BUIE_DISK:       # sources.py:431-434
            rs = p[0] * N.sqrt(u0)
            ph = 2. * N.pi * u1
            loc = N.vstack((rs * N.cos(ph), rs * N.sin(ph), N.zeros(n)))
        else:                           # sources.py:485-486
            loc = N.vstack((p[0] * (u0 - 0.5), p[1] * (u1 - 0.5), N.zeros(n)))
        th = buie_thetas_packed(u2, tab)
        xi = 2. * N.pi * u3
        st = N.sin(th)
        a = N.vstack((N.cos(xi) * st, N.sin(xi) * st, N.cos(th)))     # sources.py:380-382
    verts = N.dot(src['rot_pos'], loc) + N.asarray(src['center']).reshape(3, 1)
    dirs = N.dot(src['rot_dir'], a)
    return verts, dirs, N.ones(n) * src['energy'], rid
