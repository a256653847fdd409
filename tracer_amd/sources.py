"""
Source bundles.  Function names and arguments follow the reference's tracer/sources.py
(:88-117 direction samplers, :175-239 disk_bundle, :241-264 rect_bundle, :330-384 Buie sampling,
:412-464 buie_sunshape, :466-515 rect_buie_sunshape).

Everything that does not depend on the random draws -- the rotation_to_z frames, the per-ray
energy, the 210-interval Buie CDF table -- is evaluated here on the host once per call and packed
into a trc_source_desc; the rays themselves are sampled on the GPU (csrc/trc_core.h,
trc_source_ray), either materialised as a bundle (trc_source_generate) or fused into the trace
kernel.  The functions return a LazySourceBundle: a RayBundle whose columns appear on first access.
"""
import ctypes as C

import numpy as N

from . import _cabi, rng
from .ray_bundle import RayBundle, concatenate_rays
from .spatial_geometry import rotation_to_z


class LazySourceBundle(RayBundle):
    """
    A source bundle described by (descriptor, n, seed, ray_offset).  Ray i has random stream id
    ray_offset + i.  Any column access generates the rays on the device; an engine that receives an
    untouched LazySourceBundle generates them inside its own kernel instead.
    """
    def __init__(self, desc, n, seed, ray_offset=0, constant_columns=None):
        RayBundle.__init__(self, **dict((k, None) for k in (constant_columns or {})))
        object.__setattr__(self, '_src_const', dict(constant_columns or {}))
        object.__setattr__(self, '_src_desc', desc)
        object.__setattr__(self, '_src_n', int(n))
        object.__setattr__(self, '_src_seed', int(seed))
        object.__setattr__(self, '_src_offset', int(ray_offset))
        object.__setattr__(self, '_src_done', False)

    def is_pending(self):
        # bundles carrying extra per-ray columns (wavelength, ref_index) are materialised before tracing
        return not self._src_done and not self._src_const

    def get_num_rays(self):
        if not self._src_done:
            return self._src_n
        return RayBundle.get_num_rays(self)

    def source_args(self):
        return self._src_desc, self._src_n, self._src_seed, self._src_offset

    def _materialize(self):
        if self._src_done:
            return
        object.__setattr__(self, '_src_done', True)
        n = self._src_n
        ctx = _cabi.get_context()
        # (page-locked for large bundles: 1e7 rays are 560 MB, 11 ms over the link instead of 40 through pageable memory)
        v = _cabi.pinned_empty((3, n))
        d = _cabi.pinned_empty((3, n))
        e = _cabi.pinned_empty(n)
        rays = _cabi.make_rays(n, v[0], v[1], v[2], d[0], d[1], d[2], e)
        _cabi.check(ctx.lib.trc_source_generate(ctx.handle, C.byref(self._src_desc), n, self._src_seed,
                                                self._src_offset, C.byref(rays)))
        self._cols['vertices'] = v
        self._cols['directions'] = d
        self._cols['energy'] = e
        for k, val in self._src_const.items():
            self._cols[k] = N.ones(n) * val


def _fill_source(kind, center, rot_pos, rot_dir, params, energy, buie=None):
    s = _cabi.SourceDesc()
    s.kind = kind

    def put(field, values):
        # block copy into the ctypes array (element-wise assignment of the 639-entry Buie table was most of the
        # cost of making a bundle)
        a = N.ascontiguousarray(N.ravel(N.asarray(values, dtype=float)))
        C.memmove(field, a.ctypes.data, a.nbytes)

    put(s.center, center)
    put(s.rot_pos, rot_pos)
    put(s.rot_dir, rot_dir)
    if len(params):
        put(s.p, [float(p) for p in params])
    s.energy = float(energy)
    if buie is not None:
        put(s.buie, buie)
    return s


def _new_bundle(desc, num_rays, seed, ray_offset, constant_columns=None):
    if seed is None:
        seed = rng.next_seed()
    return LazySourceBundle(desc, int(num_rays), seed, ray_offset, constant_columns)


def _tilt_cos(rays_direction, direction):
    """cos of the angle between the bundle axis and the emitting plane normal (sources.py:235, :445)."""
    chord = N.sqrt(N.sum((N.asarray(rays_direction, dtype=float) - N.asarray(direction, dtype=float)) ** 2))
    return N.cos(2. * N.sin(chord / 2.))


def disk_bundle(num_rays, center, direction, radius, ang_range, flux=None, radius_in=0., angular_span=[0., 2. * N.pi],
                x_cut=None, procs=1, rays_direction=None, seed=None, ray_offset=0):
    """
    Pillbox/Lambertian-cone rays leaving an annular disc.  center: 3x1 column, direction: 3-vector
    normal of the disc; energies flux*area/N*cos(tilt), or 1/N/procs without flux.  x_cut keeps the part of the disc
    with local x < x_cut (positions are redrawn until they qualify; the energy still refers to the whole disc, as in
    the reference).
    """
    radius, radius_in = float(radius), float(radius_in)
    if x_cut is not None and not x_cut > -radius:
        raise ValueError("disk_bundle: x_cut leaves no part of the disc")
    if rays_direction is None:
        rays_direction = direction
    if flux is not None:
        energy = N.pi * (radius ** 2. - radius_in ** 2.) / num_rays * flux * _tilt_cos(rays_direction, direction)
    else:
        energy = 1. / float(num_rays) / procs
    rot = rotation_to_z(rays_direction)
    desc = _fill_source(_cabi.SRC_PILLBOX_DISK, center, rot, rot,
                        [radius, radius_in, angular_span[0], angular_span[1], ang_range,
                         0. if x_cut is None else 1., 0. if x_cut is None else float(x_cut)], energy)
    return _new_bundle(desc, num_rays, seed, ray_offset)


def rect_bundle(num_rays, center, direction, x, y, ang_range, flux=None, procs=1, seed=None, ray_offset=0):
    """Pillbox rays leaving an x by y rectangle normal to `direction` (sources.py:241-264)."""
    direction = N.asarray(direction)
    swap = bool((direction == N.array([0, 0, -1])).all())
    energy = x * y / num_rays * flux if flux is not None else 1. / float(num_rays) / procs
    rot = rotation_to_z(direction)
    desc = _fill_source(_cabi.SRC_PILLBOX_RECT, center, rot, rot, [x, y, ang_range, 1. if swap else 0.], energy)
    return _new_bundle(desc, num_rays, seed, ray_offset)


def oblique_solar_rect_bundle(num_rays, center, source_direction, rays_direction, x, y, ang_range, flux=None, procs=1,
                              wavelength=None, ref_index=None, seed=None, ray_offset=0):
    """Pillbox rays about `rays_direction` leaving an x by y rectangle normal to `source_direction`
    (sources.py:268-302); optional constant wavelength / ref_index columns."""
    source_direction = N.asarray(source_direction, dtype=float)
    rays_direction = N.asarray(rays_direction, dtype=float)
    swap = bool((source_direction == N.array([0, 0, -1])).all())
    if flux is not None:
        cosangle = 2. * N.arcsin(0.5 * N.sqrt(N.sum((rays_direction - source_direction) ** 2)))
        energy = x * y / num_rays * flux * N.cos(cosangle)
    else:
        energy = 1. / float(num_rays) / procs
    desc = _fill_source(_cabi.SRC_PILLBOX_RECT, center, rotation_to_z(source_direction), rotation_to_z(rays_direction),
                        [x, y, ang_range, 1. if swap else 0.], energy)
    const = {}
    if wavelength is not None:
        const['wavelengths'] = wavelength
    if ref_index is not None:
        const['ref_index'] = ref_index
    return _new_bundle(desc, num_rays, seed, ray_offset, const)


def triangular_bundle(num_rays, A, B, C, direction=None, ang_range=N.pi / 2., flux=None, procs=1, seed=None, ray_offset=0):
    """Pillbox rays leaving the triangle ABC (uniform point picking), about `direction` (default: the triangle
    normal AB x AC) -- sources.py:544-597."""
    A, B, C = [N.ravel(N.asarray(q, dtype=float)) for q in (A, B, C)]
    AB, AC = B - A, C - A
    normal = N.cross(AB, AC)
    normal = normal / N.sqrt(N.sum(normal ** 2))
    if direction is None:
        direction = normal
    direction = N.ravel(N.asarray(direction, dtype=float))
    l1, l2, l3 = N.sqrt(N.sum(AB ** 2)), N.sqrt(N.sum(AC ** 2)), N.sqrt(N.sum((-AB + AC) ** 2))
    sp = (l1 + l2 + l3) / 2.
    area = N.sqrt(sp * (sp - l1) * (sp - l2) * (sp - l3))
    if flux is not None:
        cosangle = 2. * N.arcsin(0.5 * N.sqrt(N.sum((direction - normal) ** 2)))
        energy = area / num_rays * flux * N.cos(cosangle)
    else:
        energy = 1. / float(num_rays) / procs
    rot_pos = N.zeros((3, 3))
    rot_pos[:, 0] = AB
    rot_pos[:, 1] = AC
    desc = _fill_source(_cabi.SRC_PILLBOX_TRIANGLE, A, rot_pos, rotation_to_z(direction), [ang_range], energy)
    return _new_bundle(desc, num_rays, seed, ray_offset)


def vf_cylinder_bundle(num_rays, rc, lc, center, direction, flux=None, rays_in=True, angular_span=[0., 2. * N.pi],
                       ang_range=N.pi / 2., seed=None, ray_offset=0):
    """
    Lambertian emitter on the wall of a cylinder of radius rc and length lc centred on `center` with its axis along
    `direction`, firing towards the axis (rays_in) or away from it (sources.py:716-769; the view-factor workload of
    emissive_losses).  Energies flux*area/N, or 1/N without flux.
    """
    rc, lc = float(rc), float(lc)
    if flux is None:
        energy = 1. / float(num_rays)
    else:
        energy = flux * rc * (angular_span[1] - angular_span[0]) * lc / float(num_rays)
    rot = rotation_to_z(direction)
    desc = _fill_source(_cabi.SRC_VF_CYLINDER, center, rot, rot,
                        [rc, lc, angular_span[0], angular_span[1], ang_range, 1. if rays_in else -1.], energy)
    return _new_bundle(desc, num_rays, seed, ray_offset)


def vf_frustum_bundle(num_rays, r0, r1, depth, center, direction, flux=None, rays_in=True, angular_span=[0., 2. * N.pi],
                      angular_range=N.pi / 2., seed=None, ray_offset=0):
    """
    Lambertian emitter on the wall of a frustum: radius r0 at the base centred on `center`, r1 at `depth` along
    `direction` (sources.py:644-714).  r0 == r1 is a cylinder and must use vf_cylinder_bundle (the reference divides
    by the wall slope).
    """
    r0, r1, depth = float(r0), float(r1), float(depth)
    if r0 == r1:
        raise ValueError("vf_frustum_bundle needs r0 != r1; use vf_cylinder_bundle for a cylinder")
    if flux is None:
        energy = 1. / float(num_rays)
    else:
        area = (angular_span[1] - angular_span[0]) * (r1 + r0) / 2. * N.sqrt(abs(r1 - r0) ** 2. + depth ** 2.)
        energy = flux * area / float(num_rays)
    rot = rotation_to_z(direction)
    desc = _fill_source(_cabi.SRC_VF_FRUSTUM, center, rot, rot,
                        [r0, r1, depth, angular_span[0], angular_span[1], angular_range, 1. if rays_in else -1.], energy)
    return _new_bundle(desc, num_rays, seed, ray_offset)


def Lambertian_directions(num_rays, ang_range, normals=None):
    """
    (3, num_rays) unit directions about +z (or about `normals`), cosine-weighted within ang_range of it
    (sources.py:88-101).  A host sampler on numpy's global generator, drawing what the reference draws in its order -- the
    bundles above sample on the device from Philox streams instead.
    """
    xi1 = N.random.uniform(low=0., high=2. * N.pi, size=num_rays)
    if ang_range == 0.:
        dirs = N.zeros((3, num_rays))
        dirs[2] = 1.
    else:
        xi2 = N.random.uniform(size=num_rays)
        sinsqrt = N.sin(ang_range) * N.sqrt(xi2)
        dirs = N.vstack((N.cos(xi1) * sinsqrt, N.sin(xi1) * sinsqrt, N.sqrt(1. - sinsqrt ** 2.)))
    if normals is not None:
        from .vector_manipulations import rotate_z_to_normal
        dirs = rotate_z_to_normal(dirs, normals)
    return dirs


def pillbox_sunshape_directions(num_rays, ang_range):
    """pillbox sunshape about +z: the cone-limited Lambertian distribution (sources.py:103-117)"""
    return Lambertian_directions(num_rays, ang_range)


def edge_rays_directions(num_rays, ang_range):
    """directions on the rim of the cone of half-angle ang_range about +z (sources.py:152-173)"""
    xi1 = N.random.uniform(high=2. * N.pi, size=num_rays)
    sin_th = N.ones(num_rays) * N.sin(ang_range)
    return N.vstack((N.cos(xi1) * sin_th, N.sin(xi1) * sin_th, N.cos(N.ones(num_rays) * ang_range)))


def edge_rays_bundle(num_rays, center, direction, radius, ang_range, flux=None, radius_in=0.):
    """annular disc source whose rays all leave at exactly ang_range from `direction` (sources.py:304-328); host-generated"""
    radius, radius_in = float(radius), float(radius_in)
    a = edge_rays_directions(num_rays, ang_range)
    perp_rot = rotation_to_z(direction)
    directions = N.sum(perp_rot[..., None] * a[None, ...], axis=1)
    xi1 = N.random.uniform(size=num_rays)
    thetas = N.random.uniform(high=2. * N.pi, size=num_rays)
    rs = N.sqrt(radius_in ** 2. + xi1 * (radius ** 2. - radius_in ** 2.))
    vertices_local = N.vstack((rs * N.cos(thetas), rs * N.sin(thetas), N.zeros(num_rays)))
    rayb = RayBundle(vertices=N.dot(perp_rot, vertices_local) + center, directions=directions)
    if flux is not None:
        rayb.set_energy(N.pi * (radius ** 2. - radius_in ** 2.) / num_rays * flux * N.ones(num_rays))
    return rayb


def trapezoid_bundle(num_rays, A, B, C, direction=None, ang_range=N.pi / 2., flux=None, procs=1, seed=None, ray_offset=0):
    """
    Isosceles trapezoid ABCD (AB the first base, C the third vertex, D by symmetry) as two triangular bundles sharing the
    rays in proportion to their areas (sources.py:599-642).
    """
    A, B, C = [N.asarray(v, dtype=float) for v in (A, B, C)]
    AB, AC = B - A, C - A
    l1, l2 = N.sqrt(N.sum(AB ** 2)), N.sqrt(N.sum(AC ** 2))
    cos_theta = N.dot(AC, AB) / (l1 * l2)
    cB = AB * (1. - 1. / l1 * l2 * cos_theta)
    AD = AC - (AB - 2. * cB)
    D = A + AD
    l3, l4, l5 = N.sqrt(N.sum(AD ** 2)), N.sqrt(N.sum((AC - AB) ** 2)), N.sqrt(N.sum((AD - AC) ** 2))
    s1, s2 = (l1 + l2 + l4) / 2., (l2 + l3 + l5) / 2.
    area_ABC = N.sqrt(s1 * (s1 - l1) * (s1 - l2) * (s1 - l4))      # Heron
    area_ACD = N.sqrt(s2 * (s2 - l2) * (s2 - l3) * (s2 - l5))
    n_ABC = int(area_ABC / (area_ABC + area_ACD) * num_rays)
    first = triangular_bundle(n_ABC, A, B, C, direction, ang_range, flux, seed=seed, ray_offset=ray_offset)
    second = triangular_bundle(num_rays - n_ABC, A, C, D, direction, ang_range, flux, seed=seed, ray_offset=ray_offset + n_ABC)
    rayb = concatenate_rays([first, second])
    if flux is None:
        rayb.set_energy(N.ones(num_rays) / float(num_rays) / procs)
    return rayb


def regular_square_bundle(num_rays, center, direction, width):
    """Parallel rays on a regular square grid of half-width `width` normal to `direction` (sources.py:518-542);
    deterministic, no energy column -- built on the host."""
    direction = N.asarray(direction, dtype=float)
    rot = rotation_to_z(direction)
    rng_ = N.s_[-width:width:float(2 * width) / N.sqrt(num_rays)]
    xs, ys = N.mgrid[rng_, rng_]
    local = N.array([xs.flatten(), ys.flatten(), N.zeros(len(xs.flatten()))])
    rayb = RayBundle()
    rayb.set_vertices(N.dot(rot, local) + center)
    rayb.set_directions(N.tile(direction[:, None], (1, local.shape[1])))
    return rayb


def solar_disk_bundle(num_rays, center, direction, radius, ang_range, flux=None, radius_in=0., angular_span=[0., 2. * N.pi],
                      procs=1, seed=None, ray_offset=0):
    """Older name of disk_bundle still used by scene scripts."""
    return disk_bundle(num_rays, center, direction, radius, ang_range, flux, radius_in, angular_span, None, procs,
                       None, seed, ray_offset)


_buie_tables = {}


def buie_table(CSR, pre_process_CSR=True):
    """the table of _buie_table, computed once per (CSR, pre_process_CSR): Monte-Carlo loops make one bundle per batch"""
    key = (float(CSR), bool(pre_process_CSR))
    if key not in _buie_tables:
        if len(_buie_tables) > 64:
            _buie_tables.clear()
        _buie_tables[key] = _buie_table(*key)
    return _buie_tables[key]


def _buie_table(CSR, pre_process_CSR=True):
    """
    The Buie sunshape sampling table of the reference (sources.py:333-361): 211 polar angles up to
    4.65 mrad, g = phi*cos*sin with phi = cos(0.326 theta)/cos(0.308 theta) (theta in mrad), the
    trapezoid CDF of the disc part, and the aureole constants kappa, gamma for CSR > 0.
    Layout: trc_source_desc.buie.
    """
    theta_dni = 4.65e-3
    theta_tot = 43.6e-3
    nelem = _cabi.TRC_BUIE_NELEM
    theta = N.linspace(0., theta_dni, nelem + 1)
    phi = N.cos(0.326 * theta * 1e3) / N.cos(0.308 * theta * 1e3)
    g = phi * N.cos(theta) * N.sin(theta)
    integ = 0.5 * (phi[:-1] * N.cos(theta[:-1]) * N.sin(theta[:-1]) + phi[1:] * N.cos(theta[1:]) * N.sin(theta[1:])) * \
        (theta[1:] - theta[:-1])
    I_dni = N.sum(integ)
    gamma = kappa = 0.
    if CSR == 0.:
        total = I_dni
    else:
        if pre_process_CSR:
            if CSR <= 0.1:
                CSR = -2.245e+03 * CSR ** 4. + 5.207e+02 * CSR ** 3. - 3.939e+01 * CSR ** 2. + 1.891e+00 * CSR + 8e-03
            else:
                CSR = 1.973 * CSR ** 4. - 2.481 * CSR ** 3. + 0.607 * CSR ** 2. + 1.151 * CSR - 0.020
        kappa = 0.9 * N.log(13.5 * CSR) * CSR ** (-0.3)
        gamma = 2.2 * N.log(0.52 * CSR) * CSR ** (0.43) - 0.1
        I_csr = 1e-6 * N.exp(kappa) / (gamma + 2.) * ((theta_tot * 1000.) ** (gamma + 2.) - (theta_dni * 1000.) ** (gamma + 2.))
        total = I_dni + I_csr
    cdf = N.add.accumulate(N.hstack(([0], integ / total)))
    return N.concatenate((theta, g, cdf, [I_dni, gamma, kappa, theta_dni, theta_tot, 1. if CSR > 0. else 0.]))


def buie_sunshape(num_rays, center, direction, radius, CSR, flux=None, pre_process_CSR=True, rays_direction=None,
                  seed=None, ray_offset=0):
    """
    Disc source with the Buie sunshape (sources.py:412-464): start points uniform on a disc of
    `radius` normal to `direction`, directions about `rays_direction` (default `direction`).
    """
    direction = N.asarray(direction, dtype=float)
    if rays_direction is None:
        rays_direction = direction
    energy = flux * (N.pi * radius ** 2.) / num_rays * _tilt_cos(rays_direction, direction)
    desc = _fill_source(_cabi.SRC_BUIE_DISK, center, rotation_to_z(direction), rotation_to_z(rays_direction),
                        [radius], energy, buie_table(CSR, pre_process_CSR))
    return _new_bundle(desc, num_rays, seed, ray_offset)


def rect_buie_sunshape(num_rays, center, direction, width, height, CSR, flux=None, pre_process_CSR=True,
                       rays_direction=None, seed=None, ray_offset=0):
    """Rectangular source with the Buie sunshape (sources.py:466-515)."""
    direction = N.asarray(direction, dtype=float)
    if rays_direction is None:
        rays_direction = direction
    energy = flux * (width * height) / num_rays * _tilt_cos(rays_direction, direction)
    desc = _fill_source(_cabi.SRC_BUIE_RECT, center, rotation_to_z(direction), rotation_to_z(rays_direction),
                        [width, height], energy, buie_table(CSR, pre_process_CSR))
    return _new_bundle(desc, num_rays, seed, ray_offset)


def single_ray_source(position, direction, flux=None):
    """One ray (sources.py:68-86)."""
    d = N.array(direction, dtype=float).reshape(3, 1)
    d /= N.sqrt(N.sum(d ** 2, axis=0))
    b = RayBundle(vertices=N.asarray(position, dtype=float).reshape(3, 1), directions=d)
    b.set_energy(flux * N.ones(1))
    return b
