"""
The benchmark / parity scenes of BASELINE.json, restated with the current public API from the
parameters given in SURVEY.md section 8(d) (the reference's own example scripts are Python-2 and
API-stale and cannot run).  Used by bench.py, __graft_entry__.smoke() and the tests.

  nsttf_field()   Sandia NSTTF: 218 heliostats 6.1 x 6.1 m (absorptivity 0.04, sigma 1 mrad
                  bi-variate, 'fast' option), all aimed at (0,0,60), sun azimuth 0 / zenith 35.05 deg,
                  11 x 11 m one-sided receiver at z = 60 facing the field, Buie sunshape CSR 0.01.
                  Parameters: examples/Sandia_NSTTF_field example.py:31-36, :82-112, :135-182 and
                  examples/sandia_hstat_coordinates.csv (data file shipped as
                  tracer_amd/data/sandia_hstat_coordinates.csv).
  dish()          parabolic dish D=5 f=3 + round receiver r=0.15 at the focus, Buie CSR 0.05 (config 2).
  flat_pair()     2x2 flat mirror + 4x4 Lambertian receiver, pillbox rect source (config 1).
"""
import os

import numpy as N

from .assembly import Assembly
from .object import AssembledObject
from .surface import Surface
from .flat_surface import RectPlateGM, RoundPlateGM
from .paraboloid import ParabolicDishGM
from . import optics_callables as opt
from .spatial_geometry import rotx, translate
from .models.heliostat_field import HeliostatField, solar_vector
from .models.one_sided_mirror import one_sided_receiver
from . import sources

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data')


def nsttf_positions():
    pos = N.loadtxt(os.path.join(_DATA, 'sandia_hstat_coordinates.csv'), delimiter=',')
    pos[:, 1] -= 4.   # examples/Sandia_NSTTF_field example.py:84
    return pos


def nsttf_field(sigma=1e-3, n_heliostats=None):
    """Returns (plant Assembly, field, receiver object, source-argument dict)."""
    pos = nsttf_positions()
    if n_heliostats is not None:
        pos = pos[:n_heliostats]
    field = HeliostatField(pos, 6.1, 6.1, absorptivity=0.04, sigma=sigma, bi_var=True, MCRT_option='fast')
    aim = N.tile(N.array([0., 0., 60.]), (pos.shape[0], 1))
    zenith = 35.05 * N.pi / 180.
    field.track_sun(0., zenith, aim_points=aim)
    rec = one_sided_receiver(11., 11.)
    rec.set_transform(N.dot(translate(0., 0., 60.), rotx(-N.pi / 2.)))
    plant = Assembly(objects=[rec], subassemblies=[field])
    sun = solar_vector(0., zenith)
    x0, x1 = pos[:, 0].min(), pos[:, 0].max()
    y0, y1 = pos[:, 1].min(), pos[:, 1].max()
    centre = N.array([(x0 + x1) / 2., (y0 + y1) / 2., 0.])
    radius = 1.10 * N.sqrt(((x1 - x0) / 2.) ** 2 + ((y1 - y0) / 2.) ** 2)
    src = dict(center=N.vstack(300. * sun + centre), direction=-sun, radius=radius, CSR=0.01, flux=1000.,
               pre_process_CSR=False)
    return plant, field, rec, src


def nsttf_source(n, src, seed=None, ray_offset=0, n_total=None):
    """n rays of the field's source; n_total: they are a part of a bundle of n_total rays traced elsewhere as well (ranks that
    share a bundle): every ray carries flux x area / n_total"""
    flux = src['flux'] if n_total is None else src['flux'] * (float(n) / float(n_total))
    return sources.buie_sunshape(n, src['center'], src['direction'], src['radius'], src['CSR'], flux=flux,
                                 pre_process_CSR=src['pre_process_CSR'], seed=seed, ray_offset=ray_offset)


def nsttf_fluxmap_edges(bins=50):
    e = N.linspace(-5.5, 5.5, bins + 1)
    return e, e


def dish(sigma=2e-3):
    dish_surf = Surface(ParabolicDishGM(5., 3.), opt.RealReflective(0.06, sigma, bi_var=False))
    dish_obj = AssembledObject(surfs=[dish_surf])
    rec_surf = Surface(RoundPlateGM(0.15), opt.LambertianReceiver(1.))
    rec_obj = AssembledObject(surfs=[rec_surf], transform=N.dot(translate(0., 0., 3.), rotx(N.pi)))
    asm = Assembly(objects=[dish_obj, rec_obj])
    src = dict(center=N.c_[[0., 0., 6.]], direction=N.r_[0., 0., -1.], radius=2.5, CSR=0.05, flux=1000.)
    return asm, dish_surf, rec_surf, src


def dish_source(n, src, seed=None, ray_offset=0):
    return sources.buie_sunshape(n, src['center'], src['direction'], src['radius'], src['CSR'], flux=src['flux'],
                                 seed=seed, ray_offset=ray_offset)


def flat_pair():
    d = N.r_[-0.15, 0., -1.]
    d = d / N.sqrt(N.sum(d ** 2))
    mirror = Surface(RectPlateGM(2., 2.), opt.RealReflective(0.05, 2e-3, bi_var=True))
    m_obj = AssembledObject(surfs=[mirror])
    out = d - 2. * d[2] * N.r_[0., 0., 1.]          # specular direction off the z=0 mirror
    rec = Surface(RectPlateGM(4., 4.), opt.LambertianReceiver(1.))
    r_obj = AssembledObject(surfs=[rec], transform=N.dot(translate(*(out * 10. / out[2])), rotx(N.pi)))
    asm = Assembly(objects=[m_obj, r_obj])
    src = dict(center=N.vstack(-d * 5.), direction=d, x=2., y=2., ang_range=4.65e-3, flux=1000.)
    return asm, mirror, rec, src


def flat_pair_source(n, src, seed=None, ray_offset=0):
    return sources.rect_bundle(n, src['center'], src['direction'], src['x'], src['y'], src['ang_range'],
                               flux=src['flux'], seed=seed, ray_offset=ray_offset)


def cavity_arrays():
    """
    A cavity receiver in the manner of the reference's tracer/models/Two_N_parameters_cavity.py:87-152, as scene-table arrays
    (dict for scene.TableScene): aperture annulus at z = 0 (inner radius 1), frustum 0 < z < 1 (radius 1 -> 1.4), cylinder
    1 < z < 2.5, cone back (apex at z = 3.5), a metal ring plate and a spectrally selective mirror disc inside -- walls with
    angle- and wavelength-dependent Lambertian optics, a Fresnel conductor, a spectral mirror: every table-driven optics kind.
    Rays need a `wavelengths` column.
    """
    from . import _cabi as K
    ths = N.linspace(0., N.pi / 2., 7)
    abth = N.array([0.9, 0.88, 0.85, 0.8, 0.7, 0.5, 0.1])
    wls = N.linspace(0.25e-6, 2.6e-6, 5)
    grid = 0.2 + 0.7 * N.outer(N.cos(ths) ** 0.5, 1. / (1. + (wls * 1e6 - 1.) ** 2))
    mlam = N.linspace(0.2e-6, 3e-6, 8)
    mn = N.array([0.1, 0.13, 0.2, 0.4, 0.9, 1.5, 2.4, 3.6])
    mk = N.array([2.0, 3.5, 5.0, 7.0, 9.5, 13., 18., 24.])
    slam = N.linspace(0.2e-6, 3e-6, 9)
    sab = N.array([0.1, 0.2, 0.15, 0.4, 0.9, 0.5, 0.3, 0.2, 0.25])
    tables = [N.concatenate((ths, abth)), N.concatenate(([len(ths), len(wls)], ths, wls, grid.ravel())),
              N.concatenate((mlam, mn, mk)), N.concatenate((slam, sab))]
    offs = N.concatenate(([0], N.cumsum([len(t) for t in tables])))
    gm_kind = [K.GM_FRUSTUM, K.GM_CYL_FINITE, K.GM_CONE_FINITE, K.GM_ROUND, K.GM_ROUND, K.GM_ROUND]
    frames = [N.eye(4), translate(0, 0, 1.75), N.dot(translate(0, 0, 3.5), rotx(N.pi)), N.eye(4), translate(0, 0, 2.2), translate(0.2, 0., 1.2)]
    gm = N.zeros((6, 16))
    gm[0, :4] = (1.4 - 1.0) / (1.0 - 0.0), (1.4 * 0. - 1.0 * 1.0) / (1.4 - 1.0), 0., 1.
    gm[1, :4] = 1.4, 0.75, 0., 2. * N.pi
    gm[2, :3] = 1.4 / 1.0, 0., 1.0
    gm[3, :2] = 1.6, 1.0
    gm[4, :2] = 0.6, 0.2
    gm[5, :2] = 0.3, -1.
    ok = [K.OPT_LAMBERTIAN_DIRECTIONAL_SPECTRAL, K.OPT_LAMBERTIAN_DIRECTIONAL, K.OPT_LAMBERTIAN_DIRECTIONAL_SPECTRAL, K.OPT_SEMI_LAMBERTIAN,
          K.OPT_FRESNEL_CONDUCTOR, K.OPT_REFLECTIVE_SPECTRAL]
    opt_p = N.zeros((6, 8))
    opt_p[3, :2] = 0.6, 0.9           # the aperture annulus: mirror beyond 0.9 rad of incidence, Lambertian (into 0.9 rad) below
    opt_p[4, 0] = 1.0
    which = [1, 0, 1, -1, 2, 3]
    return dict(gm_kind=N.array(gm_kind, dtype=N.int32), optics_kind=N.array(ok, dtype=N.int32), frames=N.array(frames), gm=gm, opt=opt_p,
                extra=N.concatenate(tables), extra_off=N.array([offs[w] if w >= 0 else -1 for w in which]),
                extra_len=N.array([len(tables[w]) if w >= 0 else 0 for w in which]))


def dish_cavity(sigma=2e-3, scale=0.12):
    """
    The fifth configuration of BASELINE.json on one rank: the dish of `dish()` (slope error `sigma`) focusing into the cavity of
    `cavity_arrays()` scaled by `scale` (aperture radius 0.12 m) with its aperture in the focal plane, axis along the dish's.
    Returns (TableScene, src): rays start on a disc between dish and cavity (the cavity does not shade it) and need wavelengths.
    """
    from .scene import TableScene, compile_scene, scene_arrays
    d = scene_arrays(compile_scene(dish(sigma)[0]))
    c = cavity_arrays()
    S = N.diag([scale, scale, scale, 1.])
    frames = [N.dot(translate(0., 0., 3.), N.dot(S, f)) for f in c['frames']]       # scaled copy: frames carry the scale ...
    for f in frames:                                                               # ... as lengths, rotations stay orthonormal
        f[:3, :3] /= scale
    gm = c['gm'].copy()
    gm[0, 2:4] *= scale; gm[0, 1] *= scale                  # frustum: z range and apex offset (the slope c is a ratio)
    gm[1, :2] *= scale                                      # cylinder: radius, half height
    gm[2, 2] *= scale                                       # cone: height (its slope is a ratio)
    gm[3, :2] *= scale; gm[4, :2] *= scale; gm[5, 0] *= scale
    cat = lambda a, b: N.concatenate((a[:1], b))
    ts = TableScene(cat(d['gm_kind'], c['gm_kind']), cat(d['optics_kind'], c['optics_kind']), [d['frames'][0]] + frames,
                    N.vstack((d['gm'][:1], gm)), N.vstack((d['opt'][:1], c['opt'])), c['extra'],
                    N.concatenate(([-1], c['extra_off'])), N.concatenate(([0], c['extra_len'])))
    src = dict(center=N.c_[[0., 0., 2.9]], direction=N.r_[0., 0., -1.], radius=2.5, CSR=0.05, flux=1000.)
    return ts, src


def relief_mesh(m=230, extent=10., amp=0.8):
    """
    SURVEY 8(f)4, a mesh as the reference builds it (models/triangulated_surface.py:12-52, one face per triangle) beyond what LDS holds:
    m x m quads over [-extent, extent]^2 on a relief steep enough for second and third bounces, two mirror faces each (absorptivity
    0.2; 105 800 triangles at m = 230), under a black lid 50 m above.  Returns (assembly, faces, source) -- source: (center (3, 1),
    direction, radius, CSR) of a Buie sun that fills the relief.
    """
    from .models.triangulated_surface import TriangulatedSurface
    x, y = N.meshgrid(N.linspace(-extent, extent, m + 1), N.linspace(-extent, extent, m + 1), indexing='ij')
    z = amp * N.sin(1.3 * x) * N.cos(1.1 * y)
    V = N.c_[x.ravel(), y.ravel(), z.ravel()]
    i, j = N.meshgrid(N.arange(m), N.arange(m), indexing='ij')
    a, b, c, d = (i * (m + 1) + j).ravel(), ((i + 1) * (m + 1) + j).ravel(), ((i + 1) * (m + 1) + j + 1).ravel(), (i * (m + 1) + j + 1).ravel()
    F = N.vstack((N.c_[a, b, c], N.c_[a, c, d]))
    mesh = TriangulatedSurface(V, F, opt.Reflective(0.2))
    lid = AssembledObject(surfs=[Surface(RectPlateGM(90., 90.), opt.LambertianReceiver(1.))], transform=N.dot(translate(0., 0., 50.), rotx(N.pi)))
    direction = N.r_[0.1, -0.05, -1.] / N.linalg.norm([0.1, -0.05, -1.])
    return Assembly(objects=[mesh, lid]), len(F), (N.c_[-40. * direction], direction, 12., 0.05)
