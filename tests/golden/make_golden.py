"""
Generate the golden fixtures by running the REAL reference (casselineau/Tracer at /root/reference).

Run in the build container only (the reference never travels):
    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden.py
It imports the reference unmodified (a 3-attribute stub stands in for the absent `shapely`, which only
polygon sampling / polygon meshes touch -- SURVEY.md section 8(c)) next to the tracer_amd host classes, builds
every case twice from the same constructor arguments (reference classes -> expected outputs; tracer_amd
classes -> the parameter table that is the input of the oracle and of the C-ABI) and writes small .npz/.json
files into tests/golden/.  A fixture is data: inputs and the reference's outputs.
"""
import importlib
import json
import os
import sys
import types

import numpy as N

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = '/root/reference'


def import_reference():
    stub = types.ModuleType('shapely')
    stub.Polygon = stub.constrained_delaunay_triangles = stub.MultiPolygon = None
    sys.modules.setdefault('shapely', stub)
    sys.dont_write_bytecode = True
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)


class NS(object):
    """The same module names in either package."""
    def __init__(self, pkg):
        self.pkg = pkg

    def __getattr__(self, name):
        return importlib.import_module(self.pkg + '.' + name)


def rot(axis, ang):
    import tracer_amd.spatial_geometry as sg
    return sg.general_axis_rotation(N.asarray(axis, dtype=float) / N.linalg.norm(axis), ang)


def frame_of(R, c):
    f = N.eye(4)
    f[:3, :3] = R
    f[:3, 3] = c
    return f


# ---------------------------------------------------------------------------------------------------------
# 1. geometry: (frame, params, rays) -> (t, hit points, normals) for every native kind
# ---------------------------------------------------------------------------------------------------------
def gm_cases():
    """(name, module, class, args, kwargs, scale): constructor arguments valid in both packages."""
    tri = N.array([[1.2, 0.], [0.1, 0.9], [0., 0.]])
    return [
        ('flat', 'flat_surface', 'FlatGeometryManager', (), {}, 2.),
        ('rect', 'flat_surface', 'RectPlateGM', (1.5, 0.8), {}, 1.5),
        ('rect_extruded', 'flat_surface', 'ExtrudedRectPlateGM', (2., 1.6, N.c_[[0.2, -0.1]], 0.5, 0.4), {}, 1.5),
        ('rect_perforated', 'flat_surface', 'PerforatedRectPlateGM',
         (2., 2., N.array([[0.3, 0.3], [-0.4, 0.1], [0., -0.5]]), N.array([0.2, 0.15, 0.25])), {}, 1.5),
        ('round', 'flat_surface', 'RoundPlateGM', (1.,), {}, 1.5),
        ('annulus', 'flat_surface', 'RoundPlateGM', (1., 0.4), {}, 1.5),
        ('round_cut', 'flat_surface', 'StraightCutRoundPlateGM', (1., 0.3), {}, 1.5),
        ('triangle', 'triangular_face', 'TriangularFace', (tri,), {}, 1.5),
        ('paraboloid', 'paraboloid', 'Paraboloid', (1.3, 0.9), {}, 1.5),
        ('parab_dish', 'paraboloid', 'ParabolicDishGM', (2., 1.5), {}, 1.5),
        ('parab_hex', 'paraboloid', 'HexagonalParabolicDishGM', (2., 1.5), {}, 1.5),
        ('parab_rect', 'paraboloid', 'RectangularParabolicDishGM', (1.6, 1.2, 2.), {}, 1.5),
        ('parab_rect_offaxis', 'paraboloid', 'RectangularParabolicDishGM', (1.0, 0.8, 3.),
         {'off_axis_normal': N.array([N.sin(0.2) * N.cos(0.4), N.sin(0.2) * N.sin(0.4), N.cos(0.2)])}, 2.5),
        ('parab_cyl', 'paraboloid', 'ParabolicCylinder', (1.2,), {}, 1.5),
        ('parab_trough', 'paraboloid', 'ParabolicTroughGM', (2., 1., 3.), {}, 2.),
        ('sphere', 'sphere_surface', 'SphericalGM', (1.1,), {}, 1.5),
        ('hemisphere', 'sphere_surface', 'HemisphereGM', (1.1,), {}, 1.5),
        ('sphere_rect', 'sphere_surface', 'SphericalRectFacet', (2., 1.2, 0.9), {}, 1.5),
        ('cyl_inf', 'cylinder', 'InfiniteCylinder', (1.4,), {}, 1.5),
        ('cyl_finite', 'cylinder', 'FiniteCylinder', (1.4, 2.), {}, 1.5),
        ('cyl_finite_arc', 'cylinder', 'FiniteCylinder', (1.4, 2.), {'ang_range': [0.5, 4.0]}, 1.5),
        ('cyl_rectcut', 'cylinder', 'RectCutCylinder', (1.4, 2., 1.2, 1.1), {}, 1.5),
        ('cone_inf', 'cone', 'InfiniteCone', (0.7,), {'a': 0.3}, 1.5),
        ('cone_finite', 'cone', 'FiniteCone', (0.8, 1.5), {}, 1.5),
        ('frustum', 'cone', 'ConicalFrustum', (0.2, 0.5, 1.4, 1.1), {}, 1.5),
        ('frustum_rectcut', 'cone', 'RectCutConicalFrustum', (0.2, 0.5, 1.4, 1.1, 1.2, 1.3), {}, 1.5),
        ('quadratic', 'quadratic_surface', 'FlatQuadricSurfaceGM', (0.05, 0.08, 0.01, 0.02, -0.03, 0.1), {}, 1.5),
        ('quadratic_rect', 'quadratic_surface', 'RectFlatQuadricSurfaceGM', (2., 1.5, 0.05, 0.08, 0.01, 0.02, -0.03, 0.1), {}, 1.5),
        ('ellipsoid', 'ellipsoid', 'Ellipsoid', (1.2, 0.8, 1.5), {}, 1.5),
        ('ellipsoid_cut', 'ellipsoid', 'EllipsoidGM', (1.2, 0.8, 1.5), {'xlim': [-0.5, 1.0], 'ylim': None, 'zlim': [-1., 0.7]}, 1.5),
        ('ellipsoid_alllims', 'ellipsoid', 'EllipsoidGM', (1.2, 0.8, 1.5), {'xlim': [-0.5, 1.0], 'ylim': [-0.3, 0.3], 'zlim': [-1., 0.7]}, 1.5),
        # polygons: a concave clockwise L with a notch (vertices level with random hits do not occur; edge rules are exercised by
        # the vertical and horizontal edges), a convex pentagon, and the L with three circular perforations
        ('polygon_L', 'polygon', 'FlatSimplePolygonGM', (N.array([[-1., -1., 0.2, 0.2, 0.6, 1.2, 1.2], [-0.8, 1., 1., 0.1, 0.3, 0.1, -0.8]]),), {}, 1.5),
        ('polygon_pentagon', 'polygon', 'FlatSimplePolygonGM',
         (N.array([[N.cos(-2. * N.pi * k / 5. + 0.3) for k in range(5)], [0.8 * N.sin(-2. * N.pi * k / 5. + 0.3) for k in range(5)]]),), {}, 1.2),
        ('polygon_perforated', 'polygon', 'PerforatedPolygonGM',
         (N.array([[-1., -1., 0.2, 0.2, 0.6, 1.2, 1.2], [-0.8, 1., 1., 0.1, 0.3, 0.1, -0.8]]),
          N.array([[-0.5, 0.4], [0.7, -0.4], [-0.3, -0.5]]), N.array([0.3, 0.2, 0.15])), {}, 1.5),
        # CutSphereGM: the bounding volume is built per package (callable values get the package namespace)
        ('sphere_cut_plane', 'sphere_surface', 'CutSphereGM', (1.3,),
         {'bounding_volume': lambda pkg: pkg.boundary_shape.BoundaryPlane(location=N.r_[0.1, 0., 0.4], rotation=rot([1., 0.3, 0.], 0.5))}, 1.5),
        ('sphere_cut_plane_lens_cap', 'sphere_surface', 'CutSphereGM', (2.,),
         {'bounding_volume': lambda pkg: pkg.boundary_shape.BoundaryPlane(location=N.r_[0., 0., 1.2])}, 2.),
        ('sphere_cut_sphere', 'sphere_surface', 'CutSphereGM', (2.,),
         {'bounding_volume': lambda pkg: pkg.boundary_shape.BoundarySphere(radius=4., location=N.r_[0., 0., -4 * N.sqrt(3) / 2.])}, 2.),
    ]


def ray_fan(rng, n, frame, scale):
    """Rays aimed at points near the surface from outside, from inside (origin within the shape), grazing ones,
    and a few exactly axis-parallel ones -- in global coordinates of `frame`."""
    c = frame[:3, 3]
    R = frame[:3, :3]
    k = n // 4
    targets = N.dot(R, (rng.uniform(-1., 1., size=(3, n)) * scale * N.array([[1.], [1.], [0.6]]))) + c[:, None]
    origins = N.empty((3, n))
    origins[:, :k] = N.dot(R, rng.uniform(-1, 1, size=(3, k)) * 4. * scale + N.array([[0.], [0.], [5. * scale]])) + c[:, None]   # above
    origins[:, k:2 * k] = N.dot(R, rng.uniform(-1, 1, size=(3, k)) * 6. * scale) + c[:, None]                                  # around
    origins[:, 2 * k:3 * k] = N.dot(R, rng.uniform(-0.3, 0.3, size=(3, k)) * scale) + c[:, None]                              # inside
    graz = rng.uniform(-1, 1, size=(3, n - 3 * k)) * 5. * scale
    graz[2] *= 0.02
    origins[:, 3 * k:] = N.dot(R, graz) + c[:, None]                                                                         # grazing
    d = targets - origins
    d[:, -3:] = N.dot(R, N.array([[0., 0., 1.], [1., 0., 0.], [0., 0., -1.]]).T)      # exactly along local axes
    d /= N.sqrt(N.sum(d ** 2, axis=0))
    return origins, d


def make_geometry(ref, amd, out):
    # CutSphereGM._select_coords still says `xrange` (sphere_surface.py:198); give the imported module the Python-3 name
    ref.sphere_surface.xrange = range
    rng = N.random.RandomState(20240601)
    frames = [N.eye(4),
              frame_of(rot([1, 2, 3], 0.7), [0.5, -1.0, 2.0]),
              frame_of(rot([-1, 0.5, 0.2], 2.4), [-30., 12., 7.])]
    index = []
    ci = 0
    for name, mod, cls, args, kwargs, scale in gm_cases():
        for fi, frame in enumerate(frames):
            if name == 'sphere_cut_sphere' and fi > 0:
                continue        # the reference's BoundarySphere ignores frame transforms (keeps `_loc`, boundary_shape.py:101-110)
            gm_ref = getattr(getattr(ref, mod), cls)(*args, **dict((k, a(ref) if callable(a) else a) for k, a in kwargs.items()))
            gm_amd = getattr(getattr(amd, mod), cls)(*args, **dict((k, a(amd) if callable(a) else a) for k, a in kwargs.items()))
            kind, params, extra = gm_amd._native()
            v, d = ray_fan(rng, 240, frame, scale)
            if cls in ('HemisphereGM', 'SphericalRectFacet', 'CutSphereGM'):
                # Reference defect: these two classes assign the chosen root with
                # `N.nonzero(mask[:, one_hit])[0]` (sphere_surface.py:137, :227), which lists the root indices
                # sorted, not per ray -- in a bundle where some rays keep root 0 and others root 1 the choices are
                # shuffled between rays.  One ray at a time the same code is well defined, so that is how the
                # expected values are produced here.
                t = N.empty(v.shape[1])
                normals = []
                pts = []
                for r in range(v.shape[1]):
                    b1 = ref.ray_bundle.RayBundle(vertices=v[:, r:r + 1].copy(), directions=d[:, r:r + 1].copy())
                    with N.errstate(all='ignore'):
                        t[r] = gm_ref.find_intersections(frame, b1)[0]
                        if N.isfinite(t[r]):
                            gm_ref.select_rays(N.array([0]))
                            normals.append(N.array(gm_ref.get_normals()))
                            pts.append(N.array(gm_ref.get_intersection_points_global()))
                    gm_ref.done()
                    if hasattr(gm_ref, '_params'):
                        del gm_ref._params
                hit_idx = N.nonzero(N.isfinite(t))[0]
                normals = N.hstack(normals) if normals else N.zeros((3, 0))
                pts = N.hstack(pts) if pts else N.zeros((3, 0))
            else:
                bund = ref.ray_bundle.RayBundle(vertices=v.copy(), directions=d.copy())
                with N.errstate(all='ignore'):
                    t = N.array(gm_ref.find_intersections(frame, bund), dtype=float)
                    hit_idx = N.nonzero(N.isfinite(t))[0]
                    gm_ref.select_rays(hit_idx)
                    normals = N.array(gm_ref.get_normals()) if len(hit_idx) else N.zeros((3, 0))
                    pts = N.array(gm_ref.get_intersection_points_global()) if len(hit_idx) else N.zeros((3, 0))
                gm_ref.done()
            pre = 'g%d_' % ci
            gm16 = N.zeros(16)
            gm16[:len(params)] = params
            out[pre + 'kind'] = N.int32(kind)
            out[pre + 'frame'] = frame
            out[pre + 'gm'] = gm16
            out[pre + 'extra'] = N.asarray(extra, dtype=float)
            out[pre + 'v'] = v
            out[pre + 'd'] = d
            out[pre + 't'] = t
            out[pre + 'hit_idx'] = hit_idx
            out[pre + 'hits'] = pts
            out[pre + 'normals'] = normals
            index.append('%s/frame%d' % (name, fi))
            ci += 1
    out['n_cases'] = N.int32(ci)
    out['names'] = N.array(index)
    print('geometry: %d cases' % ci)


# ---------------------------------------------------------------------------------------------------------
# 2. optics with replayed variates
# ---------------------------------------------------------------------------------------------------------
class FakeGeometry(object):
    """What an optics callable asks of its geometry manager."""
    def __init__(self, frame, normals, points):
        self._working_frame = frame
        self._n = normals
        self._p = points

    def get_normals(self):
        return self._n.copy()

    def get_intersection_points_global(self):
        return self._p

    def up(self):
        return self._working_frame[:3, 2]


def make_optics(ref, amd, out):
    rng = N.random.RandomState(77)
    H = 64
    frame = frame_of(rot([0.3, -1, 0.5], 1.1), [1., 2., 3.])
    up = frame[:3, 2]
    # normals scattered around `up` (and its opposite), unit incident directions with both orientations
    nrm = up[:, None] + 0.6 * rng.normal(size=(3, H))
    nrm /= N.sqrt(N.sum(nrm ** 2, axis=0))
    nrm[:, :3] = N.array([[0., 0., 1.], [0., 0., -1.], [1., 0., 0.]]).T      # the degenerate frames of rotation_to_z / z x n
    d = rng.normal(size=(3, H))
    d /= N.sqrt(N.sum(d ** 2, axis=0))
    flip = N.sum(d * nrm, axis=0) > 0
    nrm[:, flip] *= -1         # oriented normals oppose the ray, as get_normals() returns them
    pts = rng.uniform(-1, 1, size=(3, H))
    e = rng.uniform(0.5, 2., size=H)
    wl = rng.uniform(0.3e-6, 2.5e-6, size=H)
    sel = N.arange(H)
    oc = ref.optics_callables
    cases = []

    def run(name, opt_ref, opt_amd, ref_index=None, draws=None, wavelengths=None, lengths=None, spectra=None):
        kw = {}
        if ref_index is not None:
            kw['ref_index'] = ref_index
        if wavelengths is not None:
            kw['wavelengths'] = wavelengths
        if spectra is not None:           # polychromatic bundle: spectra (W,H) over wavelengths (W,H)
            kw['spectra'] = spectra.copy()
        # ray origins: `lengths` behind the hit points (1 when the optics does not look at the path)
        L = N.ones(H) if lengths is None else lengths
        bund = ref.ray_bundle.RayBundle(vertices=pts - d * L, directions=d.copy(), energy=e.copy(), **kw)
        geo = FakeGeometry(frame, nrm, pts)
        N.random.seed(len(cases) + 5)
        with N.errstate(all='ignore'):
            outg = opt_ref(geo, bund, sel)
        N.random.seed(len(cases) + 5)
        rec = draws() if draws else {}
        kind, params, extra = opt_amd._native()
        pre = 'o%d_' % len(cases)
        p8 = N.zeros(8)
        p8[:len(params)] = params
        out[pre + 'kind'] = N.int32(kind)
        out[pre + 'opt'] = p8
        out[pre + 'extra'] = N.asarray(extra, dtype=float)
        out[pre + 'ref_in'] = N.ones(H) if ref_index is None else ref_index
        if hasattr(opt_amd, '_materials'):      # the materials' own m() at the rays' wavelengths: what travels as trc_rays.mat
            out[pre + 'mat'] = N.array([m.m(wavelengths) for m in opt_amd._materials])
        if spectra is not None:
            out[pre + 'spec_in'], out[pre + 'spec_wl'] = spectra, wavelengths
            out[pre + 'out_spectra'] = outg.get_spectra()
        out[pre + 'path'] = L
        out[pre + 'out_dirs'] = outg.get_directions()
        out[pre + 'out_energy'] = outg.get_energy()
        out[pre + 'out_parents'] = N.asarray(outg.get_parents())
        out[pre + 'out_vertices'] = outg.get_vertices()
        if ref_index is not None:
            r = N.asarray(outg.get_ref_index())
            out[pre + 'out_ref'] = r if N.iscomplexobj(r) else N.asarray(r, dtype=float)
        for k, val in rec.items():
            out[pre + 'draw_' + k] = val
        cases.append(name)

    A = amd.optics_callables
    run('transparent', oc.Transparent(), A.Transparent())
    run('reflective', oc.Reflective(0.1), A.Reflective(0.1))
    run('one_sided_reflective', oc.OneSidedReflective(0.2), A.OneSidedReflective(0.2))
    run('real_reflective_bivar', oc.RealReflective(0.05, 3e-3, True), A.RealReflective(0.05, 3e-3, True),
        draws=lambda: dict(g0=N.random.normal(scale=3e-3, size=H), g1=N.random.normal(scale=3e-3, size=H)))
    run('real_reflective_radial', oc.RealReflective(0.05, 3e-3, False), A.RealReflective(0.05, 3e-3, False),
        draws=lambda: dict(g0=N.random.normal(scale=3e-3, size=H), phi=N.random.uniform(low=0., high=2. * N.pi, size=H)))
    run('real_reflective_sigma0', oc.RealReflective(0.05, 0., True), A.RealReflective(0.05, 0., True))
    run('one_sided_real_reflective', oc.OneSidedRealReflective(0.04, 1e-3, True), A.OneSidedRealReflective(0.04, 1e-3, True),
        draws=lambda: dict(g0=N.random.normal(scale=1e-3, size=H), g1=N.random.normal(scale=1e-3, size=H)))
    run('lambertian', oc.Lambertian(0.3), A.Lambertian(0.3),
        draws=lambda: dict(xi1=N.random.uniform(low=0., high=2. * N.pi, size=H), xi2=N.random.uniform(size=H)))
    run('lambertian_narrow', oc.Lambertian(0.3, 0.4), A.Lambertian(0.3, 0.4),
        draws=lambda: dict(xi1=N.random.uniform(low=0., high=2. * N.pi, size=H), xi2=N.random.uniform(size=H)))

    # Incidence Angle Modifier (IAM, :271-281) on the two siblings that run in the reference
    run('lambertian_iam', oc.Lambertian_IAM(0.3, 0.16), A.Lambertian_IAM(0.3, 0.16),
        draws=lambda: dict(xi1=N.random.uniform(low=0., high=2. * N.pi, size=H), xi2=N.random.uniform(size=H)))
    run('lambertian_iam_c2', oc.Lambertian_IAM(0.1, 0.3, 2), A.Lambertian_IAM(0.1, 0.3, 2),
        draws=lambda: dict(xi1=N.random.uniform(low=0., high=2. * N.pi, size=H), xi2=N.random.uniform(size=H)))
    run('real_reflective_iam', oc.RealReflective_IAM(0.05, 0.2, 2e-3, True), A.RealReflective_IAM(0.05, 0.2, 2e-3, True),
        draws=lambda: dict(g0=N.random.normal(scale=2e-3, size=H), g1=N.random.normal(scale=2e-3, size=H)))

    def ls_draws():
        u = N.random.rand(H)
        k = int(N.sum(~(u < 0.4)))
        return dict(u=u, xi1=N.random.uniform(low=0., high=2. * N.pi, size=k), xi2=N.random.uniform(size=k))
    run('lambertian_specular', oc.LambertianSpecular(0.1, 0.4), A.LambertianSpecular(0.1, 0.4), draws=ls_draws)
    run('lambertian_specular_iam', oc.LambertianSpecular_IAM(0.3, 0.4, 0.16), A.LambertianSpecular_IAM(0.3, 0.4, 0.16), draws=ls_draws)
    n_in = N.where(N.arange(H) % 2 == 0, 1.0, 1.5)
    run('refractive_split', oc.RefractiveHomogenous(1.0, 1.5, single_ray=False), A.RefractiveHomogenous(1.0, 1.5, single_ray=False),
        ref_index=n_in)
    run('refractive_single', oc.RefractiveHomogenous(1.0, 1.5, single_ray=True), A.RefractiveHomogenous(1.0, 1.5, single_ray=True),
        ref_index=N.ones(H) * 1.5, draws=lambda: dict(u=N.random.uniform(size=H)))
    run('refractive_split_sigma', oc.RefractiveHomogenous(1.0, 1.33, single_ray=False, sigma=2e-3),
        A.RefractiveHomogenous(1.0, 1.33, single_ray=False, sigma=2e-3), ref_index=N.ones(H),
        draws=lambda: dict(g0=N.random.normal(scale=2e-3, size=H), phi=N.random.uniform(low=0., high=2. * N.pi, size=H)))
    # attenuating media (Absorbant.attenuate): path lengths between 0.2 and 3 (the coefficient of LambertianAbsorbant as a list:
    # the reference takes len() of it, :883)
    Lr = rng.uniform(0.2, 3., size=H)
    run('lambertian_absorbant', oc.LambertianAbsorbant(0.3, [0.7], 1.2), A.LambertianAbsorbant(0.3, [0.7], 1.2), lengths=Lr,
        draws=lambda: dict(xi1=N.random.uniform(low=0., high=2. * N.pi, size=H), xi2=N.random.uniform(size=H)))
    run('lambertian_absorbant_scaled', oc.LambertianAbsorbant(0.1, [0.4], scaling=2.5), A.LambertianAbsorbant(0.1, [0.4], scaling=2.5), lengths=Lr,
        draws=lambda: dict(xi1=N.random.uniform(low=0., high=2. * N.pi, size=H), xi2=N.random.uniform(size=H)))
    run('refractive_transmissive_split', oc.RefractiveTransmissiveHomogenous(1.0, 1.5, [0.05, 0.9], single_ray=False),
        A.RefractiveTransmissiveHomogenous(1.0, 1.5, [0.05, 0.9], single_ray=False), ref_index=n_in, lengths=Lr)
    run('refractive_transmissive_one_coefficient', oc.RefractiveTransmissiveHomogenous(1.0, 1.5, [0.6], single_ray=False, scaling=0.5),
        A.RefractiveTransmissiveHomogenous(1.0, 1.5, [0.6], single_ray=False, scaling=0.5), ref_index=n_in, lengths=Lr)
    lam = N.linspace(0.2e-6, 3e-6, 9)
    ab = N.array([0.1, 0.2, 0.15, 0.4, 0.9, 0.5, 0.3, 0.2, 0.25])
    run('reflective_spectral', oc.Reflective_spectral(ab, lam), A.Reflective_spectral(ab, lam), wavelengths=wl)

    ths = N.linspace(0., N.pi / 2., 7)
    abth = N.array([0.9, 0.88, 0.85, 0.8, 0.7, 0.5, 0.1])
    run('lambertian_directional', oc.Lambertian_directional_axisymmetric_piecewise(ths, abth), A.Lambertian_directional_axisymmetric_piecewise(ths, abth),
        draws=lambda: dict(xi1=N.random.uniform(low=0., high=2. * N.pi, size=H), xi2=N.random.uniform(size=H)))
    spth = N.array([0.05, 0.1, 0.2, 0.35, 0.5, 0.7, 0.95])

    def spec_draws(prob):
        def f():
            u = N.random.rand(H)
            k = int(N.sum(~(u < prob())))
            return dict(u=u, xi1=N.random.uniform(low=0., high=2. * N.pi, size=k), xi2=N.random.uniform(size=k))
        return f
    th_in = N.arccos(N.abs(N.sum(d * nrm, axis=0)))
    run('lambertian_specular_directional', oc.LambertianSpecular_directional_axisymmetric_piecewise(ths, abth, 0.35),
        A.LambertianSpecular_directional_axisymmetric_piecewise(ths, abth, 0.35), draws=spec_draws(lambda: 0.35))
    run('lambertian_piecewise_specular_directional', oc.Lambertian_piecewise_Specular_directional_axisymmetric_piecewise(ths, abth, spth),
        A.Lambertian_piecewise_Specular_directional_axisymmetric_piecewise(ths, abth, spth),
        draws=spec_draws(lambda: N.interp(th_in, ths, spth)))
    wls = N.linspace(0.25e-6, 2.6e-6, 5)
    grid = 0.2 + 0.7 * N.outer(N.cos(ths) ** 0.5, 1. / (1. + (wls * 1e6 - 1.) ** 2))
    run('lambertian_directional_spectral', oc.Lambertian_directional_axisymmetric_piecewise_spectral(ths, grid, wls),
        A.Lambertian_directional_axisymmetric_piecewise_spectral(ths, grid, wls), wavelengths=wl,
        draws=lambda: dict(xi1=N.random.uniform(low=0., high=2. * N.pi, size=H), xi2=N.random.uniform(size=H)))
    mlam = N.linspace(0.2e-6, 3e-6, 8)
    mn = N.array([0.1, 0.13, 0.2, 0.4, 0.9, 1.5, 2.4, 3.6])
    mk = N.array([2.0, 3.5, 5.0, 7.0, 9.5, 13., 18., 24.])
    mat = A.TabulatedMaterial(mlam, mn, mk)
    run('fresnel_conductor', oc.FresnelConductorHomogenous(1., mat), A.FresnelConductorHomogenous(1., mat), wavelengths=wl)

    # Refractive / RefractiveAbsorbant between tabulated materials (complex indices; optics_callables.py:726-858, :908-944)
    tl = N.linspace(0.2e-6, 3e-6, 6)
    air = A.TabulatedMaterial(tl, N.ones(6), N.zeros(6))
    glass = A.TabulatedMaterial(tl, [1.56, 1.53, 1.51, 1.50, 1.49, 1.47], [4e-8, 2e-8, 1e-8, 6e-8, 3e-7, 9e-7])
    m_in = N.where(N.arange(H) % 2 == 0, air.m(wl), glass.m(wl))
    m_in[5] = 1.2 + 0.j                                      # a ray in neither medium enters material_1 (:750-751)
    run('material_split', oc.Refractive(air, glass, single_ray=False), A.Refractive(air, glass, single_ray=False),
        ref_index=m_in, wavelengths=wl)
    run('material_single', oc.Refractive(air, glass, single_ray=True), A.Refractive(air, glass, single_ray=True),
        ref_index=glass.m(wl), wavelengths=wl, draws=lambda: dict(u=N.random.uniform(size=H)))
    run('material_split_sigma', oc.Refractive(air, glass, single_ray=False, sigma=2e-3), A.Refractive(air, glass, single_ray=False, sigma=2e-3),
        ref_index=m_in, wavelengths=wl,
        draws=lambda: dict(g0=N.random.normal(scale=2e-3, size=H), phi=N.random.uniform(low=0., high=2. * N.pi, size=H)))
    run('material_absorbant_split', oc.RefractiveAbsorbant(air, glass, single_ray=False, attenuation_coefficient_1=1., attenuation_coefficient_2=1.),
        A.RefractiveAbsorbant(air, glass, single_ray=False, attenuation_coefficient_1=1., attenuation_coefficient_2=1.),
        ref_index=m_in, wavelengths=wl, lengths=Lr)
    run('material_absorbant_scaled', oc.RefractiveAbsorbant(air, glass, single_ray=False, attenuation_coefficient_1=1., scaling=0.3),
        A.RefractiveAbsorbant(air, glass, single_ray=False, attenuation_coefficient_1=1., scaling=0.3),
        ref_index=m_in, wavelengths=wl, lengths=Lr)

    # polychromatic bundles (`spectra` over `wavelengths`, both (W,H)): the wall that integrates them (:393-425) and the classes
    # that scale them
    W = 7
    swl = N.sort(rng.uniform(0.3e-6, 2.5e-6, size=(W, H)), axis=0)
    spec = rng.uniform(0.2, 3., size=(W, H)) * 1e6
    lamb = lambda: dict(xi1=N.random.uniform(low=0., high=2. * N.pi, size=H), xi2=N.random.uniform(size=H))
    run('polychromatic_wall', oc.Lambertian_directional_axisymmetric_piecewise_Polychromatic(ths, grid, wls),
        A.Lambertian_directional_axisymmetric_piecewise_Polychromatic(ths, grid, wls), wavelengths=swl, spectra=spec, draws=lamb)
    run('poly_transparent', oc.Transparent(), A.Transparent(), wavelengths=swl, spectra=spec)
    run('poly_reflective', oc.Reflective(0.1), A.Reflective(0.1), wavelengths=swl, spectra=spec)
    run('poly_one_sided_reflective', oc.OneSidedReflective(0.2), A.OneSidedReflective(0.2), wavelengths=swl, spectra=spec)
    run('poly_real_reflective', oc.RealReflective(0.05, 0., True), A.RealReflective(0.05, 0., True), wavelengths=swl, spectra=spec)
    run('poly_lambertian', oc.Lambertian(0.3), A.Lambertian(0.3), wavelengths=swl, spectra=spec, draws=lamb)
    run('poly_lambertian_absorbant', oc.LambertianAbsorbant(0.3, [0.7], 1.2), A.LambertianAbsorbant(0.3, [0.7], 1.2), lengths=Lr,
        wavelengths=swl, spectra=spec, draws=lamb)
    run('poly_lambertian_directional', oc.Lambertian_directional_axisymmetric_piecewise(ths, abth), A.Lambertian_directional_axisymmetric_piecewise(ths, abth),
        wavelengths=swl, spectra=spec, draws=lamb)
    run('poly_lambertian_specular', oc.LambertianSpecular(0.1, 0.4), A.LambertianSpecular(0.1, 0.4), wavelengths=swl, spectra=spec, draws=ls_draws)
    run('poly_refractive_split', oc.RefractiveHomogenous(1.0, 1.5, single_ray=False), A.RefractiveHomogenous(1.0, 1.5, single_ray=False),
        ref_index=n_in, wavelengths=swl, spectra=spec)

    # periodic boundary (:690-723): stub of energy 0 at the hit point, the ray itself one period along the oriented normal;
    # (last in the list: the cases before keep their numbers and their seeds)
    run('periodic_boundary', oc.PeriodicBoundary(0.7), A.PeriodicBoundary(0.7))
    run('poly_periodic_boundary', oc.PeriodicBoundary(1.3), A.PeriodicBoundary(1.3), wavelengths=swl, spectra=spec)

    # O1: optics.fresnel_to_attenuating on a grid of incidence angles x complex indices (optics.py:63-81)
    th = N.tile(N.linspace(0., N.pi / 2. - 1e-3, 40), 6)
    m2 = N.repeat(N.array([1.5 + 0.01j, 0.2 + 3.4j, 2.7 + 2.9j, 1.0 + 0.j, 0.05 + 6.j, 3.9 + 0.2j]), 40)
    rp, rs, t2 = ref.optics.fresnel_to_attenuating(1.33, m2, th)
    out['fta_theta1'], out['fta_m_re'], out['fta_m_im'], out['fta_n1'] = th, m2.real, m2.imag, N.float64(1.33)
    out['fta_rp'], out['fta_rs'], out['fta_theta2'] = rp, rs, t2

    out['n_cases'] = N.int32(len(cases))
    out['names'] = N.array(cases)
    out['frame'] = frame
    out['normals'] = nrm
    out['dirs'] = d
    out['points'] = pts
    out['energy'] = e
    out['wavelengths'] = wl
    print('optics: %d cases' % len(cases))


# ---------------------------------------------------------------------------------------------------------
# 3. sources with replayed variates
# ---------------------------------------------------------------------------------------------------------
def desc_arrays(bundle):
    d = bundle.source_args()[0]
    return dict(kind=N.int32(d.kind), center=N.array(list(d.center)), rot_pos=N.array(list(d.rot_pos)).reshape(3, 3),
                rot_dir=N.array(list(d.rot_dir)).reshape(3, 3), p=N.array(list(d.p)), energy=N.float64(d.energy),
                buie=N.array(list(d.buie)))


def make_sources(ref, amd, out):
    n = 1500
    cases = []
    S = ref.sources
    A = amd.sources

    def store(name, bund_ref, bund_amd, uniforms):
        pre = 's%d_' % len(cases)
        for k, val in desc_arrays(bund_amd).items():
            out[pre + 'desc_' + k] = val
        for i, u in enumerate(uniforms):
            out[pre + 'u%d' % i] = u
        out[pre + 'vertices'] = bund_ref.get_vertices()
        out[pre + 'directions'] = bund_ref.get_directions()
        out[pre + 'energy'] = bund_ref.get_energy()
        cases.append(name)

    sun = N.array([0., N.sin(0.6), N.cos(0.6)])
    centre = N.c_[[10., 250., 200.]]
    for name, csr, pre_csr in (('buie_csr0.01_raw', 0.01, False), ('buie_csr0.05', 0.05, True), ('buie_csr0.3', 0.3, True),
                               ('buie_csr0', 0., True)):
        N.random.seed(11)
        b = S.buie_sunshape(n, centre, -sun, 163., csr, flux=1000., pre_process_CSR=pre_csr)
        N.random.seed(11)
        xv1 = N.random.uniform(size=n)
        phiv = N.random.uniform(high=2. * N.pi, size=n)
        R = N.random.uniform(size=n)
        xi = N.random.uniform(high=2. * N.pi, size=n)
        store(name, b, A.buie_sunshape(n, centre, -sun, 163., csr, flux=1000., pre_process_CSR=pre_csr, seed=1),
              (xv1, phiv / (2. * N.pi), R, xi / (2. * N.pi)))
    # straight down (degenerate rotation_to_z frame); NB the reference's buie_sunshape cannot take an array
    # rays_direction (`== None` on an array, sources.py:441) -- the oblique case is covered by rect_buie below
    N.random.seed(12)
    b = S.buie_sunshape(n, N.c_[[0., 0., 6.]], N.r_[0., 0., -1.], 2.5, 0.05, flux=1000.)
    N.random.seed(12)
    xv1 = N.random.uniform(size=n); phiv = N.random.uniform(high=2. * N.pi, size=n); R = N.random.uniform(size=n); xi = N.random.uniform(high=2. * N.pi, size=n)
    store('buie_down', b, A.buie_sunshape(n, N.c_[[0., 0., 6.]], N.r_[0., 0., -1.], 2.5, 0.05, flux=1000., seed=1),
          (xv1, phiv / (2. * N.pi), R, xi / (2. * N.pi)))
    N.random.seed(17)
    rd = N.r_[0.1, 0., -0.99498743710662]
    b = S.rect_buie_sunshape(n, N.c_[[0., 0., 6.]], N.r_[0., 0., -1.], 3., 2., 0.02, flux=1000., rays_direction=rd)
    N.random.seed(17)
    ux = N.random.uniform(size=n); uy = N.random.uniform(size=n); R = N.random.uniform(size=n); xi = N.random.uniform(high=2. * N.pi, size=n)
    store('rect_buie_oblique', b, A.rect_buie_sunshape(n, N.c_[[0., 0., 6.]], N.r_[0., 0., -1.], 3., 2., 0.02, flux=1000., rays_direction=rd, seed=1),
          (ux, uy, R, xi / (2. * N.pi)))
    N.random.seed(13)
    b = S.rect_buie_sunshape(n, centre, -sun, 30., 20., 0.1, flux=900.)
    N.random.seed(13)
    ux = N.random.uniform(size=n); uy = N.random.uniform(size=n); R = N.random.uniform(size=n); xi = N.random.uniform(high=2. * N.pi, size=n)
    store('rect_buie', b, A.rect_buie_sunshape(n, centre, -sun, 30., 20., 0.1, flux=900., seed=1), (ux, uy, R, xi / (2. * N.pi)))
    for name, direction in (('rect_bundle', N.r_[-0.15, 0., -1.] / N.sqrt(1.0225)), ('rect_bundle_down', N.r_[0., 0., -1.])):
        N.random.seed(14)
        b = S.rect_bundle(n, N.c_[[1., 2., 5.]], direction, 2., 3., 4.65e-3, flux=1000.)
        N.random.seed(14)
        xi1 = N.random.uniform(low=0., high=2. * N.pi, size=n); xi2 = N.random.uniform(size=n)
        xs = N.random.uniform(low=-1., high=1., size=n); ys = N.random.uniform(low=-1.5, high=1.5, size=n)
        store(name, b, A.rect_bundle(n, N.c_[[1., 2., 5.]], direction, 2., 3., 4.65e-3, flux=1000., seed=1),
              (xi1 / (2. * N.pi), xi2, (xs + 1.) / 2., (ys + 1.5) / 3.))
    N.random.seed(15)
    b = S.disk_bundle(n, N.c_[[0., 1., 4.]], N.r_[0., N.sin(0.3), -N.cos(0.3)], 1.5, 0.02, flux=800., radius_in=0.3)
    N.random.seed(15)
    xi1 = N.random.uniform(low=0., high=2. * N.pi, size=n); xi2 = N.random.uniform(size=n)
    r1 = N.random.uniform(size=n); th = N.random.uniform(low=0., high=2. * N.pi, size=n)
    store('disk_bundle', b, A.disk_bundle(n, N.c_[[0., 1., 4.]], N.r_[0., N.sin(0.3), -N.cos(0.3)], 1.5, 0.02, flux=800., radius_in=0.3, seed=1),
          (xi1 / (2. * N.pi), xi2, r1, th / (2. * N.pi)))
    N.random.seed(16)
    b = S.disk_bundle(n, N.c_[[0., 0., 0.]], N.r_[0., 0., 1.], 1., N.pi / 2.)
    N.random.seed(16)
    xi1 = N.random.uniform(low=0., high=2. * N.pi, size=n); xi2 = N.random.uniform(size=n)
    r1 = N.random.uniform(size=n); th = N.random.uniform(low=0., high=2. * N.pi, size=n)
    store('disk_bundle_lambertian_noflux', b, A.disk_bundle(n, N.c_[[0., 0., 0.]], N.r_[0., 0., 1.], 1., N.pi / 2., seed=1),
          (xi1 / (2. * N.pi), xi2, r1, th / (2. * N.pi)))
    N.random.seed(18)
    TA, TB, TC = N.r_[0., 0., 1.], N.r_[2., 0.5, 1.2], N.r_[0.3, 1.5, 0.7]
    b = S.triangular_bundle(n, TA, TB, TC, ang_range=0.3, flux=500.)
    N.random.seed(18)
    r1 = N.random.uniform(size=n); r2 = N.random.uniform(size=n)
    xi1 = N.random.uniform(low=0., high=2. * N.pi, size=n); xi2 = N.random.uniform(size=n)
    store('triangular_bundle', b, A.triangular_bundle(n, TA, TB, TC, ang_range=0.3, flux=500., seed=1), (r1, r2, xi1 / (2. * N.pi), xi2))
    N.random.seed(19)
    sd = N.r_[0., N.sin(0.2), -N.cos(0.2)]; rd = N.r_[N.sin(0.1), 0., -N.cos(0.1)]
    b = S.oblique_solar_rect_bundle(n, N.c_[[1., 0., 8.]], sd, rd, 3., 2., 4.65e-3, flux=1000.)
    N.random.seed(19)
    xi1 = N.random.uniform(low=0., high=2. * N.pi, size=n); xi2 = N.random.uniform(size=n)
    xs = N.random.uniform(low=-1.5, high=1.5, size=n); ys = N.random.uniform(low=-1., high=1., size=n)
    store('oblique_solar_rect_bundle', b, A.oblique_solar_rect_bundle(n, N.c_[[1., 0., 8.]], sd, rd, 3., 2., 4.65e-3, flux=1000., seed=1),
          (xi1 / (2. * N.pi), xi2, (xs + 1.5) / 3., (ys + 1.) / 2.))
    # S4 view-factor emitters (sources.py:644-769): cylinder draws zs, phi_s, dir phi, dir R; frustum dir phi, dir R, R, phi_s
    for name, kw in (('vf_cylinder_in', dict(rays_in=True)),
                     ('vf_cylinder_out_wedge_flux', dict(rays_in=False, angular_span=[0.3, 2.1], flux=700., ang_range=1.2))):
        dvec = N.r_[0.2, -0.3, N.sqrt(1. - 0.13)]
        N.random.seed(20)
        b = S.vf_cylinder_bundle(n, 0.8, 1.7, N.c_[[0.5, -1., 2.]], dvec, **kw)
        N.random.seed(20)
        span = kw.get('angular_span', [0., 2. * N.pi])
        zs = N.random.uniform(size=n); ph = N.random.uniform(low=span[0], high=span[1], size=n)
        xi1 = N.random.uniform(low=0., high=2. * N.pi, size=n); xi2 = N.random.uniform(size=n)
        store(name, b, A.vf_cylinder_bundle(n, 0.8, 1.7, N.c_[[0.5, -1., 2.]], dvec, seed=1, **kw),
              (zs, (ph - span[0]) / (span[1] - span[0]), xi1 / (2. * N.pi), xi2))
    for name, r0, r1, kw in (('vf_frustum_widening', 0.5, 1.2, dict(rays_in=True)),
                             ('vf_frustum_narrowing_out_flux', 1.1, 0.4, dict(rays_in=False, angular_span=[1., 4.], flux=300., angular_range=1.)),
                             ('vf_frustum_cone_tip', 1., 0., dict(rays_in=True))):
        dvec = N.r_[0., 0., 1.] if name.endswith('tip') else N.r_[-0.1, 0.4, N.sqrt(1. - 0.17)]
        N.random.seed(21)
        b = S.vf_frustum_bundle(n, r0, r1, 0.9, N.c_[[0., 0.3, 1.]], dvec, **kw)
        N.random.seed(21)
        span = kw.get('angular_span', [0., 2. * N.pi])
        xi1 = N.random.uniform(low=0., high=2. * N.pi, size=n); xi2 = N.random.uniform(size=n)
        R = N.random.uniform(size=n); ph = N.random.uniform(low=span[0], high=span[1], size=n)
        store(name, b, A.vf_frustum_bundle(n, r0, r1, 0.9, N.c_[[0., 0.3, 1.]], dvec, seed=1, **kw),
              (xi1 / (2. * N.pi), xi2, R, (ph - span[0]) / (span[1] - span[0])))
    out['n_cases'] = N.int32(len(cases))
    out['names'] = N.array(cases)
    print('sources: %d cases' % len(cases))


# ---------------------------------------------------------------------------------------------------------
# 4. deterministic end-to-end scenes traced by the reference engine
# ---------------------------------------------------------------------------------------------------------
def scene_two_planes(T):
    """tests/test_tracer_engine.py TestTraceProtocol-like: two perpendicular absorbing mirrors"""
    s1 = T.surface.Surface(T.flat_surface.FlatGeometryManager(), T.optics_callables.Reflective(0.1))
    s2 = T.surface.Surface(T.flat_surface.FlatGeometryManager(), T.optics_callables.Reflective(0.2),
                           rotation=T.spatial_geometry.general_axis_rotation(N.r_[1., 0., 0.], N.pi / 2.))
    o1 = T.object.AssembledObject(surfs=[s1])
    o2 = T.object.AssembledObject(surfs=[s2], location=N.r_[0., 1., 0.])
    return T.assembly.Assembly(objects=[o1, o2])


def scene_mixed(T):
    """plates, a dish, a hemisphere, a cylinder and a cone around the origin, mirrors of various kinds (sigma=0)"""
    O = T.optics_callables
    sg = T.spatial_geometry
    objs = []
    objs.append(T.object.AssembledObject(surfs=[T.surface.Surface(T.flat_surface.RectPlateGM(3., 3.), O.Reflective(0.05))],
                                         transform=sg.translate(0, 0, -2.)))
    objs.append(T.object.AssembledObject(surfs=[T.surface.Surface(T.paraboloid.ParabolicDishGM(3., 2.), O.RealReflective(0.1, 0., True))],
                                         transform=N.dot(sg.translate(0, 0, 3.), sg.rotx(N.pi))))
    objs.append(T.object.AssembledObject(surfs=[T.surface.Surface(T.sphere_surface.HemisphereGM(1.5), O.Reflective(0.3))],
                                         transform=N.dot(sg.translate(3., 0, 0.), sg.roty(N.pi / 2.))))
    objs.append(T.object.AssembledObject(surfs=[T.surface.Surface(T.cylinder.FiniteCylinder(1., 2.), O.OneSidedReflective(0.2))],
                                         transform=N.dot(sg.translate(-3., 0.5, 0.), sg.rotx(0.4))))
    objs.append(T.object.AssembledObject(surfs=[T.surface.Surface(T.cone.ConicalFrustum(0., 0.5, 1., 1.2), O.Reflective(0.5))],
                                         transform=sg.translate(0., 3., -0.5)))
    objs.append(T.object.AssembledObject(surfs=[T.surface.Surface(T.flat_surface.RoundPlateGM(1.2), O.Reflective(0.97))],
                                         transform=N.dot(sg.translate(0., -3., 0.), sg.rotx(-N.pi / 2.))))
    sub = T.assembly.Assembly(objects=objs[3:], location=N.r_[0.1, 0.2, 0.3], rotation=sg.rotz(0.3)[:3, :3])
    return T.assembly.Assembly(objects=objs[:3], subassemblies=[sub])


def scene_lens(T):
    """plano-convex lens: paraboloidal cap over a round plate, split-mode refraction, mirror behind.
    (A HemisphereGM cap is avoided on purpose: with rays crossing it in both directions in one bundle the
    reference's root assignment gets shuffled between rays, see make_geometry.)"""
    O = T.optics_callables
    sg = T.spatial_geometry
    h = (2. / (2. * N.sqrt(1.5))) ** 2
    front = T.surface.Surface(T.paraboloid.ParabolicDishGM(4., 1.5), O.RefractiveHomogenous(1., 1.5, single_ray=False),
                              location=N.r_[0., 0., h], rotation=sg.rotx(N.pi)[:3, :3])
    back = T.surface.Surface(T.flat_surface.RoundPlateGM(2.), O.RefractiveHomogenous(1., 1.5, single_ray=False))
    lens = T.object.AssembledObject(surfs=[front, back], transform=sg.translate(0, 0, 1.))
    mirror = T.object.AssembledObject(surfs=[T.surface.Surface(T.flat_surface.RectPlateGM(6., 6.), O.Reflective(0.9))],
                                      transform=sg.translate(0, 0, -4.))
    return T.assembly.Assembly(objects=[lens, mirror])


def scene_nsttf(T, n_hel, with_bounds=True):
    pos = N.loadtxt(os.path.join(REFERENCE, 'examples', 'sandia_hstat_coordinates.csv'), delimiter=',')
    pos[:, 1] -= 4.
    if n_hel:
        pos = pos[:n_hel]
    hf = importlib.import_module(T.pkg + '.models.heliostat_field')
    osm = importlib.import_module(T.pkg + '.models.one_sided_mirror')
    field = hf.HeliostatField(pos, 6.1, 6.1, absorptivity=0.04, sigma=0., bi_var=True, MCRT_option='fast')
    aim = N.tile(N.array([0., 0., 60.]), (pos.shape[0], 1))
    field.track_sun(0., 35.05 * N.pi / 180., aim_points=aim)
    rec = osm.one_sided_receiver(11., 11.)
    rec.set_transform(N.dot(T.spatial_geometry.translate(0., 0., 60.), T.spatial_geometry.rotx(-N.pi / 2.)))
    return T.assembly.Assembly(objects=[rec], subassemblies=[field])


def tree_arrays(engine, out, pre):
    tree = engine.tree
    out[pre + 'n_levels'] = N.int32(tree.num_bunds())
    for k in range(tree.num_bunds()):
        b = tree[k]
        out[pre + 'L%d_vertices' % k] = b.get_vertices()
        out[pre + 'L%d_directions' % k] = b.get_directions()
        out[pre + 'L%d_energy' % k] = b.get_energy()
        if k > 0:
            out[pre + 'L%d_parents' % k] = N.asarray(b.get_parents(), dtype=N.int64)
        if b.has_property('ref_index') and hasattr(b, '_ref_index') and b._ref_index is not None and k > 0:
            try:
                out[pre + 'L%d_ref_index' % k] = N.asarray(b.get_ref_index(), dtype=float)
            except Exception:
                pass


def make_engine(ref, amd, out):
    from tracer_amd.scene import compile_scene, scene_arrays
    rng = N.random.RandomState(5)
    cases = []

    def run(name, builder, v, d, e, reps, min_energy, accel=False, ref_index=None, **bkw):
        asm_ref = builder(ref, **bkw)
        asm_amd = builder(amd, **bkw)
        pre = 'e%d_' % len(cases)
        for k, val in scene_arrays(compile_scene(asm_amd)).items():
            out[pre + 'scene_' + k] = val
        kw = {}
        if ref_index is not None:
            kw['ref_index'] = ref_index
        bund = ref.ray_bundle.RayBundle(vertices=v.copy(), directions=d.copy(), energy=e.copy(), **kw)
        eng = ref.tracer_engine.TracerEngine(asm_ref)
        with N.errstate(all='ignore'):
            lv, ld = eng.ray_tracer(bund, reps=reps, min_energy=min_energy, tree=True, accel=accel)
        out[pre + 'v'] = v
        out[pre + 'd'] = d
        out[pre + 'e'] = e
        if ref_index is not None:
            out[pre + 'ref_index'] = ref_index
        out[pre + 'reps'] = N.int32(reps)
        out[pre + 'min_energy'] = N.float64(min_energy)
        out[pre + 'accel'] = N.int32(1 if accel else 0)
        out[pre + 'last_vertices'] = lv
        out[pre + 'last_directions'] = ld
        tree_arrays(eng, out, pre)
        # accountants of the last surface, if any
        surfs = asm_ref.get_surfaces()
        for si, s in enumerate(surfs):
            o = s.get_optics_manager()
            if hasattr(o, 'get_all_hits'):
                for ai, arr in enumerate(o.get_all_hits()):
                    out[pre + 'acc_s%d_%d' % (si, ai)] = N.asarray(arr)
        cases.append(name)
        print('  engine case %s: levels %s' % (name, [eng.tree[k].get_num_rays() for k in range(eng.tree.num_bunds())]))

    # (a) two planes, the classic four rays + extras
    n = 12
    d = N.tile(N.r_[0., 1. / N.sqrt(2.), -1. / N.sqrt(2.)][:, None], (1, n))
    v = N.vstack((rng.uniform(-1, 1, n), rng.uniform(-2., 0.5, n), N.ones(n) * 1.))
    run('two_planes', scene_two_planes, v, d, N.ones(n), reps=6, min_energy=0.05)
    # (b) mixed scene, many bounces, culling in play
    n = 600
    v = rng.uniform(-0.5, 0.5, size=(3, n))
    d = rng.normal(size=(3, n))
    d /= N.sqrt(N.sum(d ** 2, axis=0))
    run('mixed', scene_mixed, v, d, rng.uniform(0.5, 1.5, n), reps=12, min_energy=0.05)
    run('mixed_reps3', scene_mixed, v, d, rng.uniform(0.5, 1.5, n), reps=3, min_energy=1e-10)
    # (c) lens with ray splitting
    n = 80
    v = N.vstack((rng.uniform(-1.2, 1.2, n), rng.uniform(-1.2, 1.2, n), N.ones(n) * 6.))
    d = N.vstack((rng.normal(scale=0.03, size=n), rng.normal(scale=0.03, size=n), -N.ones(n)))
    d /= N.sqrt(N.sum(d ** 2, axis=0))
    run('lens_split', scene_lens, v, d, N.ones(n), reps=5, min_energy=1e-3, ref_index=N.ones(n))
    # (d) NSTTF subset with deterministic mirrors, brute force and Kd-tree
    N.random.seed(3)
    sun = N.r_[0., N.sin(35.05 * N.pi / 180.), N.cos(35.05 * N.pi / 180.)]
    src = ref.sources.buie_sunshape(3000, N.vstack(300. * sun + N.r_[0., 80., 0.]), -sun, 60., 0.01, flux=1000., pre_process_CSR=False)
    v, d, e = src.get_vertices(), src.get_directions(), src.get_energy()
    run('nsttf30', scene_nsttf, v, d, e, reps=100, min_energy=1e-10, n_hel=30)
    run('nsttf30_accel', scene_nsttf, v, d, e, reps=100, min_energy=1e-10, accel=True, n_hel=30)
    out['n_cases'] = N.int32(len(cases))
    out['names'] = N.array(cases)
    print('engine: %d cases' % len(cases))


# ---------------------------------------------------------------------------------------------------------
# 5. Kd-tree of the full NSTTF field, 6. accountant naming table
# ---------------------------------------------------------------------------------------------------------
def make_kdtree(ref, out):
    for tag, fast in (('', False), ('fast_', True)):
        asm = scene_nsttf(ref, None)
        S = len(asm.get_surfaces())
        kd = ref.accel_tree.KdTree(asm, 8 + 1.3 * N.log(S), fast=fast, min_leaf=1)
        n = len(kd.nodes)
        flag = N.array([nd.flag for nd in kd.nodes], dtype=N.int32)
        split = N.array([nd.split if nd.flag != 3 else 0. for nd in kd.nodes], dtype=float)
        child = N.array([nd.child if nd.flag != 3 else 0 for nd in kd.nodes], dtype=N.int32)
        leaf_off, leaf_cnt, leaf_surfs = [], [], []
        for nd in kd.nodes:
            if nd.flag == 3:
                s = sorted(set(int(x) for arr in nd.surfaces_idxs for x in N.ravel(arr)))
                leaf_off.append(len(leaf_surfs)); leaf_cnt.append(len(s)); leaf_surfs.extend(s)
            else:
                leaf_off.append(0); leaf_cnt.append(0)
        out[tag + 'flag'] = flag
        out[tag + 'split'] = split
        out[tag + 'child'] = child
        out[tag + 'leaf_off'] = N.array(leaf_off, dtype=N.int32)
        out[tag + 'leaf_cnt'] = N.array(leaf_cnt, dtype=N.int32)
        out[tag + 'leaf_surfs'] = N.array(leaf_surfs, dtype=N.int32)
        out[tag + 'always_relevant'] = N.asarray(kd.always_relevant, dtype=N.int32)
        out[tag + 'minpoint'] = N.ravel(kd.minpoint)
        out[tag + 'maxpoint'] = N.ravel(kd.maxpoint)
        print('kdtree%s: %d nodes, %d leaves' % (' (fast)' if fast else '', n, int((flag == 3).sum())))
        if not fast:
            # KdTree.traversal on its own (accel_tree.py:213-312): the relevancy matrix the reference returns for rays that come
            # down on the field like the sun's, leave the receiver towards it, start inside the root box, run along an axis (a zero
            # direction component: infinite inverse), or miss the box
            rng = N.random.RandomState(31)
            lo, hi = N.ravel(kd.minpoint), N.ravel(kd.maxpoint)
            cen, ext = 0.5 * (lo + hi), (hi - lo)
            k = 140
            sun = N.array([0., N.sin(0.61), N.cos(0.61)])
            v1 = cen[:, None] + 300. * sun[:, None] + N.vstack((rng.uniform(-1, 1, k) * ext[0], rng.uniform(-1, 1, k) * ext[1], N.zeros(k)))
            d1 = N.tile(-sun[:, None], (1, k)) + 4e-3 * rng.normal(size=(3, k))
            v2 = N.tile(N.c_[[0., 0., 60.]], (1, k)) + rng.uniform(-5, 5, size=(3, k))
            d2 = N.vstack((rng.uniform(lo[0], hi[0], k), rng.uniform(lo[1], hi[1], k), N.zeros(k))) - v2
            v3 = lo[:, None] + rng.uniform(0, 1, size=(3, k)) * ext[:, None]
            d3 = rng.normal(size=(3, k))
            v4 = lo[:, None] + rng.uniform(0, 1, size=(3, k)) * ext[:, None]
            d4 = N.zeros((3, k))
            d4[rng.randint(0, 3, k), N.arange(k)] = rng.choice([-1., 1.], k)
            v4[2, :k // 2] = hi[2] + 5.                 # above the field, along x or y: parallel to the box, outside
            v5 = cen[:, None] + 3. * ext.max() * rng.normal(size=(3, k))
            d5 = rng.normal(size=(3, k))
            v = N.hstack((v1, v2, v3, v4, v5))
            d = N.hstack((d1, d2, d3, d4, d5))
            d /= N.sqrt(N.sum(d ** 2, axis=0))
            b = ref.ray_bundle.RayBundle(vertices=v, directions=d, energy=N.ones(v.shape[1]))
            with N.errstate(all='ignore'):
                any_inter, rel = kd.traversal(b)
            out['trav_vertices'], out['trav_directions'] = v, d
            out['trav_any'] = N.int32(bool(any_inter))
            out['trav_relevancy_bits'] = N.packbits(rel, axis=1)
            out['trav_n_surf'] = N.int32(S)
            print('  traversal: %d rays, %d of them meet the root box, %.1f surfaces per such ray' %
                  (v.shape[1], int(rel[:-1].any(axis=0).sum()), rel.sum() / max(1., float(rel[:-1].any(axis=0).sum()))))


def make_accountant_names(ref):
    oc = ref.optics_callables
    table = {}
    for name in dir(oc):
        if not name.startswith('Reflective') or name.startswith('Reflective_'):
            continue
        cls = getattr(oc, name)
        if not isinstance(cls, type) or name == 'Reflective':
            continue
        inst = cls(0.1)
        table[name[len('Reflective'):]] = [type(a).__name__ for a in inst.accountants]
    return table


def make_mc(ref):
    """
    Monte-Carlo references: the reference engine itself on the benchmark scenes (restated as in SURVEY.md 8(d)),
    K seeds x 1e5 rays each; mean and standard error of the scene-level quantities the GPU runs are compared to.
    """
    import time
    out = {}
    osm = importlib.import_module('tracer.models.one_sided_mirror')
    hf = importlib.import_module('tracer.models.heliostat_field')
    # --- NSTTF, 218 heliostats, sigma 1 mrad, Buie CSR 0.01 ---
    pos = N.loadtxt(os.path.join(REFERENCE, 'examples', 'sandia_hstat_coordinates.csv'), delimiter=',')
    pos[:, 1] -= 4.
    field = hf.HeliostatField(pos, 6.1, 6.1, absorptivity=0.04, sigma=1e-3, bi_var=True, MCRT_option='fast')
    zen = 35.05 * N.pi / 180.
    field.track_sun(0., zen, aim_points=N.tile(N.array([0., 0., 60.]), (pos.shape[0], 1)))
    rec = osm.one_sided_receiver(11., 11.)
    rec.set_transform(N.dot(ref.spatial_geometry.translate(0., 0., 60.), ref.spatial_geometry.rotx(-N.pi / 2.)))
    plant = ref.assembly.Assembly(objects=[rec], subassemblies=[field])
    sun = hf.solar_vector(0., zen)
    x0, x1, y0, y1 = pos[:, 0].min(), pos[:, 0].max(), pos[:, 1].min(), pos[:, 1].max()
    centre = N.array([(x0 + x1) / 2., (y0 + y1) / 2., 0.])
    radius = 1.10 * N.sqrt(((x1 - x0) / 2.) ** 2 + ((y1 - y0) / 2.) ** 2)
    n = 100000
    P, F, flux = [], [], []
    edges = N.linspace(-5.5, 5.5, 51)
    t0 = time.time()
    for k in range(10):
        N.random.seed(1000 + k)
        plant.reset_all_optics()
        src = ref.sources.buie_sunshape(n, N.vstack(300. * sun + centre), -sun, radius, 0.01, flux=1000., pre_process_CSR=False)
        eng = ref.tracer_engine.TracerEngine(plant)
        eng.ray_tracer(src, reps=100, min_energy=1e-10, tree=True)
        en, pts = rec.get_surfaces()[0].get_optics_manager().get_all_hits()
        P.append(en.sum())
        F.append([eng.tree[1].get_num_rays() / float(n), len(en) / float(n)])
        loc = rec.get_surfaces()[0].global_to_local(pts)
        flux.append(N.histogram2d(loc[0], loc[1], bins=[edges, edges], weights=en)[0])
    print('  nsttf mc: %.1f s, receiver %.1f +- %.1f kW' % (time.time() - t0, N.mean(P) / 1e3, N.std(P, ddof=1) / N.sqrt(len(P)) / 1e3))
    out['nsttf_receiver_runs'] = N.array(P)
    out['nsttf_receiver_mean'] = N.mean(P)
    out['nsttf_receiver_se'] = N.std(P, ddof=1) / N.sqrt(len(P))
    out['nsttf_bounce_fractions_mean'] = N.mean(F, axis=0)
    out['nsttf_bounce_fractions_se'] = N.std(F, axis=0, ddof=1) / N.sqrt(len(F))
    out['nsttf_flux_mean'] = N.mean(flux, axis=0)
    out['nsttf_flux_se'] = N.std(flux, axis=0, ddof=1) / N.sqrt(len(flux))
    out['nsttf_rays_per_run'] = N.int64(n)
    # --- dish (config 2): D=5 f=3, RealReflective(0.06, 2 mrad radial), round receiver r=0.15, Buie CSR 0.05 ---
    O = ref.optics_callables
    dish_s = ref.surface.Surface(ref.paraboloid.ParabolicDishGM(5., 3.), O.RealReflective(0.06, 2e-3, bi_var=False))
    rec_s = ref.surface.Surface(ref.flat_surface.RoundPlateGM(0.15), O.LambertianReceiver(1.))
    asm = ref.assembly.Assembly(objects=[ref.object.AssembledObject(surfs=[dish_s]),
                                         ref.object.AssembledObject(surfs=[rec_s], transform=N.dot(ref.spatial_geometry.translate(0., 0., 3.), ref.spatial_geometry.rotx(N.pi)))])
    D = []
    t0 = time.time()
    for k in range(8):
        N.random.seed(2000 + k)
        asm.reset_all_optics()
        src = ref.sources.buie_sunshape(n, N.c_[[0., 0., 6.]], N.r_[0., 0., -1.], 2.5, 0.05, flux=1000.)
        eng = ref.tracer_engine.TracerEngine(asm)
        eng.ray_tracer(src, reps=10, min_energy=1e-10, tree=True)
        D.append(rec_s.get_optics_manager().get_all_hits()[0].sum())
    print('  dish mc: %.1f s, receiver %.2f +- %.2f W' % (time.time() - t0, N.mean(D), N.std(D, ddof=1) / N.sqrt(len(D))))
    out['dish_receiver_mean'] = N.mean(D)
    out['dish_receiver_se'] = N.std(D, ddof=1) / N.sqrt(len(D))
    out['dish_source_power'] = 1000. * N.pi * 2.5 ** 2
    # --- flat pair (config 1): 2x2 mirror RealReflective(0.05, 2 mrad bi-variate), 4x4 LambertianReceiver, pillbox rect source ---
    dvec = N.r_[-0.15, 0., -1.]
    dvec = dvec / N.sqrt(N.sum(dvec ** 2))
    mirror = ref.surface.Surface(ref.flat_surface.RectPlateGM(2., 2.), O.RealReflective(0.05, 2e-3, bi_var=True))
    outd = dvec - 2. * dvec[2] * N.r_[0., 0., 1.]
    recp = ref.surface.Surface(ref.flat_surface.RectPlateGM(4., 4.), O.LambertianReceiver(1.))
    asm = ref.assembly.Assembly(objects=[ref.object.AssembledObject(surfs=[mirror]),
                                         ref.object.AssembledObject(surfs=[recp], transform=N.dot(ref.spatial_geometry.translate(*(outd * 10. / outd[2])), ref.spatial_geometry.rotx(N.pi)))])
    Fp = []
    for k in range(8):
        N.random.seed(3000 + k)
        asm.reset_all_optics()
        src = ref.sources.rect_bundle(n, N.vstack(-dvec * 5.), dvec, 2., 2., 4.65e-3, flux=1000.)
        eng = ref.tracer_engine.TracerEngine(asm)
        eng.ray_tracer(src, reps=10, min_energy=1e-10, tree=True)
        en, pts = recp.get_optics_manager().get_all_hits()
        loc = recp.global_to_local(pts)
        Fp.append([en.sum(), N.sqrt(N.mean(loc[0] ** 2)), N.sqrt(N.mean(loc[1] ** 2))])
    out['flat_receiver_mean'] = N.mean(Fp, axis=0)
    out['flat_receiver_se'] = N.std(Fp, axis=0, ddof=1) / N.sqrt(len(Fp))
    print('  flat mc: receiver W, rms x, rms y =', out['flat_receiver_mean'], '+-', out['flat_receiver_se'])
    N.savez_compressed(os.path.join(HERE, 'mc_reference.npz'), **out)


def make_mc_minidish(ref):
    """
    The scene of examples/test_case.py:29-52 (models/tau_minidish.py on homogenized_local_receiver.py: tilted dish, homogenizer
    duct, one-sided receiver) traced by the reference itself, 10 seeds x 1e5 rays of `disk_bundle` (the example's
    `solar_disk_bundle` under its current name): power on the receiver plate and absorbed by each duct wall and by the dish,
    the 20 x 20 map of examples/test_case.py:60, the size of every level of the ray tree.  -> mc_minidish.npz
    """
    import math
    import time
    md = importlib.import_module('tracer.models.tau_minidish')
    focus, h_depth, side, n = 6.25, 0.7, 0.4, 100000
    x = -1 / math.sqrt(2)
    P, W, H, L = [], [], [], []
    t0 = time.time()
    for k in range(10):
        N.random.seed(4000 + k)
        dish = md.MiniDish(5., focus, 0.9, focus + h_depth, side, h_depth, 0.9)
        dish.set_transform(ref.spatial_geometry.rotx(-N.pi / 4))
        sun = ref.sources.disk_bundle(n, N.c_[[0, 7., 7.]], N.array([0, x, x]), 3., 0.005, flux=1000.)
        eng = ref.tracer_engine.TracerEngine(dish)
        eng.ray_tracer(sun, 100, 1e-6)
        hist = dish.histogram_hits(bins=20)[0]
        plate = dish.get_receiver_surf().get_surfaces()[0]
        P.append(plate.get_optics_manager().get_all_hits()[0].sum())
        W.append([s_.get_optics_manager().get_all_hits()[0].sum() for s_ in dish.get_homogenizer().get_surfaces()])
        H.append(hist)
        sizes = [eng.tree[i].get_num_rays() for i in range(len(eng.tree._bunds))]
        L.append(sizes[:5] + [0] * (5 - len(sizes[:5])))
    out = dict(rays_per_run=N.int64(n), source_power=1000. * N.pi * 9.,
               receiver_runs=N.array(P), receiver_mean=N.mean(P), receiver_se=N.std(P, ddof=1) / N.sqrt(len(P)),
               walls_mean=N.mean(W, axis=0), walls_se=N.std(W, axis=0, ddof=1) / N.sqrt(len(W)),
               map_mean=N.mean(H, axis=0), map_se=N.std(H, axis=0, ddof=1) / N.sqrt(len(H)),
               levels_mean=N.mean(L, axis=0), levels_se=N.std(L, axis=0, ddof=1) / N.sqrt(len(L)))
    print('  minidish mc: %.1f s, receiver %.1f +- %.1f W of %.1f, walls %s, levels %s' %
          (time.time() - t0, out['receiver_mean'], out['receiver_se'], out['source_power'], N.round(out['walls_mean'], 1), out['levels_mean']))
    N.savez_compressed(os.path.join(HERE, 'mc_minidish.npz'), **out)


def plates_scene(T):
    """the scene of examples/accel_tree_example.py:20-53 from the modules of T (the reference or tracer_amd): two slabs, ten layers
    of 10 x 10 Lambertian plates, every object with its BoundaryBox"""
    n = 10
    side = n + 1.
    objects = []
    for z in (-1., None):
        slab = T.object.AssembledObject(T.surface.Surface(geometry=T.flat_surface.RectPlateGM(side, side), optics=T.optics_callables.LambertianReceiver(0.6)),
                                        bounds=T.boundary_shape.BoundaryBox([[-side / 2., -side / 2., 0.], [side / 2., side / 2., 0.]]))
        if z is not None:
            slab.set_location(N.array([0., 0., z]))
        objects.append(slab)
    for k in range(n):
        for i in range(n):
            for j in range(n):
                plate = T.object.AssembledObject(T.surface.Surface(geometry=T.flat_surface.RectPlateGM(.8, .8), optics=T.optics_callables.LambertianReceiver(0.9)),
                                                 bounds=T.boundary_shape.BoundaryBox([[-.4, -.4, 0.], [.4, .4, 0.]]))
                plate.set_location(N.array([i + 0.5 - n / 2., j + 0.5 - n / 2., k + 1.]))
                objects.append(plate)
    return T.assembly.Assembly(objects=objects), n, side


def make_mc_plates(ref):
    """
    examples/accel_tree_example.py traced by the reference itself (brute force: its Kd-tree build stops on this scene, where no
    object is without bounds -- N.hstack of an empty list, accel_tree.py:73): 8 seeds x 2e4 rays of the example's source, default
    reps and min_energy as in the example.  Power absorbed by the slab the sun can reach, by each of the ten layers, in total.
    -> mc_plates.npz
    """
    import time
    asm, n, side = plates_scene(ref)
    eng = ref.tracer_engine.TracerEngine(asm)
    rays = 20000
    rows = []
    t0 = time.time()
    for k in range(8):
        N.random.seed(5000 + k)
        asm.reset_all_optics()
        src = ref.sources.oblique_solar_rect_bundle(num_rays=rays, center=N.vstack([0, 0, n + 1]), source_direction=N.hstack([0, 0, -1]),
                                                    rays_direction=N.hstack([0, 0, -1]), x=side, y=side, ang_range=4.65e-3, flux=1000.)
        eng.ray_tracer(src)
        per = N.array([N.sum(s_.get_optics_manager().get_all_hits()[0]) for s_ in asm.get_surfaces()])
        rows.append(N.r_[per.sum(), per[1], per[2:].reshape(10, 100).sum(axis=1)])
    rows = N.array(rows)
    out = dict(rays_per_run=N.int64(rays), source_power=1000. * side * side, runs=rows, mean=rows.mean(axis=0),
               se=rows.std(axis=0, ddof=1) / N.sqrt(len(rows)))
    print('  plates mc: %.1f s, total %.0f +- %.0f W of %.0f; slab %.0f +- %.0f; layers %s' %
          (time.time() - t0, out['mean'][0], out['se'][0], out['source_power'], out['mean'][1], out['se'][1], N.round(out['mean'][2:])))
    N.savez_compressed(os.path.join(HERE, 'mc_plates.npz'), **out)


def make_emissive(out):
    """
    emissive_losses.  radiosity_RTVF is called as it is (emissive_losses.py imports under Python 3).
    view_factors_3D.py is Python 2 (print statements, xrange) and cannot be imported; its class RTVF (lines 20-112: the
    constructor and test_precision, plain NumPy) is valid Python 3, so that slice of the file is executed from where it
    lies and driven with seeded pass matrices.  The view-factor matrices are the text-book values the reference keeps
    in emissive_losses_test.py:12-15 and :38-42.
    """
    import importlib.util
    spec = importlib.util.spec_from_file_location('ref_emissive_losses', os.path.join(REFERENCE, 'emissive_losses', 'emissive_losses.py'))
    em = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(em)
    vf_cyl2 = N.array([[0., 0.618, 0.210, 0.172], [0.309, 0.382, 0.204, 0.105], [0.105, 0.204, 0.382, 0.309], [0.172, 0.210, 0.618, 0.]])
    vf_holman = N.array([[0., 0.63, 0.195, 0.075, 0.1], [0.315, 0.37, 0.2175, 0.06, 0.0375], [0.0975, 0.2175, 0.37, 0.2175, 0.0975],
                         [0.0375, 0.06, 0.2175, 0.37, 0.315], [0.1, 0.075, 0.195, 0.63, 0.]])
    out['vf_cyl2'] = vf_cyl2
    out['vf_holman'] = vf_holman
    nan = float('nan')
    cases = [
        ('holman_8_17', vf_holman, N.array([N.pi * 1e-4, 2 * N.pi * 1e-4, 2 * N.pi * 1e-4, 2 * N.pi * 1e-4, N.pi * 1e-4]), N.array([1., 0.6, 0.6, 0.6, 0.6]),
         N.array([293.15, 1273.15, 1273.15, 1273.15, 1273.15]), None),
        ('cyl2_temperatures', vf_cyl2, N.array([N.pi, 2. * N.pi, 2. * N.pi, N.pi]), N.array([1., 0.5, 0.5, 0.5]), N.array([300., 400., 500., 450.]), None),
        ('cyl2_all_flux', vf_cyl2, N.array([N.pi, 2. * N.pi, 2. * N.pi, N.pi]), N.array([0.9, 0.5, 0.5, 0.5]), N.array([nan, nan, nan, nan]),
         N.array([10., 2000., 1500., 1000.])),
        ('cyl2_mixed', vf_cyl2, N.array([N.pi, 2. * N.pi, 2. * N.pi, N.pi]), N.array([1., 0.5, 0.5, 0.5]), N.array([300., 400., 500., nan]),
         N.array([nan, nan, nan, 1000.])),
    ]
    out['rad_names'] = N.array([c[0] for c in cases])
    for i, (name, VF, areas, eps, T, inc) in enumerate(cases):
        pre = 'rad%d_' % i
        out[pre + 'VF'], out[pre + 'areas'], out[pre + 'eps'], out[pre + 'T_in'] = VF, areas, eps, T.copy()
        out[pre + 'has_inc'] = N.array(inc is not None)
        if inc is not None:
            out[pre + 'inc'] = inc.copy()
        res = em.radiosity_RTVF(VF, areas, eps, T.copy(), None if inc is None else inc.copy())
        for key, val in zip(('AA', 'bb', 'J', 'E', 'T', 'q', 'Q'), res):
            out[pre + key] = N.asarray(val)

    # RTVF.test_precision: the class as the reference wrote it, lines 20-112 of its file
    with open(os.path.join(REFERENCE, 'emissive_losses', 'view_factors_3D.py')) as f:
        lines = f.read().split('\n')
    ns = {'N': N}
    exec(compile('\n'.join(lines[19:112]), 'view_factors_3D.py[20:112]', 'exec'), ns)
    RTVF = ns['RTVF']
    rng = N.random.RandomState(123)
    n_pass = 12
    for ci, (option, VF_true, areas, num_rays, precision) in enumerate([
            ('absolute', vf_cyl2, N.array([N.pi, 2. * N.pi, 2. * N.pi, N.pi]), 20000., 5e-5),
            ('relative', vf_holman, N.array([N.pi * 1e-4, 2 * N.pi * 1e-4, 2 * N.pi * 1e-4, 2 * N.pi * 1e-4, N.pi * 1e-4]), 50000., 1.5e-4)]):
        n = len(areas)
        est = RTVF(num_rays=num_rays, precision=precision, precision_option=option)
        est.areas = areas
        est.VF = N.zeros((n, n)); est.VF_esperance = N.zeros((n, n)); est.Qsum = N.zeros((n, n))
        est.stdev_VF = N.zeros((n, n)); est.p = N.zeros(n)
        pre = 'tp%d_' % ci
        out[pre + 'option'], out[pre + 'areas'], out[pre + 'precision'] = N.array(option), areas, N.array(precision)
        passes, counts, esp, std, prog = [], [], [], [], []
        for k in range(n_pass):
            # a pass: multinomial estimate of every row; uneven ray counts from the 4th pass on (an emitter switched off once)
            rc = N.ones(n) * num_rays
            if k >= 3:
                rc[k % n] = num_rays / 2.
            if k == 5:
                rc[1] = 0.
            # (row 2 loses 2e-4 of its rays through the rim: the summation rule is exercised too)
            VF = N.array([rng.multinomial(int(rc[i]), N.hstack((VF_true[i] / VF_true[i].sum() * (1. - 2e-4 * (i == 2)), 2e-4 * (i == 2))))[:n] / max(rc[i], 1.)
                          for i in range(n)])
            est.VF, est.ray_counts = VF, rc
            est.p = est.p + rc
            with N.errstate(all='ignore'):
                est.test_precision()
            passes.append(VF); counts.append(rc); esp.append(est.VF_esperance.copy()); std.append(est.stdev_VF.copy()); prog.append(est.progress.copy())
        out[pre + 'VF'], out[pre + 'ray_counts'] = N.array(passes), N.array(counts)
        out[pre + 'VF_esperance'], out[pre + 'stdev_VF'], out[pre + 'progress'] = N.array(esp), N.array(std), N.array(prog)



def make_host_samplers(ref, amd, out):
    """S1 direction samplers and the host-generated bundles: outputs of the reference under a fixed seed of numpy's global
    generator (the host functions of tracer_amd draw the same variates in the same order)"""
    nrm = N.random.RandomState(1).normal(size=(3, 50))
    nrm /= N.sqrt(N.sum(nrm ** 2, axis=0))
    nrm[:, 0], nrm[:, 1] = [0, 0, 1], [0, 0, -1]
    out['normals'] = nrm
    with N.errstate(all='ignore'):
        for key, fn, args in (('lambertian', 'Lambertian_directions', (1000, 0.7)), ('lambertian_zero', 'Lambertian_directions', (200, 0.)),
                              ('pillbox', 'pillbox_sunshape_directions', (500, 4.65e-3)), ('edge', 'edge_rays_directions', (500, 0.3)),
                              ('lambertian_normals', 'Lambertian_directions', (50, 0.5, nrm))):
            N.random.seed(11)
            out[key] = getattr(ref.sources, fn)(*args)
        N.random.seed(13)
        b = ref.sources.edge_rays_bundle(300, N.c_[[1., 2., 3.]], N.r_[0., 0.6, 0.8], 2., 0.2, flux=10., radius_in=0.5)
    out['edge_bundle_vertices'], out['edge_bundle_directions'], out['edge_bundle_energy'] = b.get_vertices(), b.get_directions(), b.get_energy()


def make_scattering(ref, amd, out):
    """
    Participating media: the two pure functions of the reference's scattering optics that run --
    ray_trace_utils/sampling.py:150-168 Henyey_Greenstein.sample (draws R, then the azimuths) and tracer/optics.py:214-239
    scattering (one draw per ray; sigma = 0 never scatters) -- with numpy's draws recorded, so that the device's variate -> sample
    maps are checked against the reference's with only the generator differing.  (The classes built on them,
    optics_callables.py:946-1036 / :1108-1172 / :1350-1376, do not run in the reference: Scattering._scatter reads names it
    never defines, RefractiveScattering.__init__ an undefined g_HG.)
    """
    import ray_trace_utils.sampling as sampling
    gs = N.array([0., 0.3, -0.6, 0.9, -0.95])
    n = 4000
    out['hg_g'] = gs
    for k, g in enumerate(gs):
        N.random.seed(100 + k)
        draws = N.random.uniform(size=2 * n)            # what sample() will draw: R (n), then the azimuth uniforms (n)
        N.random.seed(100 + k)
        th, phi = sampling.Henyey_Greenstein(g).sample(n)
        out['hg%d_R' % k], out['hg%d_U' % k], out['hg%d_theta' % k], out['hg%d_phi' % k] = draws[:n], draws[n:], th, phi
    sigma = N.r_[N.full(1500, 0.7), N.full(1500, 12.), N.zeros(500)]
    paths = N.random.RandomState(5).uniform(0.01, 3., size=len(sigma))
    N.random.seed(77)
    R = N.random.uniform(size=len(sigma))
    N.random.seed(77)
    scat, lengths = ref.optics.scattering(sigma.copy(), paths.copy())
    out['sc_sigma'], out['sc_paths'], out['sc_R'], out['sc_scattered'], out['sc_lengths'] = sigma, paths, R, scat, lengths


def main():
    import_reference()
    if '--mc' in sys.argv:
        make_mc(NS('tracer'))
        return
    if '--mc-minidish' in sys.argv:
        make_mc_minidish(NS('tracer'))
        return
    if '--mc-plates' in sys.argv:
        make_mc_plates(NS('tracer'))
        return
    ref = NS('tracer')
    amd = NS('tracer_amd')
    only = [a[len('--only='):] for a in sys.argv if a.startswith('--only=')]
    for fname, maker in (('geometry.npz', make_geometry), ('optics.npz', make_optics), ('sources.npz', make_sources),
                         ('engine.npz', make_engine), ('emissive.npz', lambda ref, amd, out: make_emissive(out)),
                         ('host_samplers.npz', make_host_samplers), ('scattering.npz', make_scattering)):
        if only and fname not in only:
            continue
        out = {}
        maker(ref, amd, out)
        N.savez_compressed(os.path.join(HERE, fname), **out)
    if only:
        return
    out = {}
    make_kdtree(ref, out)
    N.savez_compressed(os.path.join(HERE, 'kdtree_nsttf.npz'), **out)
    with open(os.path.join(HERE, 'accountant_names.json'), 'w') as f:
        json.dump(make_accountant_names(ref), f, indent=0, sort_keys=True)
    for fn in sorted(os.listdir(HERE)):
        print('%-28s %8d bytes' % (fn, os.path.getsize(os.path.join(HERE, fn))))


if __name__ == '__main__':
    main()
