"""
GPU tests of the Python plugin API (run with -m gpu): the reference's own unit tests restated against tracer_amd's
classes -- same inputs, same expected values (file:line of the reference test in each docstring).  They exercise
the per-surface protocol kernels, the three engines through TracerEngine, the accountants and the compat aliases.
"""
import os
import math

import numpy as N
import pytest

pytestmark = pytest.mark.gpu

from tracer_amd.ray_bundle import RayBundle
from tracer_amd.surface import Surface
from tracer_amd.object import AssembledObject
from tracer_amd.assembly import Assembly
from tracer_amd.flat_surface import FlatGeometryManager, RectPlateGM, RoundPlateGM
from tracer_amd.triangular_face import TriangularFace
from tracer_amd.paraboloid import Paraboloid, ParabolicDishGM
from tracer_amd.sphere_surface import HemisphereGM, SphericalGM
from tracer_amd.cylinder import InfiniteCylinder, FiniteCylinder
from tracer_amd.spatial_geometry import generate_transform, rotx, roty, translate, general_axis_rotation
from tracer_amd.tracer_engine import TracerEngine
from tracer_amd import optics_callables as opt
from tracer_amd import optics, sources


def _bundle45():
    """the four rays at 45 degrees of tests/test_flat_geometry_manager.py:12-22"""
    dir = N.array([[1, 1, -1], [-1, 1, -1], [-1, -1, -1], [1, -1, -1]]).T / math.sqrt(3)
    position = N.c_[[0, 0, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1]]
    return RayBundle(position, dir)


def test_flat_gm_known_answers():
    """tests/test_flat_geometry_manager.py:10-53 (t = sqrt(3), normals, hit points), :56-80 (plane tilted by 45 deg),
    :139-152 (back-side hit flips the normal)"""
    gm = FlatGeometryManager()
    b = _bundle45()
    t = gm.find_intersections(N.eye(4), b)
    assert t.shape == (4,) and N.allclose(t, N.sqrt(3))
    gm.select_rays(N.arange(4))
    assert N.array_equal(gm.get_normals(), N.tile(N.c_[[0, 0, 1]], (1, 4)))
    correct = N.zeros((3, 4)); correct[:2, 0] = 1
    assert N.allclose(gm.get_intersection_points_global(), correct, atol=1e-15)
    gm.select_rays(N.r_[1, 3])
    assert N.array_equal(gm.get_normals(), N.tile(N.c_[[0, 0, 1]], (1, 2)))
    gm.done()
    s2 = math.sqrt(2)
    dir = N.c_[[1, 0, -s2], [-1, 0, -s2], [-1, -s2, 0], [1, -s2, 0]] / math.sqrt(3)
    position = N.c_[[0, 1 / s2, 1 / s2], [1, 0, s2], [1, s2, 0], [-1, s2, 0]]
    frame = generate_transform(N.r_[1., 0, 0], -N.pi / 4., N.zeros((3, 1)))
    t = gm.find_intersections(frame, RayBundle(position, dir))
    assert t.shape == (4,) and N.allclose(t, math.sqrt(3))
    gm.select_rays(N.arange(4))
    assert N.allclose(gm.get_normals(), N.tile(N.c_[[0, 1 / s2, 1 / s2]], (1, 4)))
    gm.done()
    up = RayBundle(N.c_[[0., 0., -1.]], N.c_[[0., 0., 1.]])           # from below: the normal opposes the ray
    t = gm.find_intersections(N.eye(4), up)
    gm.select_rays(N.r_[0])
    assert N.allclose(t, 1.) and N.allclose(gm.get_normals(), N.c_[[0., 0., -1.]])


def test_rect_round_triangle_apertures():
    """tests/test_rect_plate.py:13-27, test_round_plate.py, test_triangular_face.py:11-38"""
    pos = N.zeros((3, 4)); pos[0] = N.r_[0, 0.5, 2, -2]; pos[2] = 1.
    bund = RayBundle(pos, N.tile(N.c_[[0, 0, -1]], (1, 4)).astype(float))
    surf = Surface(RectPlateGM(1, 0.25), opt.perfect_mirror)
    assert N.array_equal(N.isinf(surf.register_incoming(bund)), N.r_[False, False, True, True])
    with pytest.raises(ValueError):
        RectPlateGM(-1, 7)
    with pytest.raises(ValueError):
        RectPlateGM(1, -7)
    pos = N.zeros((3, 4)); pos[0] = N.r_[0., 0.4, 0.9, 1.1]; pos[2] = 1.
    down = RayBundle(pos, N.tile(N.c_[[0., 0., -1.]], (1, 4)))
    assert N.array_equal(N.isfinite(RoundPlateGM(1.).find_intersections(N.eye(4), down)), [True, True, True, False])
    assert N.array_equal(N.isfinite(RoundPlateGM(1., 0.5).find_intersections(N.eye(4), down)), [False, False, True, False])
    tri = TriangularFace(N.array([[1., 0.], [0., 1.], [0., 0.]]))
    p2 = N.c_[[0.2, 0.2, 1.], [0.8, 0.8, 1.], [-0.1, 0.2, 1.], [0.5, 0.5, 1.]]
    t = tri.find_intersections(N.eye(4), RayBundle(p2, N.tile(N.c_[[0., 0., -1.]], (1, 4))))
    assert N.array_equal(N.isfinite(t), [True, False, False, True]) and N.allclose(t[[0, 3]], 1.)


def test_paraboloid_and_dish():
    """tests/test_paraboloid_gm.py:9-41: ten rays on the unit circle, Paraboloid(a=5, b=5): t = 0.96, z = 0.04, normals
    pointing to the axis; ParabolicDishGM: rays outside the aperture miss"""
    n = 10
    theta = N.linspace(0, 2 * N.pi, n, endpoint=False)
    position = N.vstack((N.cos(theta), N.sin(theta), N.ones(n)))
    b = RayBundle(position, N.tile(N.c_[[0., 0., -1.]], (1, n)))
    gm = Paraboloid(a=5., b=5.)
    t = gm.find_intersections(N.eye(4), b)
    assert t.shape == (n,) and N.allclose(t, 0.96)
    gm.select_rays(N.arange(n))
    nrm = gm.get_normals()
    assert N.allclose(nrm[-1, 0], nrm[-1, 1:])
    assert N.allclose(position[:2], -nrm[:2] / N.sqrt((nrm[:2] ** 2).sum(axis=0)))
    pts = gm.get_intersection_points_global()
    assert N.allclose(pts[:2], position[:2]) and N.allclose(pts[2], 0.04)
    dish = ParabolicDishGM(1., 1.)        # aperture radius 0.5: the unit-circle rays miss, inner ones hit
    assert N.all(N.isinf(dish.find_intersections(N.eye(4), b)))
    inner = RayBundle(position * N.c_[[0.3, 0.3, 1.]], N.tile(N.c_[[0., 0., -1.]], (1, n)))
    assert N.all(N.isfinite(dish.find_intersections(N.eye(4), inner)))


def test_hemisphere_and_cylinder():
    """tests/test_hemisphere_gm.py (t = 1 + 2 sin 60 deg), tests/test_cylinder.py:66-99 (rotated frame, finite height)"""
    gm = HemisphereGM(2.)
    pos = N.c_[[0., 0., 1.], [1., 0., 1.]]
    b = RayBundle(pos, N.tile(N.c_[[0., 0., -1.]], (1, 2)))
    t = gm.find_intersections(N.eye(4), b)
    assert N.allclose(t, [3., 1. + 2. * math.sin(math.pi / 3.)])
    cyl = InfiniteCylinder(diameter=1.)
    pos = N.c_[[-2., 0., 0.], [-2., 0.3, 5.]]
    b = RayBundle(pos, N.tile(N.c_[[1., 0., 0.]], (1, 2)))
    t = cyl.find_intersections(N.eye(4), b)
    assert N.allclose(t, [1.5, 2. - math.sqrt(0.25 - 0.09)])
    fin = FiniteCylinder(diameter=1., height=2.)
    assert N.array_equal(N.isfinite(fin.find_intersections(N.eye(4), b)), [True, False])
    frame = N.eye(4); frame[:3, :3] = rotx(N.pi / 2.)[:3, :3]          # axis along global y
    t = fin.find_intersections(frame, RayBundle(N.c_[[-2., 0.5, 0.]], N.c_[[1., 0., 0.]]))
    assert N.allclose(t, 1.5)
    with pytest.raises(ValueError):
        SphericalGM(-1.)


def test_optics_laws():
    """tests/test_optics.py:63-130: Fresnel R = 0.04 at normal incidence (n 1 -> 1.5), Snell incl. TIR, mirror law"""
    d = N.c_[[0., 0., -1.], [math.sin(0.5), 0., -math.cos(0.5)]]
    n = N.c_[[0., 0., 1.]]
    R = optics.fresnel(d, n, 1., 1.5)
    assert N.isclose(R[0], 0.04)
    assert N.allclose(optics.reflections(d, n), d * N.c_[[1., 1., -1.]])
    refr, dirs = optics.refractions(1., 1.5, d, n)
    assert refr.all() and N.allclose(dirs[:, 0], [0., 0., -1.])
    assert N.isclose(dirs[0, 1] * 1.5, math.sin(0.5))                     # Snell
    steep = N.c_[[math.sin(1.2), 0., -math.cos(1.2)]]
    refr, dirs = optics.refractions(1.5, 1., steep, n)                    # beyond the critical angle
    assert not refr.any() and dirs.shape == (3, 0)
    assert N.allclose(optics.fresnel(steep, n, 1.5, 1.), 1.)


def test_fresnel_conductor_laws():
    """optics.fresnel_to_attenuating / fresnel_conductor against the reference's values (fixture) -- optics.py:41-81"""
    from helpers import load
    from tracer_amd.optics_callables import TabulatedMaterial
    o = load('optics.npz')
    rp, rs, t2 = optics.fresnel_to_attenuating(float(o['fta_n1']), o['fta_m_re'] + 1j * o['fta_m_im'], o['fta_theta1'])
    assert N.allclose(rp, o['fta_rp'], rtol=1e-10, atol=1e-14) and N.allclose(rs, o['fta_rs'], rtol=1e-10, atol=1e-14)
    assert N.allclose(t2, o['fta_theta2'], rtol=1e-10, atol=1e-14)
    # normal incidence on a conductor: R = ((n-1)^2 + k^2) / ((n+1)^2 + k^2)
    mat = TabulatedMaterial(N.r_[0.4e-6, 0.8e-6], N.r_[0.2, 0.2], N.r_[3.4, 3.4])
    rp, rs, _ = optics.fresnel_conductor(N.c_[[0., 0., -1.]], N.c_[[0., 0., 1.]], N.r_[0.5e-6], mat)
    expect = ((0.2 - 1.) ** 2 + 3.4 ** 2) / ((0.2 + 1.) ** 2 + 3.4 ** 2)
    assert N.allclose(rp, expect, rtol=1e-12) and N.allclose(rs, expect, rtol=1e-12)


def test_optics_callables_and_accountants():
    """tests/test_opt_callable.py:20-61 (energies, parents, accumulation across calls), :92-109 (TIR), :139-189 (Lambertian)"""
    b = _bundle45()
    b.set_energy(N.ones(4))
    gm = FlatGeometryManager()
    gm.find_intersections(N.eye(4), b)
    sel = N.r_[0, 1, 3]
    gm.select_rays(sel)
    out = opt.Reflective(0.1)(gm, b, sel)
    assert N.allclose(out.get_energy(), 0.9) and N.array_equal(out.get_parents(), sel)
    assert N.allclose(out.get_directions(), b.get_directions()[:, sel] * N.c_[[1., 1., -1.]])
    assert N.allclose(out.get_vertices(), gm.get_intersection_points_global())
    rec = opt.ReflectiveReceiver(1.)
    rec(gm, b, sel); rec(gm, b, sel)
    absorbed, hits = rec.get_all_hits()
    assert absorbed.shape == (6,) and N.allclose(absorbed, 1.) and hits.shape == (3, 6)
    rec.reset()
    assert rec.get_all_hits()[0].shape == (0,)
    lam = opt.Lambertian(0.2)(gm, b, sel)
    assert N.allclose(lam.get_energy(), 0.8) and N.all(lam.get_directions()[2] >= 0.) and N.allclose(N.sum(lam.get_directions() ** 2, axis=0), 1.)
    # refraction out of glass beyond the critical angle: all reflected, index unchanged
    b2 = RayBundle(N.c_[[0., 0., 1.]], N.c_[[math.sin(1.2), 0., -math.cos(1.2)]], energy=N.r_[1.], ref_index=N.r_[1.5])
    gm.done(); gm.find_intersections(N.eye(4), b2); gm.select_rays(N.r_[0])
    out = opt.RefractiveHomogenous(1., 1.5, single_ray=False)(gm, b2, N.r_[0])
    assert out.get_num_rays() == 1 and N.allclose(out.get_energy(), 1.) and N.allclose(out.get_ref_index(), 1.5)


def test_engine_two_planes_and_depletion():
    """tests/test_tracer_engine.py TestTraceProtocol1 (:21-60): two perpendicular mirrors send the rays back; 'bundle depleted'"""
    s1 = Surface(FlatGeometryManager(), opt.perfect_mirror)
    s2 = Surface(FlatGeometryManager(), opt.Reflective(0.), rotation=general_axis_rotation(N.r_[1., 0., 0.], N.pi / 2.))
    asm = Assembly(objects=[AssembledObject(surfs=[s1]), AssembledObject(surfs=[s2], location=N.r_[0., 1., 0.])])
    d = N.array([[0., 0., 0., 0.], [1., 1., 1., 1.], [-1., -1., -1., -1.]]) / math.sqrt(2.)
    p = N.c_[[0., -0.5, 1.], [0.2, -0.3, 1.], [-0.2, -0.7, 1.], [0.5, -1., 1.]]
    eng = TracerEngine(asm)
    for engine in ('ordered', 'protocol'):
        b = RayBundle(p.copy(), d.copy(), energy=N.ones(4))
        v, dd = eng.ray_tracer(b, reps=10, min_energy=0.05, tree=True, engine=engine)
        assert eng.tree.num_bunds() == 3 and v.shape == (3, 0)             # two bounces, then every ray escapes
        assert N.allclose(eng.tree[2].get_directions(), -d)                # retro-reflection
        assert N.array_equal(eng.tree[1].get_parents(), N.arange(4))
    b = RayBundle(p.copy(), d.copy(), energy=N.ones(4))
    v, dd = eng.ray_tracer(b, reps=1, min_energy=0.05, tree=True)
    assert v.shape == (3, 4) and N.allclose(v[2], 0.)                      # reps exhausted: live rays returned


def test_tree_ordering_culled_rays_to_back():
    """tests/test_tracer_tree.py:106-153: surface-major order; rays under min_energy recorded after the live ones"""
    absorber = Surface(RectPlateGM(1., 1.), opt.Reflective(0.99), location=N.r_[-1., 0., 0.])
    mirror = Surface(RectPlateGM(1., 1.), opt.Reflective(0.), location=N.r_[1., 0., 0.])
    asm = Assembly(objects=[AssembledObject(surfs=[mirror]), AssembledObject(surfs=[absorber])])
    p = N.c_[[-1., 0., 1.], [1., 0.2, 1.], [-1.2, 0.1, 1.], [1.1, -0.2, 1.]]
    b = RayBundle(p, N.tile(N.c_[[0., 0., -1.]], (1, 4)), energy=N.ones(4))
    eng = TracerEngine(asm)
    eng.ray_tracer(b, reps=2, min_energy=0.05, tree=True)
    lvl = eng.tree[1]
    assert N.array_equal(lvl.get_parents(), [1, 3, 0, 2])                  # mirror (surface 0) first, culled absorber rays last
    assert N.allclose(lvl.get_energy(), [1., 1., 0.01, 0.01])


def test_models_and_compat_script():
    """tests/models/test_one_sided_mirror.py (energies front/back) and a scene script written against `tracer.*` names"""
    import tracer_amd.compat as compat
    compat.install(force=True)
    from tracer.models.one_sided_mirror import rect_one_sided_mirror
    from tracer.assembly import Assembly as TAssembly
    from tracer.ray_bundle import RayBundle as TBundle
    from tracer.tracer_engine import TracerEngine as TEngine
    m = rect_one_sided_mirror(2., 2., absorptivity=0.1)
    asm = TAssembly(objects=[m])
    p = N.c_[[0., 0., 1.], [0.5, 0.5, -1.]]
    d = N.c_[[0., 0., -1.], [0., 0., 1.]]
    eng = TEngine(asm)
    eng.ray_tracer(TBundle(p, d, energy=N.ones(2)), reps=1, min_energy=-1., tree=True)
    assert N.allclose(eng.tree[1].get_energy(), [0.9, 0.])                # front reflects 90 %, back absorbs everything
    det = m.get_surfaces()[0].get_optics_manager().get_all_hits()
    assert N.allclose(det[0], [0.1, 1.]) and det[1].shape == (3, 2) and det[2].shape == (3, 2)


def test_accel_example_scene_through_tracer_names():
    """examples/accel_tree_example.py:20-98 restated call for call on the aliased `tracer.*` names: 1000 Lambertian plates in ten
    layers over two slabs, every object with a BoundaryBox given before it is moved, the Coin3D star import, a single Surface
    passed positionally, `engine._asm`, `reset_all_optics`, and the three spellings of the acceleration keyword -- which give
    the same absorbed power surface by surface from one seed"""
    import logging
    import tracer_amd.compat as compat
    compat.install(force=True)
    from tracer.assembly import Assembly as TAssembly
    from tracer.object import AssembledObject as TObject
    from tracer.surface import Surface as TSurface
    from tracer.flat_surface import RectPlateGM as TRect
    from tracer.boundary_shape import BoundaryBox
    from tracer.optics_callables import LambertianReceiver
    from tracer.CoIn_rendering.rendering import Renderer                      # scripts import it; only instantiating it raises
    from tracer.tracer_engine import TracerEngine as TEngine
    from tracer.sources import oblique_solar_rect_bundle
    n = 10
    side = n + 1.
    objects = []
    for z in (-1., None):
        slab = TObject(TSurface(geometry=TRect(side, side), optics=LambertianReceiver(0.6)),
                       bounds=BoundaryBox([[-side / 2., -side / 2., 0.], [side / 2., side / 2., 0.]]))
        if z is not None:
            slab.set_location(N.array([0., 0., z]))
        objects.append(slab)
    for k in range(n):
        for i in range(n):
            for j in range(n):
                plate = TObject(TSurface(geometry=TRect(.8, .8), optics=LambertianReceiver(0.9)),
                                bounds=BoundaryBox([[-.4, -.4, 0.], [.4, .4, 0.]]))
                plate.set_location(N.array([i + 0.5 - n / 2., j + 0.5 - n / 2., k + 1.]))
                objects.append(plate)
    assembly = TAssembly(objects=objects)
    engine = TEngine(assembly, loglevel=logging.INFO)
    assert len(engine._asm.get_surfaces()) == 1002
    per_surface = {}
    for accel in ('lightweight', True, None):
        assembly.reset_all_optics()
        source = oblique_solar_rect_bundle(num_rays=200000, center=N.vstack([0, 0, n + 1]), source_direction=N.hstack([0, 0, -1]),
                                           rays_direction=N.hstack([0, 0, -1]), x=side, y=side, ang_range=4.65e-3, flux=1000., seed=7)
        if accel is None:
            engine.ray_tracer(source, seed=7)
        else:
            engine.ray_tracer(source, accel=accel, seed=7)
        per_surface[accel] = N.array([N.sum(s.get_optics_manager().get_all_hits()[0]) for s in engine._asm.get_surfaces()])
    total = per_surface[None].sum()
    assert 0.85 * 121000. < total < 0.89 * 121000.                           # the rest leaves between the plates and over the rim
    top = per_surface[None][2 + 900:]                                          # the layer the sun sees first takes 0.9 x 0.64 of it
    assert abs(top.sum() - 0.9 * 0.64 * 100 * 1000.) < 0.02 * 57600. + 0.1 * (total - 57600.)
    assert N.array_equal(per_surface['lightweight'], per_surface[None]) and N.array_equal(per_surface[True], per_surface[None])
    with pytest.raises(NotImplementedError):
        Renderer(engine)
    # the same scene without a recorded tree (fast engine): both of its forms, with and without the Kd-tree, end every ray alike
    tallies = {}
    for form in ('megakernel', 'stream'):
        for accel in (False, True):
            assembly.reset_all_optics()
            engine.reset_tallies()
            source = oblique_solar_rect_bundle(num_rays=200000, center=N.vstack([0, 0, n + 1]), source_direction=N.hstack([0, 0, -1]),
                                               rays_direction=N.hstack([0, 0, -1]), x=side, y=side, ang_range=4.65e-3, flux=1000., seed=7)
            engine.ray_tracer(source, accel=accel, seed=7, tree=False, fast_kernel=form)
            assert engine.stats['engine'] == 'fast'
            tallies[form, accel] = engine.get_tallies()
    a0, r0, h0 = tallies['megakernel', False]
    assert h0.sum() > 250000 and abs(a0.sum() - total) < 0.01 * total
    for key, (a1, r1, h1) in tallies.items():
        assert N.array_equal(h1, h0), key
        assert N.allclose(a1, a0, rtol=1e-9, atol=1e-9), key


def test_cut_sphere_gm():
    """tests/test_cut_sphere.py: a sphere of radius 2 trimmed to its bottom part by a bounding sphere"""
    from tracer_amd.sphere_surface import CutSphereGM
    from tracer_amd.boundary_shape import BoundarySphere
    n = 10
    theta = N.linspace(0, 2 * N.pi, n, endpoint=False)
    pos = N.vstack((N.cos(theta), N.sin(theta), N.ones(n)))
    bund = RayBundle(pos, N.tile(N.c_[[0, 0, -1]], (1, n)))
    gm = CutSphereGM(2., BoundarySphere(radius=4., location=N.r_[0., 0., -4 * N.sqrt(3) / 2.]))
    prm = gm.find_intersections(N.eye(4), bund)
    assert prm.shape == (n,) and N.allclose(prm, 1 + 2 * N.sin(N.pi / 3))
    gm.select_rays(N.arange(n))
    nrm = gm.get_normals()
    assert N.allclose(nrm[-1, 0], nrm[-1, 1:])
    assert N.allclose(pos[:2], -nrm[:2] / N.sqrt((nrm[:2] ** 2).sum(axis=0)))      # centre-pointing
    pts = gm.get_intersection_points_global()
    assert N.allclose(pts[:2], pos[:2], atol=1e-15) and N.allclose(pts[2], -2 * N.sin(N.pi / 3))


def test_homogenizer_first_hits():
    """tests/models/test_homogenizer.py::test_first_hits: one ray to each wall of a 5 x 3 x 10 duct"""
    from tracer_amd.models.homogenizer import rect_homogenizer
    eng = TracerEngine(rect_homogenizer(5., 3., 10., 0.9))
    pos = N.zeros((3, 4))
    pos[2] = 11.
    dirs = N.c_[[1, 0, -1], [-1, 0, -1], [0, 1, -1], [0, -1, -1]] / N.sqrt(2)
    v, d = eng.ray_tracer(RayBundle(pos, dirs, energy=N.ones(4) * 4., ref_index=N.ones(4)), 1, 0.05)
    assert N.allclose(d, N.c_[[-1, 0, -1], [1, 0, -1], [0, -1, -1], [0, 1, -1]] / N.sqrt(2))
    assert N.allclose(v, N.c_[[2.5, 0, 8.5], [-2.5, 0, 8.5], [0, 1.5, 9.5], [0, -1.5, 9.5]])
    assert N.allclose(eng.tree[1].get_energy(), 3.6)


def test_minidish_upright_and_rotated():
    """tests/models/test_minidish.py::test_upright/test_rotated: five rays down onto a 90 % dish with a 90 % homogenizer; the
    two inner rays reach the plate directly (90), the two outer ones off one duct wall (81), the axial ray is taken whole by the
    plate's back (the upstream test predates that entry: energies and abscissae below are the reference's, run on this scene)"""
    from tracer_amd.models.tau_minidish import MiniDish
    pos = N.zeros((3, 5))
    pos[0] = N.r_[-2:2:5j]
    pos[2] = 6.
    dirs = N.zeros((3, 5))
    dirs[2] = -1.
    for turn in (N.eye(4), roty(N.pi / 4)):
        md = MiniDish(5, 5, 0.9, 5.7, .4, 0.7, 0.9)
        md.set_transform(turn)
        eng = TracerEngine(md)
        bund = RayBundle(N.dot(turn[:3, :3], pos), N.dot(turn[:3, :3], dirs), energy=N.ones(5) * 100, ref_index=N.ones(5))
        eng.ray_tracer(bund, 1776, 0.05)
        plate = md.get_receiver_surf().get_surfaces()[0]
        energy, pts = plate.get_optics_manager().get_all_hits()
        x, y = plate.global_to_local(pts)[:2]
        assert N.allclose(y, 0)
        order = N.argsort(energy)
        assert N.allclose(energy[order], [81., 81., 90., 90., 100.], rtol=0, atol=1e-12)
        assert N.allclose(N.abs(x[order]), [13 / 120., 13 / 120., 14 / 99., 14 / 99., 0.], rtol=0, atol=1e-9)
        H, xe, ye = md.histogram_hits(bins=4)
        assert H.shape == (4, 4) and abs(H.sum() - 442.) < 1e-9 and N.allclose(xe, N.r_[-.2:.2:5j])


def test_engines_built_in_a_loop_leave_device_memory_bounded():
    """scripts build an assembly and an engine per run: freed device blocks wait in the library's pool for the next request of
    their size class (DevPool: blocks up to 128 MiB, at most 2 GiB idle; larger ones by exact size, 2 GiB more).  Thirty runs of growing size -- 60 GB of levels, scratch and hit buffers
    allocated and released in all -- end with no more than that held back, and every run gives the same physics."""
    import ctypes
    from tracer_amd.models.tau_minidish import MiniDish
    from tracer_amd.sources import solar_disk_bundle
    hip = ctypes.CDLL('libamdhip64.so')

    def free_bytes():
        free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
        assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
        return free.value
    x = -1 / math.sqrt(2)
    share = []
    free_before = None
    for i in range(30):
        n = 1000000 + 130003 * i
        dish = MiniDish(5., 6.25, 0.9, 6.95, 0.4, 0.7, 0.9)
        dish.set_transform(rotx(-N.pi / 4))
        sun = solar_disk_bundle(n, N.c_[[0, 7., 7.]], N.array([0, x, x]), 3., 0.005, flux=1000., seed=i)
        eng = TracerEngine(dish)
        eng.ray_tracer(sun, 100, 1e-6, tree=(i % 2 == 0))
        share.append(dish.histogram_hits(bins=10)[0].sum() / (1000. * math.pi * 9.))
        del eng, dish, sun
        if i == 0:
            free_before = free_bytes()                          # after the first run: context and library are up
    held = free_before - free_bytes()
    assert held < 4.5 * 2 ** 30, held
    assert max(share) - min(share) < 0.004 and abs(N.mean(share) - 0.6012) < 0.001


def test_without_the_device_pool():
    """TRC_DEV_POOL=0 (every freed block straight back to the driver) in a process of its own: the minidish known answers"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import numpy as N\n"
        "from tracer_amd.models.tau_minidish import MiniDish\n"
        "from tracer_amd.tracer_engine import TracerEngine\n"
        "from tracer_amd.ray_bundle import RayBundle\n"
        "pos = N.zeros((3, 5)); pos[0] = N.r_[-2:2:5j]; pos[2] = 6.\n"
        "d = N.zeros((3, 5)); d[2] = -1.\n"
        "for k in range(3):\n"
        "    md = MiniDish(5, 5, 0.9, 5.7, .4, 0.7, 0.9)\n"
        "    TracerEngine(md).ray_tracer(RayBundle(pos, d, energy=N.ones(5) * 100, ref_index=N.ones(5)), 1776, 0.05, tree=bool(k % 2))\n"
        "    e = md.get_receiver_surf().get_surfaces()[0].get_optics_manager().get_all_hits()[0]\n"
        "    assert N.allclose(N.sort(e), [81., 81., 90., 90., 100.]), e\n"
        "print('ok')\n")
    env = dict(os.environ, TRC_DEV_POOL='0', PYTHONPATH=root)
    out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith('ok'), out.stderr[-2000:]


def test_sg4_zones_and_petal_under_a_parallel_beam():
    """models/SG4.py:14-61 (two nested zones, the inner one met first) and models/PETAL_dish.py:12-50 (hexagonal aperture)
    under 2e5 axial rays: which zone a ray meets, the absorbed share, the perfect focus of the zone without slope error"""
    from tracer_amd.models.SG4 import SG4
    from tracer_amd.models.PETAL_dish import PETAL
    n = 200000
    g = N.random.default_rng(11)
    r, th = 12.5 * N.sqrt(g.random(n)), 2 * N.pi * g.random(n)
    pos = N.vstack((r * N.cos(th), r * N.sin(th), N.full(n, 20.)))
    dish = SG4(25., 13.4, 0.05, 0., dishDiameter_in=20., sigma_in=0.)
    eng = TracerEngine(dish)
    v, d = eng.ray_tracer(RayBundle(pos, N.tile(N.c_[[0., 0., -1.]], (1, n)), energy=N.ones(n), ref_index=N.ones(n)), 1, 1e-9,
                          seed=3)
    hits, absorbed = dish.get_all_hits()
    assert hits.shape == (3, n) and abs(dish.total_abs - n * dish.absDish) < 1e-6 * n
    outer, inner = [s.get_optics_manager().get_all_hits()[1] for s in dish.get_surfaces()]
    assert outer.shape[1] == (r > 10.).sum() and inner.shape[1] == (r <= 10.).sum()
    assert N.all(N.hypot(outer[0], outer[1]) > 10.) and N.all(N.hypot(inner[0], inner[1]) <= 10.)
    assert N.allclose(inner[2], (inner[0] ** 2 + inner[1] ** 2) / (4 * 13.4) + 0.0001)
    # no slope error: every reflected ray goes through the focus of its zone (the inner one is 0.1 mm higher)
    t = -(v[0] * d[0] + v[1] * d[1]) / N.maximum(d[0] ** 2 + d[1] ** 2, 1e-300)          # closest approach to the axis
    z_axis = v[2] + t * d[2]
    off_axis = N.hypot(v[0], v[1]) > 0.5
    lifted = N.hypot(v[0], v[1]) <= 10.
    assert N.allclose(z_axis[off_axis & ~lifted], 13.4, atol=1e-9) and N.allclose(z_axis[off_axis & lifted], 13.4001, atol=1e-9)

    petal = PETAL(5., 5., 0.9, 5.7, .4, 0.7, 0.9)
    m = 50000
    p2 = N.vstack((g.uniform(-2.5, 2.5, m), g.uniform(-2.5, 2.5, m), N.full(m, 6.)))
    eng = TracerEngine(petal)
    eng.ray_tracer(RayBundle(p2, N.tile(N.c_[[0., 0., -1.]], (1, m)), energy=N.ones(m), ref_index=N.ones(m)), 1, 1e-9, tree=True)
    on_dish = eng.tree[1].get_parents()[eng.tree[1].get_energy() == 0.9]
    # a regular hexagon of circumradius 2.5 with two vertices on the y axis (paraboloid.py:213-216)
    x, y = N.abs(p2[0, on_dish]), N.abs(p2[1, on_dish])
    assert N.all(x <= 2.5 * N.sqrt(3) / 2 + 1e-12) and N.all(y <= 2.5 - x / N.sqrt(3) + 1e-9)
    shadow = (N.abs(p2[0]) <= .2) & (N.abs(p2[1]) <= .2)                                    # the receiver's back is met first
    inside = (N.abs(p2[0]) <= 2.5 * N.sqrt(3) / 2) & (N.abs(p2[1]) <= 2.5 - N.abs(p2[0]) / N.sqrt(3))
    assert abs(on_dish.size - (inside & ~shadow).sum()) <= 0.002 * m                        # duct walls shade a sliver more


def test_spherical_lens_imaging():
    """tests/models/test_spherical_lens.py: paraxial rays reach the focus (biconvex, planoconvex), chief-ray image height
    of a biconcave lens, the edge cylinder exists.  RefractiveHomogenous draws reflect-or-refract per ray, so a fan of
    identical rays is sent and the ones that reached the screen are checked."""
    from tracer_amd.models.spherical_lens import SphericalLens
    from tracer_amd.models.one_sided_mirror import rect_one_sided_mirror
    k = 64

    def to_screen(lens, origin, direct, screen_z, seed):
        screen = rect_one_sided_mirror(5, 5)
        screen.set_transform(translate(0, 0, screen_z))
        rb = RayBundle(N.tile(origin, (1, k)), N.tile(direct, (1, k)), energy=N.ones(k), ref_index=N.ones(k))
        eng = TracerEngine(Assembly([lens, screen]))
        vert, _ = eng.ray_tracer(rb, 3, 1e-6, seed=seed)
        on_screen = N.abs(vert[2] - screen_z) < 1e-9
        assert on_screen.sum() > k // 2          # ~8 % are lost to the two Fresnel reflections
        return vert[:, on_screen]

    for lens in (SphericalLens(diameter=1., depth=0.1, R1=10., R2=-10., refr_idx=1.5),
                 SphericalLens(diameter=1., depth=0.05, R1=10., R2=N.inf, refr_idx=1.5)):
        hits = to_screen(lens, N.c_[[0., 0.001, 1.]], N.c_[[0., 0., -1.]], -lens.focal_length(), seed=5)
        assert N.all(N.abs(hits[1]) < 5e-5) and N.all(N.abs(hits[0]) < 1e-12)

    lens = SphericalLens(diameter=1., depth=0.1, R1=-10., R2=10., refr_idx=1.5)
    origin = N.c_[[0., 0.001, 1.]]
    f = lens.focal_length()
    m = f / (origin[2, 0] + f)
    hits = to_screen(lens, origin, -origin / N.linalg.norm(origin), -origin[2, 0] * m, seed=6)
    assert N.all(N.abs(hits[1] + m * origin[1, 0]) < 5e-5)

    # the bounding cylinder: a ray travelling inside the glass towards the rim meets it at r = 0.5
    for lens, z in ((SphericalLens(diameter=1., depth=0.1, R1=10., R2=-10., refr_idx=1.5), 0.08),
                    (SphericalLens(diameter=1., depth=0.1, R1=-10., R2=10., refr_idx=1.5), 0.08),
                    (SphericalLens(diameter=1., depth=0.05, R1=10., R2=N.inf, refr_idx=1.5), 0.001)):
        rb = RayBundle(N.tile(N.c_[[0., 0., z]], (1, k)), N.tile(N.c_[[1., 0., 0.]], (1, k)), energy=N.ones(k), ref_index=N.ones(k) * 1.5)
        verts, dirs = TracerEngine(Assembly([lens])).ray_tracer(rb, 1, 1e-6, seed=7)
        assert verts.shape == (3, k) and N.allclose(verts, N.c_[[0.5, 0., z]])
        assert set(dirs[0]) == {-1., 1.} and N.all(dirs[1:] == 0)       # reflected back or transmitted straight
    lens = SphericalLens(diameter=1., depth=0.05, R1=10., R2=N.inf, refr_idx=1.5)
    rb = RayBundle(N.c_[[0., 0., -0.01]], N.c_[[1., 0., 0.]], energy=N.r_[1.], ref_index=N.r_[1.5])
    verts, dirs = TracerEngine(Assembly([lens])).ray_tracer(rb, 1, 1e-6)
    assert verts.shape == (3, 0)                                         # below the flat back face: no cylinder there


def test_protocol_engine_with_python_plugin():
    """a user-defined optics callable (pure Python) in the scene: engine='auto' falls back to the protocol loop, native
    geometry still runs on the GPU, results equal the all-native scene"""
    class HalfMirror(object):
        def __call__(self, geometry, rays, selector):
            return rays.inherit(selector, vertices=geometry.get_intersection_points_global(),
                                direction=optics.reflections(rays.get_directions(selector), geometry.get_normals()),
                                energy=rays.get_energy(selector) * 0.5, parents=selector)

    def build(o1):
        s1 = Surface(RectPlateGM(4., 4.), o1)
        s2 = Surface(RoundPlateGM(3.), opt.ReflectiveReceiver(1.), location=N.r_[0., 0., 2.], rotation=rotx(N.pi)[:3, :3])
        return Assembly(objects=[AssembledObject(surfs=[s1]), AssembledObject(surfs=[s2])]), s2
    rng = N.random.RandomState(3)
    p = N.vstack((rng.uniform(-1., 1., (2, 50)), N.ones(50)))
    d = N.vstack((rng.normal(scale=0.2, size=(2, 50)), -N.ones(50)))
    d /= N.sqrt(N.sum(d ** 2, axis=0))
    res = []
    for o1 in (HalfMirror(), opt.Reflective(0.5)):
        asm, rec = build(o1)
        eng = TracerEngine(asm)
        eng.ray_tracer(RayBundle(p.copy(), d.copy(), energy=N.ones(50)), reps=5, min_energy=1e-6, tree=True)
        e, h = rec.get_optics_manager().get_all_hits()
        res.append((eng.tree.num_bunds(), e.copy(), h.copy()))
    assert res[0][0] == res[1][0] and N.allclose(res[0][1], res[1][1]) and N.allclose(res[0][2], res[1][2])
    assert len(res[0][1]) > 10


def test_sources_statistics():
    """tests/test_ray_bundle.py:102-183: directions inside the cone, uniform azimuth (KS), total energy"""
    from scipy import stats
    n = 20000
    b = sources.disk_bundle(n, N.c_[[0., 0., 0.]], N.r_[0., 0., 1.], 1., 0.05, flux=2., seed=11)
    d, v, e = b.get_directions(), b.get_vertices(), b.get_energy()
    assert N.all(N.arccos(d[2]) <= 0.05 + 1e-12) and N.all(N.sum(v[:2] ** 2, axis=0) <= 1. + 1e-12)
    assert N.isclose(e.sum(), 2. * N.pi)
    az = N.arctan2(d[1], d[0]) % (2. * N.pi)
    assert stats.kstest(az / (2. * N.pi), 'uniform').pvalue > 1e-3
    assert stats.kstest(N.sum(v[:2] ** 2, axis=0), 'uniform').pvalue > 1e-3      # r^2 uniform on the disc
    b2 = sources.disk_bundle(n, N.c_[[0., 0., 0.]], N.r_[0., 0., 1.], 1., 0.05, flux=2., seed=11)
    assert N.array_equal(b2.get_vertices(), v)                                     # same seed, same rays
    b3 = sources.buie_sunshape(n, N.c_[[0., 0., 6.]], N.r_[0., 0., -1.], 2.5, 0.05, flux=1000., seed=5)
    th = N.arccos(-b3.get_directions()[2])
    assert th.max() <= 43.6e-3 + 1e-9 and 0.9 < N.mean(th < 4.65e-3) < 0.99


def test_view_factors_of_a_cylindrical_cavity():
    """
    The view-factor workload of emissive_losses (configs[4]): Lambertian emission from the aperture disc of a cylinder of
    radius 1 m, depth 2 m in two 1 m wall sections, one bounce, black receivers.  Known answers (first row of the matrix in
    emissive_losses/emissive_losses_test.py:15-18, analytic coaxial-disc formula): [0, 0.618, 0.210, 0.172].
    """
    walls = [Surface(FiniteCylinder(diameter=2., height=1.), opt.Lambertian(1.)) for _ in range(2)]   # black, tallies only
    bottom = Surface(RoundPlateGM(1.), opt.Lambertian(1.))
    asm = Assembly(objects=[AssembledObject(surfs=[walls[0]], transform=translate(0, 0, 0.5)),
                            AssembledObject(surfs=[walls[1]], transform=translate(0, 0, 1.5)),
                            AssembledObject(surfs=[bottom], transform=translate(0, 0, 2.))])
    n = 4000000
    src = sources.disk_bundle(n, N.c_[[0., 0., 0.]], N.r_[0., 0., 1.], 1., N.pi / 2., seed=2)      # energies 1/n
    eng = TracerEngine(asm)
    eng.ray_tracer(src, reps=1, min_energy=1e-10, tree=False, seed=2)
    a, r, h = eng.get_tallies()
    def disc_to_disc(hgt):
        X = 1. + (1. + (1. / hgt) ** 2) / (1. / hgt) ** 2
        return 0.5 * (X - N.sqrt(X ** 2 - 4.))
    f_bottom, f_mid = disc_to_disc(2.), disc_to_disc(1.)
    exact = N.array([1. - f_mid, f_mid - f_bottom, f_bottom])
    assert N.allclose(exact, [0.618, 0.210, 0.172], atol=5e-4)
    sigma = N.sqrt(exact * (1. - exact) / n)
    assert N.all(N.abs(a - exact) <= 4. * sigma), (a, exact)
    assert n - 5 <= h.sum() <= n and N.isclose(a.sum(), 1., atol=2e-6)   # a ray starting on the rim can slip under the 1e-6 threshold

    # second row of the same matrix: the first wall section as the emitter (sources.vf_cylinder_bundle, S4), aperture closed
    # by a black disc: [aperture, itself, other wall, bottom] = [0.309, 0.382, 0.204, 0.105]
    aperture = Surface(RoundPlateGM(1.), opt.Lambertian(1.))
    asm2 = Assembly(objects=[AssembledObject(surfs=[aperture]),
                             AssembledObject(surfs=[Surface(FiniteCylinder(diameter=2., height=1.), opt.Lambertian(1.))], transform=translate(0, 0, 0.5)),
                             AssembledObject(surfs=[Surface(FiniteCylinder(diameter=2., height=1.), opt.Lambertian(1.))], transform=translate(0, 0, 1.5)),
                             AssembledObject(surfs=[Surface(RoundPlateGM(1.), opt.Lambertian(1.))], transform=translate(0, 0, 2.))])
    src = sources.vf_cylinder_bundle(n, 1., 1., N.c_[[0., 0., 0.5]], N.r_[0., 0., 1.], rays_in=True, seed=3)
    eng = TracerEngine(asm2)
    eng.ray_tracer(src, reps=1, min_energy=1e-10, tree=False, seed=3)
    a, r, h = eng.get_tallies()
    f_w_ap = 0.5 * (1. - f_mid)                   # reciprocity: A_ap / A_wall = 1/2
    f_w_bot = 0.5 * (f_mid - f_bottom)           # reciprocity with the bottom disc's view of the far wall section
    exact = N.array([f_w_ap, 1. - 2. * f_w_ap, f_w_ap - f_w_bot, f_w_bot])
    assert N.allclose(exact, [0.309, 0.382, 0.204, 0.105], atol=5e-4)
    sigma = N.sqrt(exact * (1. - exact) / n)
    assert N.all(N.abs(a - exact) <= 4. * sigma), (a, exact)
    assert h.sum() >= n - 50 and N.isclose(a.sum(), 1., atol=2e-5)

    # a frustum wall emitter (sources.vf_frustum_bundle): cone frustum r 1 -> 0.5 over depth 1 closed by two discs;
    # wall -> base discs by reciprocity with the analytic coaxial-disc factor
    from tracer_amd.cone import ConicalFrustum
    def discs(r1, r2, hgt):
        R1, R2 = r1 / hgt, r2 / hgt
        X = 1. + (1. + R2 ** 2) / R1 ** 2
        return 0.5 * (X - N.sqrt(X ** 2 - 4. * (R2 / R1) ** 2))
    f12 = discs(1., 0.5, 1.)                       # big disc -> small disc
    a_wall = N.pi * 1.5 * N.sqrt(0.25 + 1.)
    exact = N.array([N.pi * (1. - f12) / a_wall, N.pi * 0.25 * (1. - f12 * 4.) / a_wall])
    asm3 = Assembly(objects=[AssembledObject(surfs=[Surface(RoundPlateGM(1.), opt.Lambertian(1.))]),
                             AssembledObject(surfs=[Surface(RoundPlateGM(0.5), opt.Lambertian(1.))], transform=translate(0, 0, 1.)),
                             AssembledObject(surfs=[Surface(ConicalFrustum(z1=0., r1=1., z2=1., r2=0.5), opt.Lambertian(1.))])])
    src = sources.vf_frustum_bundle(n, 1., 0.5, 1., N.c_[[0., 0., 0.]], N.r_[0., 0., 1.], rays_in=True, seed=4)
    eng = TracerEngine(asm3)
    eng.ray_tracer(src, reps=1, min_energy=1e-10, tree=False, seed=4)
    a, r, h = eng.get_tallies()
    sigma = N.sqrt(exact * (1. - exact) / n)
    assert N.all(N.abs(a[:2] - exact) <= 4. * sigma), (a, exact)
    assert N.isclose(a.sum(), 1., atol=1e-4)


def test_sun_tracking_updates_frames_on_the_device():
    """a heliostat field re-aimed for another sun position: the engine sends new frames to the scene it already has on the
    device (trc_scene_update_frames) -- same results as an engine that compiles the re-aimed field from scratch"""
    from tracer_amd import scenes
    from tracer_amd.models.heliostat_field import solar_vector
    plant, field, rec, src = scenes.nsttf_field(n_heliostats=30)
    eng = TracerEngine(plant)
    ue, ve = scenes.nsttf_fluxmap_edges()
    eng.set_fluxmap(30, ue, ve)                   # the receiver does not move: its flux map survives the update
    n = 200000
    eng.ray_tracer(scenes.nsttf_source(n, src, seed=8), reps=20, min_energy=1e-10, tree=False, accel=True, seed=8)
    first_dev = eng._dev
    a0, _, h0 = eng.get_tallies()
    # afternoon sun: re-aim the field at the same aim point, new source direction
    az, zen = N.radians(60.), N.radians(50.)
    aim = N.tile(N.r_[0., 0., 60.], (30, 1))
    field.track_sun(az, zen, aim_points=aim)
    sun = solar_vector(az, zen)
    src2 = dict(src, center=N.vstack(300. * sun + N.r_[0., 90., 0.]), direction=-sun)
    eng.reset_tallies(); plant.reset_all_optics()
    eng.ray_tracer(scenes.nsttf_source(n, src2, seed=9), reps=20, min_energy=1e-10, tree=False, accel=True, seed=9)
    assert eng._dev is first_dev                   # updated in place
    a1, _, h1 = eng.get_tallies()
    f1 = eng.get_fluxmap(30).copy()
    fresh = TracerEngine(plant)
    fresh.set_fluxmap(30, ue, ve)
    plant.reset_all_optics()
    fresh.ray_tracer(scenes.nsttf_source(n, src2, seed=9), reps=20, min_energy=1e-10, tree=False, accel=True, seed=9)
    a2, _, h2 = fresh.get_tallies()
    assert N.array_equal(h1, h2) and N.allclose(a1, a2, rtol=1e-12) and N.allclose(f1, fresh.get_fluxmap(30), rtol=1e-12)
    assert not N.array_equal(h0, h1) and h1[30] > 0
    # and the ordered engine after an update (the Kd-tree is rebuilt for the new poses)
    plant.reset_all_optics()
    eng.reset_tallies()
    eng.ray_tracer(scenes.nsttf_source(20000, src2, seed=9), reps=20, min_energy=1e-10, tree=True, accel=True, seed=9)
    fresh.reset_tallies()
    fresh.ray_tracer(scenes.nsttf_source(20000, src2, seed=9), reps=20, min_energy=1e-10, tree=True, accel=True, seed=9)
    assert N.array_equal(eng.get_tallies()[2], fresh.get_tallies()[2])


def test_polygon_plates_and_triangulated_surface():
    """
    polygon.py (FlatSimplePolygonGM, PerforatedPolygonGM) and models/triangulated_surface.py: point-in-polygon on the device
    against the host copy of the rule, absorbed power of an L-shaped plate = flux x its area, perforations remove theirs,
    and two triangles of an indexed face set collect what the rectangle they tile collects.
    """
    from tracer_amd.polygon import FlatSimplePolygonGM, PerforatedPolygonGM
    from tracer_amd.models.triangulated_surface import TriangulatedSurface
    prof = N.array([[-1., -1., 0.2, 0.2, 0.6, 1.2, 1.2], [-0.8, 1., 1., 0.1, 0.3, 0.1, -0.8]])
    gm = FlatSimplePolygonGM(prof)
    rng = N.random.RandomState(4)
    pts = N.vstack((rng.uniform(-1.3, 1.5, 5000), rng.uniform(-1., 1.2, 5000), N.ones(5000)))
    t = gm.find_intersections(N.eye(4), RayBundle(pts, N.tile(N.c_[[0., 0., -1.]], (1, 5000))))
    closed = N.concatenate((prof, prof[:, :1]), axis=1)
    assert N.array_equal(N.isfinite(t), gm.in_poly(pts[:2], closed)) and 1500 < N.isfinite(t).sum() < 3500
    with pytest.raises(ValueError):
        FlatSimplePolygonGM(N.array([[0., 1.], [0., 1.]]))
    # shoelace area of the clockwise profile
    x, y = prof
    area = 0.5 * abs(N.sum(x * N.roll(y, -1) - N.roll(x, -1) * y))
    holes_c, holes_r = N.array([[-0.5, 0.4], [0.7, -0.4]]), N.array([0.3, 0.2])
    n = 2000000
    for geom, expect in ((FlatSimplePolygonGM(prof), area), (PerforatedPolygonGM(prof, holes_c, holes_r), area - N.pi * (0.3 ** 2 + 0.2 ** 2))):
        asm = Assembly(objects=[AssembledObject(surfs=[Surface(geom, opt.Lambertian(1.))])])
        src = sources.rect_bundle(n, N.c_[[0.1, 0.1, 2.]], N.r_[0., 0., -1.], 3., 2.6, 0., flux=1000., seed=8)
        eng = TracerEngine(asm)
        eng.ray_tracer(src, reps=1, min_energy=1e-10, tree=False, seed=8)
        a, r, h = eng.get_tallies()
        sigma = 1000. * 3. * 2.6 * N.sqrt((expect / 7.8) * (1. - expect / 7.8) / n)
        assert abs(a[0] - 1000. * expect) < 4. * sigma, (a[0], 1000. * expect)
    # the two triangles of a unit square, tilted and shifted, against the same square as a RectPlateGM
    verts = N.array([[0., 0., 0.], [1., 0., 0.], [1., 1., 0.], [0., 1., 0.], [0.5, 0.5, 0.]])
    faces = N.array([[0, 1, 2], [0, 2, 3], [0, 0, 1], [0, 4, 2]])           # one degenerate and one collinear face are dropped
    frame = N.dot(translate(0.3, -0.2, 0.1), rotx(0.3))
    mesh = TriangulatedSurface(verts, faces, opt.Lambertian(1.), transform=frame)
    assert len(mesh.get_surfaces()) == 2
    plate = AssembledObject(surfs=[Surface(RectPlateGM(1., 1.), opt.Lambertian(1.))], transform=N.dot(frame, translate(0.5, 0.5, 0.)))
    got = []
    for obj in (mesh, plate):
        eng = TracerEngine(Assembly(objects=[obj]))
        eng.ray_tracer(sources.rect_bundle(400000, N.c_[[0.8, 0.3, 3.]], N.r_[0., 0., -1.], 2.5, 2.5, 0.02, flux=10., seed=6), reps=1,
                       min_energy=1e-10, tree=False, seed=6)
        a, r, h = eng.get_tallies()
        got.append((a.sum(), h.sum()))
    assert got[0][1] == got[1][1] and N.isclose(got[0][0], got[1][0], rtol=1e-12)


def test_attenuating_media_in_the_engines():
    """
    Absorbant.attenuate (optics_callables.py:874-889) through the engines: a slab of glass (RefractiveTransmissiveHomogenous on both
    faces) over a black Lambertian floor in an absorbing atmosphere (LambertianAbsorbant).  Normal incidence, known answer:
    what reaches the floor is (1-R)^2 exp(-a_glass d) exp(-a_air h) of what enters, R = ((n-1)/(n+1))^2; ordered and fast
    engines agree with the oracle ray by ray / tally by tally.
    """
    from oracle import engine as oeng
    from tracer_amd.scene import compile_scene
    n_glass, a_air, a_glass, thick, gap = 1.5, 0.2, 0.8, 0.3, 1.1
    def scene(single_ray=False):
        top = Surface(RectPlateGM(4., 4.), opt.RefractiveTransmissiveHomogenous(1., n_glass, [a_air, a_glass], single_ray=single_ray))
        bottom = Surface(RectPlateGM(4., 4.), opt.RefractiveTransmissiveHomogenous(1., n_glass, [a_air, a_glass], single_ray=single_ray))
        floor = Surface(RectPlateGM(6., 6.), opt.LambertianAbsorbant(1., a_air))
        return Assembly(objects=[AssembledObject(surfs=[top], transform=translate(0, 0, gap + thick)),
                                 AssembledObject(surfs=[bottom], transform=translate(0, 0, gap)),
                                 AssembledObject(surfs=[floor])])
    n = 20000
    rng = N.random.RandomState(3)
    pos = N.vstack((rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), N.ones(n) * (gap + thick + 0.5)))
    dirs = N.tile(N.c_[[0., 0., -1.]], (1, n))
    en = N.ones(n) / n
    R = ((n_glass - 1.) / (n_glass + 1.)) ** 2
    first_pass = N.exp(-a_air * 0.5) * (1. - R) * N.exp(-a_glass * thick) * (1. - R) * N.exp(-a_air * gap)
    asm = scene()
    eng = TracerEngine(asm)
    eng.ray_tracer(RayBundle(pos, dirs, energy=en, ref_index=N.ones(n)), reps=1000, min_energy=1e-9 / n, tree=True, seed=4)
    a, r, h = eng.get_tallies()
    # the floor also receives the light that bounced inside the slab: geometric series in R^2 exp(-2 a_glass d)
    series = first_pass / (1. - R * R * N.exp(-2. * a_glass * thick))
    # (a surface's absorbed tally is E_in - E_out, AbsorptionAccountant :1638-1643: the attenuation over the last leg is
    # booked on the surface the leg ends on, so the floor's tally is what left the slab downwards)
    series /= N.exp(-a_air * gap)
    assert N.isclose(a[2], series, rtol=1e-6), (a[2], series)
    ref = oeng.trace_bundle(compile_scene(asm), pos, dirs, en, 1000, 1e-9 / n, 4, ref_index=N.ones(n))
    assert N.array_equal(h, ref['hits']) and N.allclose(a, ref['absorbed'], rtol=1e-9, atol=1e-15)
    assert [b.get_num_rays() for b in eng.tree._bunds] == [l['vertices'].shape[1] for l in ref['levels']]
    # single-ray mode runs on the fast engine: same physics statistically, and equal to the oracle on the same streams
    asm1 = scene(single_ray=True)
    eng1 = TracerEngine(asm1)
    eng1.ray_tracer(RayBundle(pos, dirs, energy=en, ref_index=N.ones(n)), reps=1000, min_energy=1e-9 / n, tree=False, seed=4)
    a1, r1, h1 = eng1.get_tallies()
    ref1 = oeng.trace_bundle(compile_scene(asm1), pos, dirs, en, 1000, 1e-9 / n, 4, ref_index=N.ones(n))
    assert N.array_equal(h1, ref1['hits']) and N.allclose(a1, ref1['absorbed'], rtol=1e-9, atol=1e-15)
    assert abs(a1[2] - series) < 5. * N.sqrt(series * (1. - series) / n)


def test_bifacial_and_periodic_boundary_plugins():
    """
    BiFacial (optics_callables.py:1877-1926) and PeriodicBoundary (:690-723), host compositions that run under the protocol
    engine: a plate that mirrors from above and absorbs 40 % from below, between two mirrors; and a periodic face that moves
    the rays one period along its normal with their direction and energy.
    """
    plate = Surface(RectPlateGM(4., 4.), opt.BiFacial(opt.Reflective(0.), opt.Reflective(0.4)))
    above = Surface(RectPlateGM(4., 4.), opt.ReflectiveReceiver(1.), location=N.r_[0., 0., 1.], rotation=rotx(N.pi)[:3, :3])
    below = Surface(RectPlateGM(4., 4.), opt.ReflectiveReceiver(1.), location=N.r_[0., 0., -1.])
    asm = Assembly(objects=[AssembledObject(surfs=[plate]), AssembledObject(surfs=[above]), AssembledObject(surfs=[below])])
    pos = N.c_[[0.1, 0., 0.5], [-0.2, 0.3, 0.5], [0.3, 0.1, -0.5], [0., -0.4, -0.5]]
    dirs = N.c_[[0., 0., -1.], [0., 0., -1.], [0., 0., 1.], [0., 0., 1.]]
    eng = TracerEngine(asm)
    eng.ray_tracer(RayBundle(pos, dirs, energy=N.ones(4)), reps=3, min_energy=1e-9, tree=True)
    e_up, h_up = above.get_optics_manager().get_all_hits()
    e_dn, h_dn = below.get_optics_manager().get_all_hits()
    assert N.allclose(N.sort(e_up), [1., 1.]) and N.allclose(N.sort(e_dn), [0.6, 0.6])
    assert N.allclose(N.sort(h_up[0]), [-0.2, 0.1]) and N.allclose(N.sort(h_dn[1]), [-0.4, 0.1])

    face = Surface(RectPlateGM(4., 4.), opt.PeriodicBoundary(2.5))
    target = Surface(RectPlateGM(6., 6.), opt.ReflectiveReceiver(1.), location=N.r_[0., 0., -1.5])
    asm = Assembly(objects=[AssembledObject(surfs=[face]), AssembledObject(surfs=[target])])
    d = N.c_[[0.3, 0., 1.], [0., -0.2, 1.]]
    d /= N.sqrt(N.sum(d ** 2, axis=0))
    start = N.c_[[0., 0., -1.], [0.5, 0.5, -1.]]
    eng = TracerEngine(asm)
    eng.ray_tracer(RayBundle(start, d, energy=N.r_[2., 3.]), reps=3, min_energy=1e-9, tree=True)
    e, h = target.get_optics_manager().get_all_hits()
    # flight: 1 in z up to the face, a jump of one period along the normal that faces the ray (back to z = -2.5, as into the
    # opposite face of a periodic cell), then 1 more in z up to the target at z = -1.5, which the rays started above
    expect = start + d / d[2] * 1. + N.c_[[0., 0., -2.5]] + d / d[2] * 1.
    order = N.argsort(e)
    assert N.allclose(e[order], [2., 3.]) and N.allclose(h[:, order], expect, atol=1e-9)
    assert eng.tree._bunds[1].get_num_rays() == 4 and N.allclose(N.sort(eng.tree._bunds[1].get_energy()), [0., 0., 2., 3.])


def test_tracer_engine_mp_merges_trees_and_hits():
    """
    TracerEngineMP.multi_ray_sim (tracer_engine_mp.py:19-130): three bundles traced "by three processes" = the three bundles
    traced one after another; the merged tree is their level-by-level concatenation with consistent parents, the receiver's
    accountant holds the hits of all three.
    """
    from tracer_amd.tracer_engine_mp import TracerEngineMP
    def build():
        mirror = Surface(RectPlateGM(4., 4.), opt.Reflective(0.1))
        rec = Surface(RoundPlateGM(3.), opt.ReflectiveReceiver(1.), location=N.r_[0., 0., 2.], rotation=rotx(N.pi)[:3, :3])
        return Assembly(objects=[AssembledObject(surfs=[mirror]), AssembledObject(surfs=[rec])]), rec
    rng = N.random.RandomState(5)
    bundles = []
    for k in range(3):
        m = 40 + 10 * k
        p = N.vstack((rng.uniform(-2.5, 2.5, (2, m)), N.ones(m)))          # some rays miss the mirror
        d = N.vstack((rng.normal(scale=0.2, size=(2, m)), -N.ones(m)))
        d /= N.sqrt(N.sum(d ** 2, axis=0))
        bundles.append((p, d, rng.uniform(0.5, 1.5, m)))
    asm, rec = build()
    eng = TracerEngineMP(asm)
    eng.multi_ray_sim([RayBundle(p.copy(), d.copy(), energy=e.copy()) for p, d, e in bundles], procs=3, minener=1e-9, reps=5, tree=True)
    with pytest.raises(Exception):
        eng.multi_ray_sim([RayBundle(*bundles[0][:2], energy=bundles[0][2])], procs=2)
    singles = []
    for p, d, e in bundles:
        a1, r1 = build()
        e1 = TracerEngine(a1)
        e1.ray_tracer(RayBundle(p.copy(), d.copy(), energy=e.copy()), reps=5, min_energy=1e-9, tree=True)
        singles.append((e1.tree, r1.get_optics_manager().get_all_hits()))
    tree = eng.tree
    assert tree.num_bunds() == max(t.num_bunds() for t, _ in singles)
    for level in range(tree.num_bunds()):
        parts = [t._bunds[level] for t, _ in singles if level < t.num_bunds()]
        assert tree._bunds[level].get_num_rays() == sum(b.get_num_rays() for b in parts)
        assert N.allclose(tree._bunds[level].get_vertices(), N.hstack([b.get_vertices() for b in parts]))
        assert N.allclose(tree._bunds[level].get_energy(), N.hstack([b.get_energy() for b in parts]))
        if level > 0:       # a ray starts where its parent ended up one level further -- with the shifted indices
            par = N.asarray(tree._bunds[level].get_parents())
            prev = tree._bunds[level - 1]
            assert par.max() < prev.get_num_rays()
            hit = tree._bunds[level].get_vertices() - prev.get_vertices()[:, par]
            along = prev.get_directions()[:, par]
            assert N.allclose(N.cross(hit.T, along.T), 0., atol=1e-9)
    e_all, h_all = rec.get_optics_manager().get_all_hits()
    assert N.allclose(N.sort(e_all), N.sort(N.hstack([h[0] for _, h in singles])))
    assert len(e_all) > 60


def test_stl_mesh_object(tmp_path):
    """ray_trace_utils/stl_utils.py:156-235 through tracer_amd.stl_utils: a closed box written to STL, loaded as polygons and as
    triangles; rays from inside all land on it, both forms and six RectPlateGM faces absorb the same per wall"""
    from tracer_amd import stl_utils as su
    verts = N.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]], dtype=float) * N.r_[2., 1.5, 1.]
    faces = N.array([[0, 2, 1], [0, 3, 2], [4, 5, 6], [4, 6, 7], [0, 1, 5], [0, 5, 4], [2, 3, 7], [2, 7, 6], [1, 2, 6], [1, 6, 5], [0, 4, 7], [0, 7, 3]])
    path = str(tmp_path / 'box.stl')
    su.make_stl(verts, faces, path)
    n = 100000
    rng = N.random.RandomState(8)
    pos = N.tile(N.c_[[0.7, 0.6, 0.4]], (1, n))
    dirs = rng.normal(size=(3, n))
    dirs /= N.sqrt(N.sum(dirs ** 2, axis=0))
    per_wall = []
    for option in ('polygon', 'triangle'):
        obj = su.load_stl_into_tracer(path, opt.Lambertian, dict(absorptivity=1.), option=option)
        assert len(obj.get_surfaces()) == 12
        eng = TracerEngine(Assembly(objects=[obj]))
        eng.ray_tracer(RayBundle(pos.copy(), dirs.copy(), energy=N.ones(n) / n), reps=1, min_energy=1e-12, tree=False, accel=True)
        a, r, h = eng.get_tallies()
        assert n - 5 <= h.sum() <= n                      # (a ray through an edge can slip between two triangles)
        per_wall.append(a.reshape(6, 2).sum(axis=1))
    assert N.allclose(per_wall[0], per_wall[1], atol=3e-5)
    # the same box from six plates: bottom, top, y = 0, y = 1.5, x = 2, x = 0 (the order of the faces above)
    plates = [(RectPlateGM(2., 1.5), translate(1., 0.75, 0.)), (RectPlateGM(2., 1.5), translate(1., 0.75, 1.)),
              (RectPlateGM(2., 1.), N.dot(translate(1., 0., 0.5), rotx(N.pi / 2.))), (RectPlateGM(2., 1.), N.dot(translate(1., 1.5, 0.5), rotx(N.pi / 2.))),
              (RectPlateGM(1., 1.5), N.dot(translate(2., 0.75, 0.5), generate_transform(N.r_[0., 1., 0.], N.pi / 2., N.c_[[0., 0., 0.]]))),
              (RectPlateGM(1., 1.5), N.dot(translate(0., 0.75, 0.5), generate_transform(N.r_[0., 1., 0.], N.pi / 2., N.c_[[0., 0., 0.]])))]
    box = Assembly(objects=[AssembledObject(surfs=[Surface(g, opt.Lambertian(1.))], transform=t) for g, t in plates])
    eng = TracerEngine(box)
    eng.ray_tracer(RayBundle(pos.copy(), dirs.copy(), energy=N.ones(n) / n), reps=1, min_energy=1e-12, tree=False)
    a, r, h = eng.get_tallies()
    assert N.allclose(a, per_wall[0], atol=3e-5)


def test_trapezoid_bundle():
    """sources.trapezoid_bundle (sources.py:599-642): two triangular bundles sharing the rays by area; every ray starts inside the
    isosceles trapezoid ABCD, energies 1/n"""
    A, B, C = N.r_[0., 0., 0.], N.r_[4., 0., 0.], N.r_[3., 2., 0.]
    n = 20000
    b = sources.trapezoid_bundle(n, A, B, C, ang_range=0.3, seed=4)
    v, d, e = b.get_vertices(), b.get_directions(), b.get_energy()
    assert v.shape == (3, n) and N.allclose(e, 1. / n) and N.allclose(v[2], 0.)
    # D = (1, 2, 0) by symmetry: inside means 0 <= y <= 2 and y / 2 <= x <= 4 - y / 2
    assert (v[1] >= -1e-12).all() and (v[1] <= 2. + 1e-12).all()
    assert (v[0] >= v[1] / 2. - 1e-12).all() and (v[0] <= 4. - v[1] / 2. + 1e-12).all()
    # ABC holds 4 of the 6 area units, ACD the other 2
    upper = N.sum(v[1] > 1.)
    assert abs(upper / float(n) - 2.5 / 6.) < 0.015        # the strip 1 < y < 2 holds 2.5 of the 6 area units
    assert (N.arccos(N.clip(d[2], -1., 1.)) <= 0.3 + 1e-9).all()
