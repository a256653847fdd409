"""Host-side logic (no GPU): bundles, frames, scene compilation, sources' host part, RNG plumbing, sharding."""
import numpy as N
import pytest

from tracer_amd.ray_bundle import RayBundle, concatenate_rays
from tracer_amd.assembly import Assembly
from tracer_amd.object import AssembledObject
from tracer_amd.surface import Surface
from tracer_amd.flat_surface import RectPlateGM, RoundPlateGM, FlatGeometryManager
from tracer_amd import optics_callables as opt
from tracer_amd.spatial_geometry import general_axis_rotation, rotation_to_z, rotx, roty, rotz, translate
from tracer_amd import _cabi, rng, sources, distributed
from tracer_amd.scene import compile_scene, NotNativeError, scene_arrays, TableScene


def test_ray_bundle_accessors_inherit_add_delete():
    """behaviour pinned by the reference's tests/test_ray_bundle.py:12-100 (restated inputs)"""
    v = N.arange(12.).reshape(3, 4)
    d = N.tile(N.c_[[0., 0., 1.]], (1, 4))
    b = RayBundle(vertices=v, directions=d, energy=N.r_[1., 2., 3., 4.], parents=N.arange(4), ref_index=N.ones(4), wavelengths=N.r_[5., 6., 7., 8.])
    assert b.get_num_rays() == 4 and b.has_property('wavelengths') and not b.has_property('spectra')
    assert N.array_equal(b.get_energy([1, 3]), [2., 4.])
    assert N.array_equal(b.get_vertices(N.r_[0, 2]), v[:, [0, 2]])
    c = b.inherit(N.r_[2, 0], energy=N.r_[9., 8.])
    assert N.array_equal(c.get_energy(), [9., 8.]) and N.array_equal(c.get_wavelengths(), [7., 5.])
    assert N.array_equal(c.get_vertices(), v[:, [2, 0]])
    s = b + c
    assert s.get_num_rays() == 6 and N.array_equal(s.get_parents(), [0, 1, 2, 3, 2, 0])
    k = b.delete_rays(N.r_[1, 2])
    assert N.array_equal(k.get_energy(), [1., 4.])
    b.set_energy(N.r_[0., 0.], selector=N.r_[0, 1])
    assert N.array_equal(b.get_energy(), [0., 0., 3., 4.])
    e = RayBundle.empty_bund()
    assert e.get_num_rays() == 0 and concatenate_rays([]).get_num_rays() == 0
    assert concatenate_rays([k, c]).get_num_rays() == 4
    with pytest.raises(AttributeError):
        b.get_nonexistent()


def test_frames_propagate_through_nested_assemblies():
    """transform_children semantics of tests/test_objects.py:191-269 (restated)"""
    s = Surface(FlatGeometryManager(), opt.perfect_mirror, location=N.r_[0., 0., 1.])
    o = AssembledObject(surfs=[s], transform=translate(1., 0., 0.))
    inner = Assembly(objects=[o], location=N.r_[0., 2., 0.])
    outer = Assembly(subassemblies=[inner], rotation=rotz(N.pi / 2.)[:3, :3])
    expected = N.dot(rotz(N.pi / 2.), N.dot(translate(0, 2., 0), N.dot(translate(1., 0, 0), translate(0, 0, 1.))))
    assert N.allclose(s._temp_frame, expected)
    outer.set_location(N.r_[5., 5., 5.])
    assert N.allclose(s._temp_frame[:3, 3], expected[:3, 3] + 5.)
    assert outer.get_surfaces() == [s] and outer.get_objects() == [o]
    # ordering contract: objects of sub-assemblies first, own objects last
    o2 = AssembledObject(surfs=[Surface(FlatGeometryManager(), opt.perfect_mirror)])
    top = Assembly(objects=[o2], subassemblies=[inner])
    assert top.get_objects() == [o, o2]


def test_rotation_helpers():
    R = general_axis_rotation(N.r_[0., 0., 1.], N.pi / 2.)
    assert N.allclose(R, [[0, -1, 0], [1, 0, 0], [0, 0, 1]])
    for v in (N.r_[0., 0., 1.], N.r_[0., 0., -1.], N.r_[0.6, 0., 0.8], N.r_[0.36, 0.48, 0.8]):
        M = rotation_to_z(v)
        assert N.allclose(M[:, 2], v) and N.allclose(N.dot(M.T, M), N.eye(3), atol=1e-14)
    assert N.allclose(N.dot(rotx(0.3), rotx(-0.3)), N.eye(4)) and N.allclose(roty(0.2)[:3, :3].T, roty(-0.2)[:3, :3])


def test_scene_compilation_and_table_roundtrip():
    m = Surface(RectPlateGM(2., 3.), opt.OneSidedRealReflectiveDetector(0.04, 1e-3, True))
    r = Surface(RoundPlateGM(1., 0.25), opt.LambertianReceiver(0.9), location=N.r_[0., 0., 5.], rotation=rotx(N.pi)[:3, :3])
    asm = Assembly(objects=[AssembledObject(surfs=[m]), AssembledObject(surfs=[r], transform=translate(0, 1., 0))])
    cs = compile_scene(asm)
    assert cs.n_surf == 2 and cs.capture == [True, True] and not cs.splits
    d0, d1 = cs.descs[0], cs.descs[1]
    assert (d0.gm_kind, d0.optics_kind) == (_cabi.GM_RECT, _cabi.OPT_ONE_SIDED_REAL_REFLECTIVE)
    assert list(d0.gm)[:2] == [1., 1.5] and list(d0.opt)[:3] == [0.04, 1e-3, 1.]
    assert (d1.gm_kind, d1.optics_kind) == (_cabi.GM_ROUND, _cabi.OPT_LAMBERTIAN) and list(d1.gm)[:2] == [1., 0.25]
    assert N.allclose(N.array(list(d1.frame)).reshape(3, 4)[:, 3], [0., 1., 5.])
    a = scene_arrays(cs)
    ts = TableScene(a['gm_kind'], a['optics_kind'], a['frames'], a['gm'], a['opt'], a['extra'], a['extra_off'], a['extra_len'])
    assert bytes(ts.descs)[:8] == bytes(cs.descs)[:8] and N.allclose(scene_arrays(ts)['frames'], a['frames'])

    class MyOptics(object):
        def __call__(self, geometry, rays, selector):
            return rays.inherit(selector)
    asm2 = Assembly(objects=[AssembledObject(surfs=[Surface(RectPlateGM(1., 1.), MyOptics())])])
    with pytest.raises(NotNativeError):
        compile_scene(asm2)
    with pytest.raises(ValueError):
        RectPlateGM(-1., 1.)
    with pytest.raises(ValueError):
        RoundPlateGM(1., 2.)


def test_source_descriptors_host_part():
    b = sources.buie_sunshape(1000, N.c_[[0., 0., 6.]], N.r_[0., 0., -1.], 2.5, 0.05, flux=1000., seed=3)
    desc, n, seed, off = b.source_args()
    assert (n, seed, off) == (1000, 3, 0) and b.is_pending() and b.get_num_rays() == 1000
    assert N.isclose(desc.energy, 1000. * N.pi * 2.5 ** 2 / 1000.)
    tab = N.array(list(desc.buie))
    cdf = tab[422:633]
    assert cdf[0] == 0. and N.all(N.diff(cdf) > 0) and cdf[-1] < 1. and tab[638] == 1.
    assert N.allclose(N.array(list(desc.rot_dir)).reshape(3, 3)[:, 2], [0., 0., -1.])
    r = sources.rect_bundle(10, N.c_[[0., 0., 1.]], N.r_[0., 0., -1.], 2., 3., 0.1)
    assert list(r.source_args()[0].p)[:4] == [2., 3., 0.1, 1.] and N.isclose(r.source_args()[0].energy, 1. / 10.)
    d = sources.disk_bundle(10, N.c_[[0., 0., 1.]], N.r_[0., 0., 1.], 1., N.pi / 2., flux=2.)
    assert N.isclose(d.source_args()[0].energy, N.pi / 10. * 2.)
    dc = sources.disk_bundle(10, N.c_[[0., 0., 1.]], N.r_[0., 0., 1.], 1., 0.1, x_cut=0.5)
    assert list(dc.source_args()[0].p)[5:7] == [1., 0.5]
    with pytest.raises(ValueError):
        sources.disk_bundle(10, N.c_[[0., 0., 1.]], N.r_[0., 0., 1.], 1., 0.1, x_cut=-1.)


def test_rng_plumbing_and_sharding():
    rng.seed(5)
    a = [rng.next_seed() for _ in range(3)]
    rng.seed(5)
    assert a == [rng.next_seed() for _ in range(3)] and len(set(a)) == 3
    covered = []
    for r in range(8):
        lo, hi = distributed.shard(1003, r, 8)
        covered += list(range(lo, hi))
    assert covered == list(range(1003))
    offs = sorted(distributed.batch_offset(s, r, 4, 100) for s in range(3) for r in range(4))
    assert offs == [100 * k for k in range(12)]


def test_compat_aliases():
    import sys
    saved = dict((k, v) for k, v in sys.modules.items() if k == 'tracer' or k.startswith('tracer.'))
    try:
        for k in list(saved):
            del sys.modules[k]
        import tracer_amd.compat as compat
        compat.install()
        from tracer.surface import Surface as S2
        from tracer.models.heliostat_field import HeliostatField
        from tracer.optics_callables import LambertianReceiver
        assert S2 is Surface and HeliostatField.__module__ == 'tracer_amd.models.heliostat_field'
        assert LambertianReceiver(1.)._native()[0] == _cabi.OPT_LAMBERTIAN
    finally:
        for k in [k for k in sys.modules if k == 'tracer' or k.startswith('tracer.')]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_boundary_shapes_in_bounds():
    """tests/test_boundary_surface.py::TestInBounds"""
    from tracer_amd.boundary_shape import BoundarySphere, BoundaryCylinder, BoundaryPlane
    from tracer_amd.spatial_geometry import rotx
    pts = N.array([[0., 0., 0.], [1., 1., 1.], [2., 2., 2.]])
    assert list(BoundarySphere(radius=2.).in_bounds(pts)) == [True, True, False]
    assert list(BoundaryCylinder(diameter=3.).in_bounds(pts)) == [True, True, False]
    plane = BoundaryPlane(rotation=rotx(-N.pi / 6)[:3, :3], location=N.r_[0., 1., 0.])
    assert list(plane.in_bounds(pts)) == [False, True, True]


def test_radial_stagger_positions():
    """tests/models/test_tower.py::TestRadialStagger"""
    from tracer_amd.models.heliostat_field import radial_stagger
    pos = radial_stagger(-N.pi / 4, N.pi / 4 + 0.0001, N.pi / 2, 5, 10, 1)
    assert N.allclose(N.sqrt(N.sum(pos ** 2, axis=1)), N.r_[5, 5, 7, 7, 9, 9, 6, 8])


def test_homogenized_receiver_dishes_layout():
    """models/homogenized_local_receiver.py:14-46, tau_minidish.py:22-103, PETAL_dish.py:12-50, SG4.py:14-43: what the
    constructors build (values taken from the reference's own instances of the same arguments)"""
    from tracer_amd import _cabi
    from tracer_amd.models.tau_minidish import MiniDish, standard_minidish, standard_minidish_measures
    from tracer_amd.models.PETAL_dish import PETAL
    from tracer_amd.models.SG4 import SG4
    md = MiniDish(5, 5, 0.9, 5.7, .4, 0.7, 0.9, 1.5)
    surfs = md.get_surfaces()
    kinds = [s.get_geometry_manager()._native()[0] for s in surfs]
    assert kinds == [_cabi.GM_RECT] * 5 + [_cabi.GM_PARAB_DISH]            # four duct walls, the receiver plate, the dish
    assert [type(s.get_optics_manager()).__name__ for s in surfs] == \
        ['OneSidedRealReflectiveDetector'] * 4 + ['OneSidedReflectiveReceiver', 'Reflective']
    rec = md.get_receiver_surf().get_surfaces()[0]
    assert rec is surfs[4] and md.get_main_reflector() is surfs[5] and len(md.get_homogenizer().get_surfaces()) == 4
    assert N.allclose(rec._temp_frame, [[1, 0, 0, 0], [0, -1, 0, 0], [0, 0, -1, 5.7], [0, 0, 0, 1]])       # looking down the axis
    # the duct stands on the plate and opens towards the dish: walls centred 0.35 below the plate, at x = +-0.2 and y = -+0.3
    centres = N.array([s._temp_frame[:3, 3] for s in surfs[:4]])
    assert N.allclose(centres, [[0.2, 0, 5.35], [-0.2, 0, 5.35], [0, -0.3, 5.35], [0, 0.3, 5.35]])
    inward = N.array([s._temp_frame[:3, 2] for s in surfs[:4]])                  # each wall's mirrored side faces the axis
    assert N.allclose(inward, [[-1, 0, 0], [1, 0, 0], [0, 1, 0], [0, -1, 0]])
    assert md.get_external_dimensions() == (5, 5.7)
    md.set_transform(roty(N.pi / 4))                                            # the whole collector turns as one
    assert N.allclose(rec._temp_frame[:3, 3], 5.7 * N.r_[N.sin(N.pi / 4), 0, N.cos(N.pi / 4)])
    assert N.allclose(standard_minidish_measures(1., 500., 1), (0.6035533905932736, 0.03963327297606011, 0.05196030660088499))
    dish, f, W, H = standard_minidish(1., 500., 1)
    assert dish.get_external_dimensions() == (1., f + H)
    petal = PETAL(5, 5, 0.9, 5.7, .4, 0.7, 0.9, 2.)
    assert petal.get_surfaces()[5].get_geometry_manager()._native()[0] == _cabi.GM_PARAB_HEX
    sg4 = SG4(25., 13.4, 0.05, 2e-3)
    assert abs(sg4.absDish - 0.05362651118924833) < 1e-15
    zones = sg4.get_surfaces()
    assert len(zones) == 2 and zones[0]._temp_frame[2, 3] == 0. and zones[1]._temp_frame[2, 3] == 0.0001
    assert [type(s.get_optics_manager()).__name__ for s in zones] == ['RealReflectiveReceiver'] * 2


def test_spherical_lens_focal_lengths_and_layout():
    """tests/models/test_spherical_lens.py::*::test_focal_length (lensmaker equation, exact) + the surfaces created"""
    from tracer_amd.models.spherical_lens import SphericalLens
    assert SphericalLens(diameter=1., depth=0.1, R1=10., R2=-10., refr_idx=1.5).focal_length() == 2. / (0.2 - 0.05 / 150)
    assert SphericalLens(diameter=1., depth=0.1, R1=-10., R2=10., refr_idx=1.5).focal_length() == 2. / (-0.2 - 0.05 / 150)
    pc = SphericalLens(diameter=1., depth=0.05, R1=10., R2=N.inf, refr_idx=1.5)
    assert pc.focal_length() == 20.
    assert len(pc.get_surfaces()) == 3                       # front cap, flat back, edge cylinder
    kinds = [s.get_geometry_manager()._native()[0] for s in pc.get_surfaces()]
    from tracer_amd import _cabi
    assert kinds == [_cabi.GM_SPHERE_CUT, _cabi.GM_ROUND, _cabi.GM_CYL_FINITE]


def test_homogenizer_layout():
    from tracer_amd.models.homogenizer import rect_homogenizer
    hmg = rect_homogenizer(5., 3., 10., 0.9)
    frames = [s.get_surfaces()[0] for s in hmg.get_objects()]
    hmg.transform_children(N.eye(4))
    normals = N.array([s._temp_frame[:3, 2] for s in frames])
    assert N.allclose(normals, [[-1, 0, 0], [1, 0, 0], [0, -1, 0], [0, 1, 0]], atol=1e-15)   # mirrors face the duct axis
    centres = N.array([s._temp_frame[:3, 3] for s in frames])
    assert N.allclose(centres, [[2.5, 0, 5], [-2.5, 0, 5], [0, 1.5, 5], [0, -1.5, 5]])


def test_stl_reader_writer_and_triangle_frames(tmp_path):
    """tracer_amd.stl_utils: binary and ASCII STL parsed with numpy alone; the frame of every triangle has z along its normal and
    maps the planar profile back onto the vertices (ray_trace_utils/stl_utils.py:178-210)"""
    from tracer_amd import stl_utils as su
    verts = N.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]], dtype=float)
    faces = N.array([[0, 2, 1], [0, 3, 2], [4, 5, 6], [4, 6, 7], [0, 1, 5], [0, 5, 4], [2, 3, 7], [2, 7, 6], [1, 2, 6], [1, 6, 5], [0, 4, 7], [0, 7, 3]])
    path = str(tmp_path / 'box.stl')
    su.make_stl(verts, faces, path)
    tri = su.load_stl(path)
    assert tri.shape == (12, 3, 3) and N.allclose(tri, verts[faces])
    ascii_path = str(tmp_path / 'two.stl')
    with open(ascii_path, 'w') as f:
        f.write('solid two\n')
        for t in tri[:2]:
            f.write('facet normal 0 0 0\n outer loop\n' + ''.join('  vertex %r %r %r\n' % tuple(float(x) for x in v) for v in t) + ' endloop\nendfacet\n')
        f.write('endsolid two\n')
    assert N.array_equal(su.load_stl(ascii_path), tri[:2])
    with open(str(tmp_path / 'bad.stl'), 'w') as f:
        f.write('not a mesh')
    with pytest.raises(ValueError):
        su.load_stl(str(tmp_path / 'bad.stl'))
    geoms, locs, rots = su.stl_to_tracer_geom(tri, 'polygon')
    for k in range(12):
        A, B, C = tri[k]
        normal = N.cross(B - A, C - B)
        assert N.allclose(rots[k][:, 2], normal / N.linalg.norm(normal)) and N.allclose(N.dot(rots[k], rots[k].T), N.eye(3))
        back = locs[k][:, None] + N.dot(rots[k], N.vstack((geoms[k].profile, N.zeros(3))))
        assert N.allclose(back.T, tri[k], atol=1e-12)
    with pytest.raises(ValueError):
        su.stl_to_tracer_geom(tri, 'quad')


def test_host_direction_samplers_equal_reference():
    """S1: Lambertian_directions / pillbox_sunshape_directions / edge_rays_directions / edge_rays_bundle draw from numpy's global
    generator what the reference draws, in its order: equal to the reference's outputs under the same seed"""
    from tracer_amd import sources
    from helpers import load
    g = load("host_samplers.npz")
    nrm = g['normals']
    for key, fn, args in (('lambertian', sources.Lambertian_directions, (1000, 0.7)), ('lambertian_zero', sources.Lambertian_directions, (200, 0.)),
                          ('pillbox', sources.pillbox_sunshape_directions, (500, 4.65e-3)), ('edge', sources.edge_rays_directions, (500, 0.3)),
                          ('lambertian_normals', sources.Lambertian_directions, (50, 0.5, nrm))):
        N.random.seed(11)
        got = fn(*args)
        assert N.allclose(got, g[key], rtol=0., atol=1e-13), key
        assert N.allclose(N.sum(got ** 2, axis=0), 1.)
    N.random.seed(13)
    b = sources.edge_rays_bundle(300, N.c_[[1., 2., 3.]], N.r_[0., 0.6, 0.8], 2., 0.2, flux=10., radius_in=0.5)
    assert N.allclose(b.get_vertices(), g['edge_bundle_vertices'], atol=1e-13) and N.allclose(b.get_directions(), g['edge_bundle_directions'], atol=1e-13)
    assert N.allclose(b.get_energy(), g['edge_bundle_energy'])


def test_bench_reports_the_committed_traffic():
    """bench.py's roofline.traffic comes from profiles/traffic.json: the committed passes must be those of the default workload"""
    import importlib.util, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench_module', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    hi, lo = bench.traffic_for(1e8, True)
    assert hi is not None and lo is not None and 1e9 < lo <= hi < 2 * lo
    assert bench.traffic_for(1e7, True) == (None, None) and bench.traffic_for(1e8, False) == (None, None)


def test_estimator_running_mean_and_interval():
    """tracer_amd.estimator (the reference's ray_trace_utils/estimator.py:3-56 by name and arguments): the running weighted mean
    and its confidence interval against a direct evaluation of the formulas, batches of unequal size, vector-valued samples"""
    from tracer_amd.estimator import Estimator, MCRT_to_CI
    rng = N.random.default_rng(3)
    w = N.array([100., 250., 80., 400., 170.])
    x = rng.normal(5., 0.3, size=(5, 3))
    est = Estimator(n_sigmas=2.5, relative_CI=False)
    assert N.isinf(est.get_CI()).all()
    est.update(x[0], w[0])
    assert N.isinf(est.get_CI()).all() and N.allclose(est.mean, x[0])
    for k in range(1, 5):
        est.update(x[k], w[k])
    W, W2 = w.sum(), (w ** 2).sum()
    mean = (w[:, None] * x).sum(axis=0) / W
    S = (w[:, None] * (x - mean) ** 2).sum(axis=0)
    half = 2.5 * N.sqrt(S / (W - W2 / W)) / N.sqrt(W * W / W2)
    assert N.allclose(est.mean, mean, rtol=1e-13) and N.allclose(est.M2, S, rtol=1e-10) and N.allclose(est.get_CI(), half, rtol=1e-10)
    rel = Estimator(n_sigmas=2.5)
    for k in range(5):
        rel.update(x[k], w[k])
    assert N.allclose(rel.get_CI(), half / mean, rtol=1e-10)
    same = Estimator()
    same.update(N.r_[2.], 10); same.update(N.r_[2.], 30)
    assert same.get_CI()[0] == 0.
    calls = []
    def batch(num_rays):
        calls.append(num_rays)
        return N.r_[1. + 0.01 * rng.normal()]
    out = MCRT_to_CI(batch, 0.01, 1000)
    assert len(calls) >= 2 and out.get_CI()[0] <= 0.01 and abs(out.mean[0] - 1.) < 0.05
    import tracer_amd.compat as compat
    compat.install()
    from ray_trace_utils.estimator import Estimator as E2
    assert E2 is Estimator


def test_mesh_faces_kept_as_arrays_compile_like_one_surface_per_face(tmp_path):
    """
    models/triangulated_surface.py:12-52 and ray_trace_utils/stl_utils.py:178-235 make one Surface per face; here the faces of a mesh
    are one FaceSet of arrays (face_set.py) and a Surface exists once a script asks for it.  The device table compiled from the arrays
    is byte for byte the one compiled from the Surfaces; moving the object moves the faces; nothing is materialised by compiling.
    """
    import ctypes as C
    from tracer_amd import optics_callables as opt, stl_utils
    from tracer_amd.assembly import Assembly
    from tracer_amd.object import AssembledObject
    from tracer_amd.surface import Surface
    from tracer_amd.flat_surface import RectPlateGM
    from tracer_amd.face_set import FaceSet, SurfaceSeq
    from tracer_amd.models.triangulated_surface import TriangulatedSurface
    from tracer_amd.triangular_face import TriangularFace
    from tracer_amd.spatial_geometry import translate, rotx, rotz
    from tracer_amd.scene import compile_scene
    rng = N.random.default_rng(5)
    V = rng.normal(size=(40, 3))
    F = N.array([rng.choice(40, 3, replace=False) for _ in range(120)])
    F[7] = (3, 3, 9)                 # degenerate: dropped, as in the reference
    shared = opt.Reflective(0.2)
    mesh = TriangulatedSurface(V, F, shared, transform=N.dot(translate(0.3, -0.2, 1.), rotx(0.4)))
    plate = AssembledObject(surfs=[Surface(RectPlateGM(2., 3.), opt.LambertianReceiver(1.))], transform=translate(0., 0., 5.))
    asm = Assembly(objects=[plate, mesh])
    faces = mesh.get_surfaces()
    assert isinstance(faces, FaceSet) and len(faces) == 119 and len(faces._made) == 0
    surfaces = asm.get_surfaces()
    assert isinstance(surfaces, SurfaceSeq) and len(surfaces) == 120
    cs = compile_scene(asm)
    assert len(faces._made) == 0 and cs.n_surf == 120 and cs.optics[1] is shared and len(cs.optics) == 2
    eager = compile_scene(list(surfaces))            # one Surface per face now exists
    assert len(faces._made) == 119
    raw = lambda c: bytes(C.string_at(C.addressof(c.descs), C.sizeof(c.descs)))
    assert raw(cs) == raw(eager) and N.array_equal(cs.frames12(), eager.frames12())
    assert compile_scene(asm).signature() == cs.signature() and compile_scene(asm).descs is not cs.descs      # (compiled again: the same table)
    s5 = surfaces[6]
    assert s5 is faces[5] and isinstance(s5.get_geometry_manager(), TriangularFace) and s5.get_optics_manager() is shared
    assert surfaces.index(s5) == 6 and surfaces[-1] is faces[118] and [s for s in surfaces][0] is plate.get_surfaces()[0]
    # moving the assembly moves the faces, made or not
    asm.set_transform(N.dot(translate(1., 2., 3.), rotz(0.3)))
    moved = compile_scene(asm)
    assert raw(moved) == raw(compile_scene(list(asm.get_surfaces()))) and raw(moved) != raw(cs)
    assert moved.signature_without_frames() == cs.signature_without_frames() and moved.signature() != cs.signature()
    asm.reset_all_optics()
    # an STL file of the same triangles: frames as stl_to_tracer_geom gives them one by one, optics instances per face on demand
    tri = V[F[[0, 1, 2, 5]]]
    path = str(tmp_path / 'm.stl')
    stl_utils.make_stl(V, F[[0, 1, 2, 5]], path)
    obj = stl_utils.load_stl_into_tracer(path, opt.Reflective, dict(absorptivity=0.3), option='triangle')
    fs = obj.get_surfaces()
    assert isinstance(fs, FaceSet) and len(fs) == 4 and len(obj.get_boundaries()) == 4
    geoms, locs, rots = stl_utils.stl_to_tracer_geom(stl_utils.load_stl(path), option='triangle')
    for k in range(4):
        assert N.allclose(fs[k].get_location(), locs[k], atol=0) and N.allclose(fs[k].get_rotation(), rots[k], atol=1e-15)
        assert N.allclose(fs[k].get_geometry_manager()._verts, geoms[k]._verts, atol=1e-15)
    assert fs[0].get_optics_manager() is not fs[1].get_optics_manager() and fs[2] is fs[2]
    box = obj.get_boundaries()[1]
    assert N.allclose(box._AABB, [tri[1].min(axis=0), tri[1].max(axis=0)], atol=1e-6)
    cs_stl = compile_scene(obj)
    assert raw(cs_stl) == raw(compile_scene(list(fs)))
