"""
GPU tests of the streaming fast engine's own paths (run with -m gpu on the MI355X box), through the C-ABI.

The engine answers one question -- nearest hit of every ray, the reference's tie rule (tracer_engine.py:27-64) -- along several
routes: fresh rays of a plane source through the footprint map (k_s_cull + k_s_fresh, csrc/trc_footprint.h) or through the
general path (k_s_gen + k_s_walk + k_s_exact), continued rays through k_s_bounce or the general path, the megakernel as a third
form.  Every draw is a pure function of (seed, ray, event), so all routes must agree ray for ray: hit counts per surface are
compared exactly, energies to the order-of-summation noise of float64 atomics (1e-9).
"""
import ctypes as C
import os

import numpy as N
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    from tracer_amd import _cabi
    return _cabi.get_context(0)


class env(object):
    """environment knobs of the library for the duration of a block (read at every call)"""
    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = dict((k, os.environ.get(k)) for k in self.kw)
        for k, v in self.kw.items():
            os.environ[k] = str(v)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _trace(ctx, cs, bundle, reps=20, accel=True, stream=True, kd=None, fluxmap=None, hit_capacity=0, **knobs):
    from tracer_amd.scene import DeviceScene
    with env(**knobs):
        dev = DeviceScene(cs, ctx)
        if kd is not None:
            dev.set_kdtree(kd)
        if fluxmap is not None:
            dev.set_fluxmap(*fluxmap)
        if hit_capacity:
            dev.set_hit_capacity(hit_capacity)
        st, _ = dev.trace_fast(bundle(), reps, 1e-10, 1, accel=accel, stream=stream)
        a, r, h = dev.get_tallies()
        out = dict(a=a.copy(), r=r.copy(), h=h.copy(), segments=st.segments, hits=st.hits, launches=st.launches)
        if fluxmap is not None:
            out['fm'] = dev.get_fluxmap(fluxmap[0]).copy()
        if hit_capacity:
            out['captured'] = dev.get_hits()
        dev.close()
    return out


def _same(x, y, what=''):
    assert N.array_equal(x['h'], y['h']), what
    assert x['segments'] == y['segments'] and x['hits'] == y['hits'], what
    assert N.allclose(x['a'], y['a'], rtol=1e-9, atol=1e-12) and N.allclose(x['r'], y['r'], rtol=1e-9, atol=1e-12), what
    if 'fm' in x:
        assert N.allclose(x['fm'], y['fm'], rtol=1e-9, atol=1e-12), what


def test_source_start_points_float32_vs_float64(ctx):
    """
    The culling kernel decides from a float32 start point; the footprint map allows it eps = 1e-4 of the source's half extent
    from the float64 one (trc_footprint.h).  On the device (v_sqrt_f32 / v_sin_f32 / v_cos_f32): every source kind the map
    applies to, 2e6 rays each, the largest difference must stay below a tenth of eps.
    """
    from tracer_amd import _cabi, sources
    direction = N.r_[0.3, -0.2, -1.] / N.linalg.norm([0.3, -0.2, -1.])
    center = N.c_[[3., -2., 40.]]
    cases = [sources.buie_sunshape(10, center, direction, 163., 0.01, flux=1., seed=1),
             sources.rect_buie_sunshape(10, center, direction, 17., 15., 0.1, flux=1., seed=1),
             sources.disk_bundle(10, center, direction, 9., 0.004, flux=1., radius_in=2., angular_span=[0.3, 5.1], seed=1),
             sources.rect_bundle(10, center, direction, 16., 13., 0.01, flux=1., seed=1),
             sources.rect_bundle(10, center, N.r_[0., 0., -1.], 16., 13., 0.01, flux=1., seed=1)]
    n = 2000000
    for k, b in enumerate(cases):
        desc = b.source_args()[0]
        lx, ly = N.empty(n, dtype=N.float32), N.empty(n, dtype=N.float32)
        eps = C.c_double(0.)
        f32 = C.POINTER(C.c_float)
        _cabi.check(ctx.lib.trc_source_start32(ctx.handle, C.byref(desc), n, 77, 10 ** 9, lx.ctypes.data_as(f32), ly.ctypes.data_as(f32), C.byref(eps)))
        v = N.empty((3, n)); d = N.empty((3, n)); e = N.empty(n)
        rays = _cabi.make_rays(n, v[0], v[1], v[2], d[0], d[1], d[2], e)
        _cabi.check(ctx.lib.trc_source_generate(ctx.handle, C.byref(desc), n, 77, 10 ** 9, C.byref(rays)))
        rot = N.array(list(desc.rot_pos)).reshape(3, 3)
        loc = N.dot(rot.T, v - N.array(list(desc.center))[:, None])
        err = max(N.abs(loc[0] - lx).max(), N.abs(loc[1] - ly).max())
        assert eps.value > 0 and err < 0.1 * eps.value, (k, err, eps.value)
    # a source the map does not apply to says so
    wide = sources.disk_bundle(10, center, direction, 9., 1.2, flux=1., seed=1).source_args()[0]
    lx = N.empty(4, dtype=N.float32)
    rc = ctx.lib.trc_source_start32(ctx.handle, C.byref(wide), 4, 1, 0, lx.ctypes.data_as(C.POINTER(C.c_float)), lx.ctypes.data_as(C.POINTER(C.c_float)), None)
    assert rc == _cabi.ERR_UNSUPPORTED


def test_fresh_and_bounce_kernels_equal_the_general_path_nsttf(ctx):
    """
    NSTTF, 4e6 rays (two batches would need 2^23; one batch here): the default route (footprint map + k_s_bounce), the general
    path for the fresh rays (TRC_STREAM_FRESH=0), for the continued rays (TRC_STREAM_BOUNCE=0), for both, the Kd walk of the
    queues (TRC_STREAM_SEARCH=1), brute force, and the megakernel: identical hit counts per surface, tallies and flux map.
    """
    from tracer_amd import scenes
    from tracer_amd.accel_tree import KdTree
    from tracer_amd.scene import compile_scene
    plant, field, rec, src = scenes.nsttf_field()
    cs = compile_scene(plant)
    kd = KdTree(plant, 8 + 1.3 * N.log(cs.n_surf), min_leaf=1)
    ue, ve = scenes.nsttf_fluxmap_edges()
    n = 4000000
    bundle = lambda: scenes.nsttf_source(n, src, seed=5, ray_offset=123456789)
    kw = dict(kd=kd, fluxmap=(218, ue, ve), reps=100)
    ref = _trace(ctx, cs, bundle, **kw)
    assert ref['h'][218] > 0.06 * n and ref['h'][:218].sum() > 0.06 * n and ref['segments'] > 1.06 * n
    assert N.isclose(ref['fm'].sum(), ref['a'][218], rtol=1e-9)
    _same(ref, _trace(ctx, cs, bundle, TRC_STREAM_FRESH=0, **kw), 'fresh rays on the general path')
    _same(ref, _trace(ctx, cs, bundle, TRC_STREAM_BOUNCE=0, **kw), 'continued rays on the general path')
    _same(ref, _trace(ctx, cs, bundle, TRC_STREAM_FRESH=0, TRC_STREAM_BOUNCE=0, **kw), 'general path only')
    _same(ref, _trace(ctx, cs, bundle, TRC_STREAM_SEARCH=1, **kw), 'Kd walk')
    _same(ref, _trace(ctx, cs, bundle, TRC_STREAM_SEARCH=1, TRC_STREAM_FRESH=0, **kw), 'Kd walk, general path')
    _same(ref, _trace(ctx, cs, bundle, accel=False, **kw), 'all boxes (accel=False)')
    _same(ref, _trace(ctx, cs, bundle, accel=False, TRC_STREAM_FRESH=0, TRC_STREAM_BOUNCE=0, **kw), 'all boxes, general path')
    _same(ref, _trace(ctx, cs, bundle, stream=False, **kw), 'megakernel')
    _same(ref, _trace(ctx, cs, bundle, TRC_STREAM_FP_CELLS=128, **kw), 'coarser footprint map')
    _same(ref, _trace(ctx, cs, bundle, TRC_STREAM_STATIC=0, **kw), 'no pre-assigned chunks')


def _mixed_scene(seed, n_obj=24, flat_only=False):
    from tracer_amd.assembly import Assembly
    from tracer_amd.object import AssembledObject
    from tracer_amd.surface import Surface
    from tracer_amd.flat_surface import RectPlateGM, RoundPlateGM
    from tracer_amd.paraboloid import ParabolicDishGM
    from tracer_amd.sphere_surface import HemisphereGM
    from tracer_amd.cylinder import FiniteCylinder
    from tracer_amd.triangular_face import TriangularFace
    from tracer_amd import optics_callables as opt
    from tracer_amd.spatial_geometry import translate, rotx, roty, rotz
    rng = N.random.RandomState(seed)
    objs = []
    for k in range(n_obj):
        if flat_only:
            gm = [RectPlateGM(1.6, 0.9), RoundPlateGM(0.8), TriangularFace(N.c_[[1.2, 0., 0.], [0.2, 1.1, 0.]])][k % 3]
        else:
            gm = [RectPlateGM(1.6, 0.9), RoundPlateGM(0.8), ParabolicDishGM(1.6, 1.1), HemisphereGM(0.6), FiniteCylinder(0.8, 1.2)][k % 5]
        o = [opt.Reflective(0.1), opt.RealReflective(0.2, 3e-3), opt.LambertianReceiver(0.6), opt.Reflective(0.05)][k % 4]
        tr = N.dot(translate(*rng.uniform(-5, 5, 3)), N.dot(rotx(rng.uniform(0, 6.3)), N.dot(roty(rng.uniform(0, 6.3)), rotz(rng.uniform(0, 6.3)))))
        objs.append(AssembledObject(surfs=[Surface(gm, o)], transform=tr))
    return Assembly(objects=objs)


def test_fresh_and_bounce_kernels_on_mixed_scenes_and_sources(ctx):
    """
    Scenes of mixed kinds (the general instance of k_s_fresh: quadrics, spheres, cylinders) and of flat kinds only (its flat
    instance: plates, discs, triangles), under every source kind the footprint map applies to, a Buie source with a strong
    aureole (CSR 0.3: a third of the rays take the general path, listed per wave) and an oblique source: default route against
    the general path and the megakernel over 12 bounces.
    """
    from tracer_amd import sources
    from tracer_amd.scene import compile_scene
    direction = N.r_[0.25, -0.15, -1.] / N.linalg.norm([0.25, -0.15, -1.])
    center = N.c_[-30. * direction]
    n = 1500000
    srcs = [lambda s: sources.buie_sunshape(n, center, direction, 9., 0.05, flux=1., seed=s),
            lambda s: sources.buie_sunshape(n, center, direction, 9., 0.3, flux=1., seed=s),
            lambda s: sources.rect_buie_sunshape(n, center, direction, 17., 15., 0.02, flux=1., seed=s),
            lambda s: sources.disk_bundle(n, center, direction, 9., 0.006, flux=1., radius_in=1., angular_span=[0.2, 6.0], seed=s),
            lambda s: sources.rect_bundle(n, center, direction, 16., 14., 0.01, flux=1., seed=s),
            lambda s: sources.oblique_solar_rect_bundle(n, N.c_[[-7.5, 4.5, 30.]], N.r_[0., 0., -1.], direction, 22., 20., 0.008, flux=1., seed=s)]
    for flat_only in (False, True):
        cs = compile_scene(_mixed_scene(3 + flat_only, flat_only=flat_only))
        for k, make in enumerate(srcs):
            bundle = lambda: make(40 + k)
            ref = _trace(ctx, cs, bundle, reps=12)
            assert ref['hits'] > 0.02 * n and ref['segments'] > n, (flat_only, k)
            _same(ref, _trace(ctx, cs, bundle, reps=12, TRC_STREAM_FRESH=0, TRC_STREAM_BOUNCE=0), ('general', flat_only, k))
            _same(ref, _trace(ctx, cs, bundle, reps=12, stream=False), ('megakernel', flat_only, k))
            if k in (0, 4):
                _same(ref, _trace(ctx, cs, bundle, reps=12, accel=False), ('all boxes', flat_only, k))


def test_terminal_surfaces_are_finished_by_k_s_absorb(ctx):
    """
    Surfaces that end every ray (absorptivity 1: the receivers) among mirrors and partly absorbing walls.  k_s_bounce lists the hits on
    them apart and k_s_absorb finishes them (no optics sampled, eight waves per SIMD): tallies, flux map and the captured hits equal
    those of the route without the split (TRC_STREAM_ABSORB=0) and of the megakernel.  Capture: a Receiver (absorbed energy + hit
    points) is captured lean -- incident energy reads back as the absorbed one, directions as 0; a Detector keeps everything.
    """
    from tracer_amd import sources, optics_callables as opt
    from tracer_amd.assembly import Assembly
    from tracer_amd.object import AssembledObject
    from tracer_amd.surface import Surface
    from tracer_amd.flat_surface import RectPlateGM, RoundPlateGM
    from tracer_amd.spatial_geometry import translate, rotx, roty, rotz
    from tracer_amd.scene import compile_scene
    from tracer_amd import _cabi
    rng = N.random.RandomState(12)
    kinds = [lambda: opt.Reflective(0.1), lambda: opt.LambertianReceiver(1.0), lambda: opt.RealReflective(0.2, 3e-3),
             lambda: opt.ReflectiveDetector(1.0), lambda: opt.LambertianReceiver(0.6), lambda: opt.Reflective(0.05)]
    objs = []
    for k in range(30):
        gm = [RectPlateGM(1.8, 1.1), RoundPlateGM(0.9)][k % 2]
        tr = N.dot(translate(*rng.uniform(-5, 5, 3)), N.dot(rotx(rng.uniform(0, 6.3)), N.dot(roty(rng.uniform(0, 6.3)), rotz(rng.uniform(0, 6.3)))))
        objs.append(AssembledObject(surfs=[Surface(gm, kinds[k % 6]())], transform=tr))
    cs = compile_scene(Assembly(objects=objs))
    flags = N.array([cs.descs[i].flags for i in range(cs.n_surf)])
    lean = (flags & _cabi.SURF_CAPTURE_LEAN) != 0
    assert lean.sum() == 10 and ((flags & _cabi.SURF_CAPTURE_HITS) != 0).sum() == 15         # two Receivers lean, the Detector not
    direction = N.r_[0.25, -0.15, -1.] / N.linalg.norm([0.25, -0.15, -1.])
    n = 1500000
    bundle = lambda: sources.buie_sunshape(n, N.c_[-30. * direction], direction, 9., 0.05, flux=1., seed=77)
    ue = N.linspace(-0.9, 0.9, 21)
    kw = dict(reps=12, fluxmap=(1, ue, ue), hit_capacity=4 * n)
    ref = _trace(ctx, cs, bundle, **kw)
    term = N.array([k % 6 in (1, 3) for k in range(30)])
    assert ref['h'][term].sum() > 0.01 * n and ref['h'][~term].sum() > 0.05 * n and ref['segments'] > 1.05 * n
    assert N.isclose(ref['fm'].sum(), ref['a'][1], rtol=1e-9) and ref['a'][1] > 0

    def hits_sorted(c):
        o = N.lexsort((c['points'][2], c['points'][1], c['points'][0], c['surf']))
        return dict((k, (v[..., o] if isinstance(v, N.ndarray) else v)) for k, v in c.items())
    a = hits_sorted(ref['captured'])
    assert len(a['surf']) == ref['h'][(flags & _cabi.SURF_CAPTURE_HITS) != 0].sum()
    is_lean = lean[a['surf']]
    assert is_lean.any() and (~is_lean).any()
    assert N.array_equal(a['e_in'][is_lean], a['e_abs'][is_lean]) and not a['directions'][:, is_lean].any()
    assert N.allclose(N.sum(a['directions'][:, ~is_lean] ** 2, axis=0), 1.) and (a['e_in'][~is_lean] >= a['e_abs'][~is_lean]).all()
    for what, knobs in (('no split', dict(TRC_STREAM_ABSORB=0)), ('split behind a list', dict(TRC_STREAM_ABSORB=1)), ('megakernel', dict(stream=False)),
                        ('general path', dict(TRC_STREAM_BOUNCE=0)),
                        ('no pre-assigned chunks', dict(TRC_STREAM_STATIC=0))):
        other = _trace(ctx, cs, bundle, **dict(kw, **knobs))
        _same(ref, other, what)
        b = hits_sorted(other['captured'])
        assert N.array_equal(a['surf'], b['surf']), what
        for key in ('e_abs', 'e_in', 'points', 'directions'):
            assert N.allclose(a[key], b[key], rtol=1e-9, atol=1e-12), (what, key)


@pytest.mark.parametrize('n', [20000000, 125000000])
def test_dish_into_spectral_cavity_at_scale(ctx, n):
    """
    BASELINE configs[4] on one rank, at 2e7 rays and at the 1.25e8 of a rank's share of 1e9 rays over 8 GPUs: a dish with slope error
    focuses a Buie sun into a cavity whose walls have angle- and wavelength-dependent optics (frustum, cylinder, cone, annulus, a
    Fresnel conductor, a spectral mirror); every ray carries a wavelength.  Streaming form (two batches in flight) == megakernel:
    hit counts and segments exactly, energies to 1e-9; the first 3000 rays == the oracle ray for ray; the shares of the 2e7 rays
    per surface agree with those 3000 within 5 sigma; nothing is created: absorbed + still alive <= sent.  The classes of optics present -- mirror, diffuse walls, the
    conductor -- are shaded by three kernels off a hit list parted by class (k_s_partition).
    """
    from tracer_amd import scenes
    from tracer_amd.scene import DeviceScene
    from tracer_amd.ray_bundle import RayBundle
    from oracle import engine as oracle_engine
    ts, src = scenes.dish_cavity()
    b0 = scenes.dish_source(n, src, seed=9)
    v, d, e = N.asarray(b0.get_vertices()), N.asarray(b0.get_directions()), N.asarray(b0.get_energy())
    wl = N.random.RandomState(4).uniform(0.3e-6, 2.5e-6, n)
    reps, emin, seed = 12, 1e-3 * e[0], 31
    out = {}
    for name, stream in (('stream', True), ('mega', False)):
        dev = DeviceScene(ts, ctx)
        st, _ = dev.trace_fast(RayBundle(vertices=v, directions=d, energy=e, wavelengths=wl), reps, emin, seed, stream=stream)
        a, r, h = dev.get_tallies()
        out[name] = dict(a=a, r=r, h=h, segments=st.segments, hits=st.hits, left=st.energy_left, launches=st.launches)
        dev.close()
    A, B = out['stream'], out['mega']
    assert A['launches'] > 10 and B['launches'] == 1
    assert N.array_equal(A['h'], B['h']) and A['segments'] == B['segments'] and A['hits'] == B['hits']
    assert N.allclose(A['a'], B['a'], rtol=1e-9) and N.allclose(A['r'], B['r'], rtol=1e-9) and N.isclose(A['left'], B['left'], rtol=1e-9, atol=1e-12)
    assert (A['h'] > 1000).all() and A['h'][0] > n and A['segments'] > 4 * n       # every surface takes part; rays come back to the dish
    assert 0.5 * e.sum() < A['a'].sum() + A['left'] <= e.sum() * (1. + 1e-12)
    m = 3000
    with N.errstate(all='ignore'):
        o = oracle_engine.trace_bundle(ts, v[:, :m], d[:, :m], e[:m], reps, emin, seed, wavelengths=wl[:m])
    dev = DeviceScene(ts, ctx)
    st, _ = dev.trace_fast(RayBundle(vertices=v[:, :m], directions=d[:, :m], energy=e[:m], wavelengths=wl[:m]), reps, emin, seed, stream=True)
    a, r, h = dev.get_tallies()
    dev.close()
    assert N.array_equal(h, o['hits']) and st.segments == o['segments'] and N.allclose(a, o['absorbed'], rtol=1e-9, atol=1e-12)
    p_small, p_big = o['hits'] / float(m), A['h'] / float(n)
    sigma = N.sqrt(N.maximum(p_big, 1e-4) / m) * 2.        # hits per ray are not Bernoulli (a ray hits a wall several times): factor 2
    assert (N.abs(p_small - p_big) < 5. * sigma).all(), (p_small, p_big, sigma)


def test_modest_hit_buffer_on_a_large_streaming_call(ctx):
    """
    A caller-sized hit buffer just above the hits to come -- NSTTF, 1e7 rays, room for 1.2 times the 6.5 % of the rays that reach the
    receiver -- drops nothing: the chunk the shading waves reserve per atomic follows the buffer's size (256 entries here, 1024 for
    the buffers of full-size runs) and the buffer carries the room every wave that can hold an open chunk may leave unused.
    """
    from tracer_amd import scenes
    from tracer_amd.scene import compile_scene, DeviceScene
    plant, field, rec, src = scenes.nsttf_field()
    cs = compile_scene(plant)
    n = 10000000
    dev = DeviceScene(cs, ctx)
    dev.set_hit_capacity(int(1.2 * 0.065 * n))
    for k in range(2):                   # (the second call continues the chunks the first left open)
        dev.lib.trc_scene_clear_hits(dev.handle) if k == 0 else None
        st, _ = dev.trace_fast(scenes.nsttf_source(n, src, seed=8, ray_offset=k * n), 100, 1e-10, 8, accel=True, stream=True)
        assert st.hits_dropped == 0, (k, st.hits_dropped)
        if k == 0:
            a, r, h = dev.get_tallies()
            hits = dev.get_hits()
            assert len(hits['surf']) == h[218] and N.isclose(hits['e_abs'].sum(), a[218], rtol=1e-9)
            dev.lib.trc_scene_clear_hits(dev.handle)
            dev.reset_tallies()
    a, r, h = dev.get_tallies()
    assert len(dev.get_hits()['surf']) == h[218] > 0.06 * n
    dev.close()


def test_calls_in_sequence_accumulate_like_fresh_scenes(ctx):
    """
    What the library keeps between calls (counters shadowed on the host, the source descriptor on the device, private tally copies
    whose merge is not waited for, open chunks of the hit buffer) must never show: a scene traced several times -- streaming form,
    megakernel, hits cleared in between, tallies reset in between, another source -- reports per call and in total what fresh
    scenes report for the same calls.
    """
    from tracer_amd import scenes
    from tracer_amd.scene import compile_scene, DeviceScene
    plant, field, rec, src = scenes.nsttf_field(n_heliostats=40)
    cs = compile_scene(plant)
    n = 1500000
    ue, ve = scenes.nsttf_fluxmap_edges()
    S = cs.n_surf

    def fresh(offset, stream, seed=7, radius=None):
        dev = DeviceScene(cs, ctx)
        dev.set_fluxmap(S - 1, ue, ve)
        dev.set_hit_capacity(n)
        b = scenes.nsttf_source(n, dict(src, radius=radius or src['radius']), seed=seed, ray_offset=offset)
        st, _ = dev.trace_fast(b, 100, 1e-10, seed, accel=True, stream=stream)
        a, r, h = dev.get_tallies()
        out = dict(a=a, r=r, h=h, fm=dev.get_fluxmap(S - 1), seg=st.segments, hits=st.hits, left=st.energy_left, cap=len(dev.get_hits()['surf']))
        dev.close()
        return out
    calls = [(0, True, None), (n, True, None), (2 * n, False, None), (3 * n, True, 0.6 * src['radius']), (4 * n, True, None)]
    ref = [fresh(off, stream, radius=rad) for off, stream, rad in calls]
    dev = DeviceScene(cs, ctx)
    dev.set_fluxmap(S - 1, ue, ve)
    dev.set_hit_capacity(5 * n)
    tot = dict(a=0., r=0., h=0, fm=0., cap=0)
    for k, (off, stream, rad) in enumerate(calls):
        b = scenes.nsttf_source(n, dict(src, radius=rad or src['radius']), seed=7, ray_offset=off)
        st, _ = dev.trace_fast(b, 100, 1e-10, 7, accel=True, stream=stream)
        assert (st.segments, st.hits) == (ref[k]['seg'], ref[k]['hits']), k
        assert N.isclose(st.energy_left, ref[k]['left'], rtol=1e-9, atol=1e-12), k
        for key in ('a', 'r', 'h', 'fm', 'cap'):
            tot[key] = tot[key] + ref[k][key]
        if k in (0, 2, 3):              # read right after the call: the merge of the private copies must have happened
            a, r, h = dev.get_tallies()
            assert N.array_equal(h, tot['h']) and N.allclose(a, tot['a'], rtol=1e-9, atol=1e-12) and N.allclose(r, tot['r'], rtol=1e-9, atol=1e-12), k
            assert N.allclose(dev.get_fluxmap(S - 1), tot['fm'], rtol=1e-9, atol=1e-12), k
        if k == 1:
            assert len(dev.get_hits()['surf']) == tot['cap']
            dev.lib.trc_scene_clear_hits(dev.handle)
            tot['cap'] = 0
        if k == 3:
            dev.reset_tallies()
            tot = dict(a=0., r=0., h=0, fm=0., cap=0)
    a, r, h = dev.get_tallies()
    assert N.array_equal(h, ref[4]['h']) and N.allclose(a, ref[4]['a'], rtol=1e-9, atol=1e-12)
    assert len(dev.get_hits()['surf']) == ref[4]['cap'] > 0
    dev.close()


def test_list_overflows_are_reported_and_leave_no_trace(ctx):
    """
    The lists of the engine (ray table, footprint list, general-path list, walker queue, hit list, active list) are sized for
    the batch plus room for the chunks' unused tails; whatever overflows one sets flag 2 and the call ends with
    TRC_ERR_CAPACITY before anything reads beyond an allocation.  TRC_STREAM_ROOM (entries per list) provokes it on each
    route.  A failed call leaves nothing behind: the next call on the same scene gives what a fresh scene gives, tallies, flux
    map and captured hits alike (the private tally copies and the open chunks of the hit buffer are wound back).
    """
    from tracer_amd import _cabi, scenes
    from tracer_amd.accel_tree import KdTree
    from tracer_amd.scene import compile_scene, DeviceScene
    plant, field, rec, src = scenes.nsttf_field(n_heliostats=30)
    cs = compile_scene(plant)
    kd = KdTree(plant, 8 + 1.3 * N.log(cs.n_surf), min_leaf=1)
    ue, ve = scenes.nsttf_fluxmap_edges()
    n = 600000
    bundle = lambda: scenes.nsttf_source(n, src, seed=9)
    good = _trace(ctx, cs, bundle, kd=kd, fluxmap=(30, ue, ve), reps=100, hit_capacity=n)
    assert good['h'][30] > 1000 and len(good['captured']['surf']) == good['h'][30]
    # too little room on every route (TRC_STREAM_ROOM: entries of the ray table and of each list, instead of cap + slack)
    for knobs in (dict(), dict(TRC_STREAM_FRESH=0), dict(TRC_STREAM_BOUNCE=0), dict(TRC_STREAM_STATIC=0)):
        with env(TRC_STREAM_ROOM=2048, **knobs):
            dev = DeviceScene(cs, ctx)
            dev.set_kdtree(kd)
            dev.set_fluxmap(30, ue, ve)
            dev.set_hit_capacity(n)
            with pytest.raises(_cabi.TracerAmdError) as ei:
                dev.trace_fast(bundle(), 100, 1e-10, 1, accel=True, stream=True)
            assert ei.value.status == _cabi.ERR_CAPACITY, knobs
        # the same scene, room as usual: nothing of the failed call is left
        st, _ = dev.trace_fast(bundle(), 100, 1e-10, 1, accel=True, stream=True)
        a, r, h = dev.get_tallies()
        assert N.array_equal(h, good['h']) and N.allclose(a, good['a'], rtol=1e-9, atol=1e-12), knobs
        assert N.allclose(dev.get_fluxmap(30), good['fm'], rtol=1e-9, atol=1e-12), knobs
        cap = dev.get_hits()
        assert len(cap['surf']) == good['h'][30] and N.isclose(cap['e_abs'].sum(), good['a'][30], rtol=1e-9), knobs
        dev.close()


def test_full_size_routes_agree(ctx):
    """
    configs[2] / configs[3] at their real size: 1e8 NSTTF source rays (two batches of 5e7 in flight) and the per-GPU share of
    the 8-GPU job, 1.25e8 (two batches of 6.25e7 < 2^26), through the default route; the same 1e8 on the Kd walk of the queues
    (configs[3] names accel_tree's traversal) and by brute force must give identical hit counts; flux map == receiver tally;
    the receiver's power within 3 sigma of the reference's own Monte-Carlo runs (tests/golden/mc_reference.npz).
    """
    from helpers import load
    from tracer_amd import scenes
    from tracer_amd.accel_tree import KdTree
    from tracer_amd.scene import compile_scene
    plant, field, rec, src = scenes.nsttf_field()
    cs = compile_scene(plant)
    kd = KdTree(plant, 8 + 1.3 * N.log(cs.n_surf), min_leaf=1)
    ue, ve = scenes.nsttf_fluxmap_edges()
    mc = load('mc_reference.npz')
    p_ref, se_ref = float(mc['nsttf_receiver_mean']), float(mc['nsttf_receiver_se'])
    out = {}
    for n in (100000000, 125000000):
        bundle = lambda: scenes.nsttf_source(n, src, seed=2024, ray_offset=7 * n)
        res = _trace(ctx, cs, bundle, kd=kd, fluxmap=(218, ue, ve), reps=100, stream=None)
        out[n] = res
        e_ray = 1000. * N.pi * src['radius'] ** 2 / n
        assert res['launches'] > 1                                                  # the streaming form, by default at this size
        assert n <= res['segments'] <= n + res['h'][:218].sum()
        assert N.isclose(res['fm'].sum(), res['a'][218], rtol=1e-9)
        # receiver hits carry 0.96 e_ray (one mirror) -- at most e_ray: the standard error of the sum from its own count
        se_gpu = e_ray * N.sqrt(res['h'][218])
        assert abs(res['a'][218] - p_ref) <= 3. * N.sqrt(se_gpu ** 2 + se_ref ** 2), (n, res['a'][218], p_ref)
    n = 100000000
    bundle = lambda: scenes.nsttf_source(n, src, seed=2024, ray_offset=7 * n)
    _same(out[n], _trace(ctx, cs, bundle, kd=kd, fluxmap=(218, ue, ve), reps=100, stream=None, TRC_STREAM_SEARCH=1, TRC_STREAM_FRESH=0), 'Kd walk, 1e8')
    _same(out[n], _trace(ctx, cs, bundle, kd=kd, fluxmap=(218, ue, ve), reps=100, stream=None, accel=False), 'brute force, 1e8')


@pytest.mark.parametrize('world', [2, 4])
def test_bench_as_two_ranks_on_one_gpu(ctx, world):
    """
    (world 2 and 4: the GPU box allows six processes on its card, and this one is the fifth; eight ranks are rehearsed on the CPU,
    tests/test_distributed_cpu.py)
    bench.py under torch.distributed.run with two ranks sharing this GPU (TRC_BENCH_BACKEND=gloo: the tallies cross ranks through
    the host; on an 8-GPU node the same code path uses RCCL): rays sharded by stream id, one reduction of the packed tally buffer.
    The line it prints must hold the sum of both ranks -- hits and segments equal to the same four batches traced in this
    process, `value` counting both ranks' segments over the slowest rank's time.
    """
    import json, subprocess, sys
    from tracer_amd import scenes
    from tracer_amd.accel_tree import KdTree
    from tracer_amd.scene import compile_scene, DeviceScene
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n, steps, warmup = 3000000, 2, 1
    port = 29600 + os.getpid() % 300 + world
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(root, 'bench.py'), '--gpus', str(world), '--steps', str(steps), '--warmup', str(warmup),
           '--rays', str(n), '--cpu-rays', '0', '--api-steps', '0']
    envv = dict(os.environ, TRC_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    p = subprocess.run(cmd, cwd=root, env=envv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    line = [l for l in p.stdout.decode().splitlines() if l.startswith('{')][-1]
    out = json.loads(line)
    assert out['n_gpus'] == world and out['steps'] == steps and out['scaling'] == 'weak' and out['check']['ok'] is True
    # the same batches here: stream ids (step * world + rank) * n, seed 2024 (bench.py's)
    plant, field, rec, src = scenes.nsttf_field()
    cs = compile_scene(plant)
    dev = DeviceScene(cs, ctx)
    dev.set_kdtree(KdTree(plant, 8 + 1.3 * N.log(cs.n_surf), min_leaf=1))
    seg = 0
    for k in range(warmup, warmup + steps):
        for rank in range(world):
            st, _ = dev.trace_fast(scenes.nsttf_source(n, src, seed=2024, ray_offset=(k * world + rank) * n), 100, 1e-10, 2024, accel=True)
            seg += st.segments
    a, r, h = dev.get_tallies()
    dev.close()
    c = out['check']
    assert c['receiver_hits'] == h[218] and c['heliostat_hits'] == h[:218].sum() and c['segments_total'] == seg
    assert N.isclose(c['receiver_kW'], a[218] / (steps * world) / 1e3, rtol=1e-9)
    assert N.isclose(out['value'], seg / (out['ms_per_step'] * 1e-3 * steps) / 1e6, rtol=1e-6)


def test_bench_strong_scaling_gives_the_same_tallies_for_every_world_size(ctx):
    """
    bench.py --scaling strong: the ranks share the rays of a step (distributed.shard over the stream ids), so the job's
    tallies do not depend on the number of ranks -- hit counts and segments to the last ray, energies to rounding (the
    reference's multi-process driver, tracer_engine_mp.py:19-124, merges independent bundles; here one bundle is split).
    """
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n, steps, warmup = 4000001, 2, 1
    outs = {}
    for world in (1, 2, 4):
        port = 29900 + os.getpid() % 300 + world
        tail = [os.path.join(root, 'bench.py'), '--gpus', str(world), '--steps', str(steps), '--warmup', str(warmup), '--rays', str(n),
                '--cpu-rays', '0', '--api-steps', '0', '--scaling', 'strong']
        cmd = [sys.executable] + tail if world == 1 else \
            [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
             '--master-port', str(port)] + tail
        envv = dict(os.environ, TRC_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
        p = subprocess.run(cmd, cwd=root, env=envv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        outs[world] = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith('{')][-1])
        assert outs[world]['scaling'] == 'strong' and outs[world]['n_gpus'] == world and outs[world]['check']['ok'] is True
    c1 = outs[1]['check']
    for world in (2, 4):
        c = outs[world]['check']
        for key in ('receiver_hits', 'heliostat_hits', 'segments_total'):
            assert c[key] == c1[key], (world, key)
        assert N.isclose(c['receiver_kW'], c1['receiver_kW'], rtol=1e-12) and N.isclose(c['fluxmap_sum_kW'], c1['fluxmap_sum_kW'], rtol=1e-12)


def test_forms_of_the_fast_engine_end_every_ray_alike(ctx):
    """
    The megakernel on brute force and the streaming kernels on the grid, 1e7 rays of a pillbox disc into 150 plates, discs,
    spheres, hemispheres, cylinders and dishes with and without slope error -- thrown at random, and on integer positions with
    quarter turns: identical hit counts on every surface and the same number of segments.  This holds to the last ray because
    the library is built with -ffp-contract=off (Makefile): fused, a*b+c rounds differently in each kernel a formula is inlined
    into and about one ray in 1e7 ends on another surface (measured: 58 of 150 surfaces with other counts at 2e7 rays).
    """
    from tracer_amd.assembly import Assembly
    from tracer_amd.object import AssembledObject
    from tracer_amd.surface import Surface
    from tracer_amd.flat_surface import RectPlateGM, RoundPlateGM
    from tracer_amd.sphere_surface import SphericalGM, HemisphereGM
    from tracer_amd.cylinder import FiniteCylinder
    from tracer_amd.paraboloid import ParabolicDishGM
    from tracer_amd.optics_callables import Reflective, RealReflective
    from tracer_amd.tracer_engine import TracerEngine
    from tracer_amd.sources import disk_bundle
    from tracer_amd.spatial_geometry import generate_transform
    n = 10000000
    sun = N.r_[0.2, -0.1, -1.] / N.linalg.norm([0.2, -0.1, -1.])
    for aligned in (False, True):
        rng = N.random.RandomState(5 + aligned)
        objs = []
        for _ in range(150):
            kind, s = rng.randint(0, 6), rng.uniform(0.2, 1.5)
            gm = (RectPlateGM(2 * s, s), RoundPlateGM(s), SphericalGM(s), HemisphereGM(s), FiniteCylinder(2 * s, 3 * s),
                  ParabolicDishGM(2 * s, rng.uniform(0.5, 2.)))[kind]
            o = AssembledObject(surfs=[Surface(gm, RealReflective(0.2, 2e-3) if kind % 2 else Reflective(0.2))])
            loc = rng.uniform(-6., 6., 3)
            if aligned:
                o.set_transform(generate_transform(N.r_[1., 0, 0], rng.choice([0., N.pi / 2, N.pi]), N.round(loc)[:, None]))
            else:
                ax = rng.normal(size=3)
                o.set_transform(generate_transform(ax / N.linalg.norm(ax), rng.uniform(0, 2 * N.pi), loc[:, None]))
            objs.append(o)
        eng = TracerEngine(Assembly(objects=objs))
        seen = {}
        for form, accel in (('megakernel', False), ('stream', True)):
            eng.reset_tallies()
            eng.ray_tracer(disk_bundle(n, N.c_[-12. * sun], sun, 9., 4.65e-3, flux=1., seed=9), 50, 1e-8, accel=accel, seed=9,
                           tree=False, fast_kernel=form)
            a, r, h = eng.get_tallies()
            seen[form] = (a.copy(), h.copy(), eng.stats['segments'])
        (a0, h0, s0), (a1, h1, s1) = seen['megakernel'], seen['stream']
        assert h0.sum() > n and (h0 > 0).sum() > 100
        assert s1 == s0 and N.array_equal(h1, h0), aligned
        assert N.allclose(a1, a0, rtol=1e-9, atol=1e-12), aligned


def test_hits_of_several_capturing_surfaces_come_grouped(ctx):
    """
    trc_scene_get_hits with more than one capturing surface (the duct walls and the plate of the minidish): the hits come surface
    by surface, so that the accountants are fed slices; counts and absorbed energy per surface are those of the tallies, in both
    forms of the fast engine, and the accountants see what the ordered engine's accountants see.
    """
    import math
    from tracer_amd.models.tau_minidish import MiniDish
    from tracer_amd.sources import solar_disk_bundle
    from tracer_amd.spatial_geometry import rotx
    from tracer_amd.tracer_engine import TracerEngine
    x = -1 / math.sqrt(2)
    n = 600000
    per_form = {}
    for form in ('megakernel', 'stream', 'ordered'):
        dish = MiniDish(5., 6.25, 0.9, 6.95, 0.4, 0.7, 0.9)
        dish.set_transform(rotx(-N.pi / 4))
        eng = TracerEngine(dish)
        sun = solar_disk_bundle(n, N.c_[[0, 7., 7.]], N.array([0, x, x]), 3., 0.005, flux=1000., seed=6)
        if form == 'ordered':
            eng.ray_tracer(sun, 100, 1e-6, seed=6)
        else:
            eng.ray_tracer(sun, 100, 1e-6, seed=6, tree=False, fast_kernel=form)
            h = eng._dev.get_hits()
            a, r, cnt = eng.get_tallies()
            assert (N.diff(h['surf']) >= 0).all() and len(N.unique(h['surf'])) == 5
            assert N.array_equal(N.bincount(h['surf'], minlength=len(cnt))[:5], cnt[:5])          # every surface but the dish captures
            assert N.allclose(N.bincount(h['surf'], weights=h['e_abs'], minlength=len(a))[:5], a[:5], rtol=1e-9)
        per_form[form] = [s.get_optics_manager().get_all_hits() for s in dish.get_surfaces()[:5]]
    for k in range(5):
        e0, p0 = per_form['ordered'][k][:2]
        for form in ('megakernel', 'stream'):
            e1, p1 = per_form[form][k][:2]
            assert len(e1) == len(e0) and abs(e1.sum() - e0.sum()) < 1e-9 * e0.sum()
            assert N.allclose(N.sort(p1[0]), N.sort(p0[0]), atol=1e-9)          # the same hit points, in another order


def test_auto_form_settles_on_the_faster_one_for_a_dense_scene(ctx):
    """
    fast_kernel='auto' on a scene where the streaming form is slow (200 overlapping curved shapes, every segment a hit): the first
    large call streams, the second tries the megakernel, the following ones keep whichever was faster -- and every call ends its
    rays alike.  A heliostat field never leaves the streaming form.
    """
    from tracer_amd import scenes
    from tracer_amd.assembly import Assembly
    from tracer_amd.object import AssembledObject
    from tracer_amd.surface import Surface
    from tracer_amd.flat_surface import RectPlateGM, RoundPlateGM
    from tracer_amd.sphere_surface import SphericalGM, HemisphereGM
    from tracer_amd.cylinder import FiniteCylinder
    from tracer_amd.paraboloid import ParabolicDishGM
    from tracer_amd.optics_callables import Reflective
    from tracer_amd.tracer_engine import TracerEngine
    from tracer_amd.sources import disk_bundle
    from tracer_amd.spatial_geometry import generate_transform
    rng = N.random.RandomState(8)
    objs = []
    for _ in range(200):
        kind, s = rng.randint(0, 6), rng.uniform(0.2, 1.5)
        gm = (RectPlateGM(2 * s, s), RoundPlateGM(s), SphericalGM(s), HemisphereGM(s), FiniteCylinder(2 * s, 3 * s),
              ParabolicDishGM(2 * s, rng.uniform(0.5, 2.)))[kind]
        o = AssembledObject(surfs=[Surface(gm, Reflective(0.2))])
        ax = rng.normal(size=3)
        o.set_transform(generate_transform(ax / N.linalg.norm(ax), rng.uniform(0, 2 * N.pi), rng.uniform(-6., 6., 3)[:, None]))
        objs.append(o)
    eng = TracerEngine(Assembly(objects=objs))
    sun = N.r_[0.2, -0.1, -1.] / N.linalg.norm([0.2, -0.1, -1.])
    n = 4000000
    forms, counts, rates = [], [], []
    for call in range(4):
        eng.reset_tallies()
        eng.ray_tracer(disk_bundle(n, N.c_[-12. * sun], sun, 9., 4.65e-3, flux=1., seed=2), 50, 1e-8, seed=2, tree=False)
        forms.append(eng.stats['form'])
        rates.append(eng.stats['segments'] / eng.stats['kernel_ms'])
        counts.append(eng.get_tallies()[2].copy())
    assert forms[:2] == ['stream', 'megakernel'] and forms[2] == forms[3]
    assert forms[2] == ('megakernel' if rates[1] > rates[0] else 'stream')
    assert all(N.array_equal(c, counts[0]) for c in counts[1:]) and counts[0].sum() > n
    plant, field, rec, src = scenes.nsttf_field(n_heliostats=40)
    eng = TracerEngine(plant)
    for call in range(3):
        eng.ray_tracer(scenes.nsttf_source(n, src, seed=3), reps=100, min_energy=1e-10, tree=False, accel=True, seed=3)
        assert eng.stats['form'] == 'stream'


def test_accel_keyword_forms_through_ray_tracer(ctx):
    """
    ray_tracer(accel=...) as the reference spells it (tracer_engine.py:171-185): False, True, 'fast' (KdTree with at most 12
    candidate planes per axis, accel_tree.py:118) and 'lightweight' (the reference's per-leaf sequencing, :66-122 -- a
    scheduling artefact the device traversal subsumes).  All four give the same tallies and the same recorded tree through
    both engines; the engine keeps the Kd-tree it built (engine.Kd_Tree) with the reference's node count for the form asked.
    """
    from tracer_amd import scenes
    from tracer_amd.tracer_engine import TracerEngine
    from tracer_amd.ray_bundle import RayBundle
    n = 300000
    res = {}
    for accel in (False, True, 'fast', 'lightweight'):
        plant, field, rec, src = scenes.nsttf_field(n_heliostats=40)
        eng = TracerEngine(plant)
        eng.ray_tracer(scenes.nsttf_source(n, src, seed=3), reps=100, min_energy=1e-10, tree=False, accel=accel, seed=3)
        a, r, h = eng.get_tallies()
        assert eng.stats['engine'] == 'fast' and len(eng.tree._bunds) == 1 and eng.tree._bunds[-1].get_num_rays() == eng.stats['rays_left']
        kd_nodes = None if not accel else len(eng.Kd_Tree.flat()['flag'])
        # the ordered engine (tree=True) on a materialised bundle of the same rays: the recorded levels
        lazy = scenes.nsttf_source(20000, src, seed=3)
        b = RayBundle(vertices=lazy.get_vertices(), directions=lazy.get_directions(), energy=lazy.get_energy())
        plant.reset_all_optics(); eng.reset_tallies()
        eng.ray_tracer(b, reps=100, min_energy=1e-10, tree=True, accel=accel, seed=3)
        levels = [(bd.get_vertices().copy(), bd.get_energy().copy(), N.asarray(bd.get_parents()).copy()) for bd in eng.tree._bunds[1:]]
        res[accel] = (a.copy(), h.copy(), eng.stats['segments'], levels, kd_nodes)
    a0, h0, s0, lv0, _ = res[False]
    assert h0[40] > 0.05 * n * 40 / 218. and len(lv0) >= 2
    for accel in (True, 'fast', 'lightweight'):
        a1, h1, s1, lv1, nodes = res[accel]
        assert N.array_equal(h1, h0) and s1 == s0 and N.allclose(a1, a0, rtol=1e-10), accel
        assert len(lv1) == len(lv0), accel
        for (v1, e1, p1), (v0, e0, p0) in zip(lv1, lv0):
            assert N.array_equal(p1, p0) and N.array_equal(v1, v0) and N.array_equal(e1, e0), accel
        assert nodes is not None and nodes > 40
    assert res['lightweight'][4] == res[True][4]           # the same tree as accel=True; 'fast' may split elsewhere


def _height_field(m, extent=10., amp=0.8):
    """m x m quads over [-extent, extent]^2 on a relief steep enough for second and third bounces, two triangles each:
    vertices (n, 3), faces (2 m^2, 3)"""
    x, y = N.meshgrid(N.linspace(-extent, extent, m + 1), N.linspace(-extent, extent, m + 1), indexing='ij')
    z = amp * N.sin(1.3 * x) * N.cos(1.1 * y)
    V = N.c_[x.ravel(), y.ravel(), z.ravel()]
    i, j = N.meshgrid(N.arange(m), N.arange(m), indexing='ij')
    a, b, c, d = (i * (m + 1) + j).ravel(), ((i + 1) * (m + 1) + j).ravel(), ((i + 1) * (m + 1) + j + 1).ravel(), (i * (m + 1) + j + 1).ravel()
    return V, N.vstack((N.c_[a, b, c], N.c_[a, c, d]))


def test_mesh_of_1e5_triangles(ctx, tmp_path):
    """
    SURVEY 8(f)4: meshes as the reference builds them -- one Surface per face (models/triangulated_surface.py:12-52,
    ray_trace_utils/stl_utils.py:178-235) -- at a size where nothing fits LDS: 105 800 triangles + a receiver.  The device keeps
    records, boxes, the 32-bit uniform grid and the footprint lists in global memory (L2-resident) and searches with k_s_bounce
    for every ray.  Checked: (1) a 10 082-triangle mesh against brute force over all boxes (streaming form and megakernel) and
    against the oracle ray by ray on the same Philox streams; (2) the 105 800-triangle mesh: footprint route == the per-ray route
    for all fresh rays == given bundle; energy conservation; (3) the same mesh written to STL and loaded with
    load_stl_into_tracer gives the same totals (float32 vertices in the file: 1e-3).
    """
    from tracer_amd import sources, stl_utils
    from tracer_amd.assembly import Assembly
    from tracer_amd.object import AssembledObject
    from tracer_amd.surface import Surface
    from tracer_amd.flat_surface import RectPlateGM
    from tracer_amd.models.triangulated_surface import TriangulatedSurface
    from tracer_amd import optics_callables as opt
    from tracer_amd.spatial_geometry import translate, rotx
    from tracer_amd.scene import compile_scene
    from tracer_amd.ray_bundle import RayBundle
    from tracer_amd.tracer_engine import TracerEngine
    direction = N.r_[0.1, -0.05, -1.] / N.linalg.norm([0.1, -0.05, -1.])
    center = N.c_[-40. * direction]

    def scene(m):
        V, F = _height_field(m)
        mesh = TriangulatedSurface(V, F, opt.Reflective(0.2))
        lid = AssembledObject(surfs=[Surface(RectPlateGM(90., 90.), opt.LambertianReceiver(1.))], transform=N.dot(translate(0., 0., 50.), rotx(N.pi)))
        return Assembly(objects=[mesh, lid]), len(F)

    # (1) 1e4 triangles: every route, brute force, the oracle
    asm, nf = scene(71)
    assert nf == 10082
    cs = compile_scene(asm)
    n = 400000
    bundle = lambda: sources.buie_sunshape(n, center, direction, 12., 0.05, flux=1., seed=21)
    ref = _trace(ctx, cs, bundle, reps=6)
    assert ref['hits'] > 0.5 * n and ref['h'][nf] > 0.15 * n and ref['segments'] > 1.8 * n
    # (a mesh that fills the source's view gets no footprint map by default -- it would cull nothing: asking for a resolution forces it)
    _same(ref, _trace(ctx, cs, bundle, reps=6, TRC_STREAM_FP_CELLS=512), 'footprint map, lists in global memory')
    _same(ref, _trace(ctx, cs, bundle, reps=6, TRC_STREAM_FRESH=0), 'fresh rays through k_s_bounce<FRESH> / the general path')
    _same(ref, _trace(ctx, cs, bundle, reps=6, TRC_STREAM_FIRST=1), 'aureole through k_s_bounce<FRESH>')
    _same(ref, _trace(ctx, cs, bundle, reps=6, TRC_STREAM_FRESH=0, TRC_STREAM_FIRST=1), 'all fresh rays through k_s_bounce<FRESH>')
    _same(ref, _trace(ctx, cs, bundle, reps=6, TRC_STREAM_COOP=0), 'a lane per ray (k_s_bounce<2>) against the shared tests of k_s_bounce_coop')
    small = lambda: sources.buie_sunshape(20000, center, direction, 12., 0.05, flux=1., seed=22)
    brute = _trace(ctx, cs, small, reps=6, accel=False, TRC_STREAM_FRESH=0, TRC_STREAM_BOUNCE=0)
    _same(brute, _trace(ctx, cs, small, reps=6), 'default route vs all boxes')
    _same(brute, _trace(ctx, cs, small, reps=6, stream=False, accel=False), 'megakernel, brute force')
    from oracle import engine as oracle_engine
    with N.errstate(all='ignore'):
        o = oracle_engine.trace_from_compiled(cs, small().source_args(), reps=6, min_energy=1e-10)
    assert N.array_equal(o['hits'], brute['h']) and o['segments'] == brute['segments']
    assert N.allclose(o['absorbed'], brute['a'], rtol=1e-9, atol=1e-12)

    # (2) 1e5 triangles through the public entry point
    asm, nf = scene(230)
    assert nf == 105800
    eng = TracerEngine(asm)
    n = 2000000
    out = {}
    for key, knobs in (('map', dict(TRC_STREAM_FP_CELLS=1024)), ('per ray', {})):
        with env(**knobs):
            eng.reset_tallies(); asm.reset_all_optics()
            eng.ray_tracer(sources.buie_sunshape(n, center, direction, 12., 0.05, flux=1., seed=23), reps=6, min_energy=1e-10, tree=False, accel=True, seed=23)
            a, r, h = eng.get_tallies()
            out[key] = (a.copy(), r.copy(), h.copy(), eng.stats['segments'])
    assert eng.Kd_Tree is None and eng.stats['launches'] > 1
    a0, r0, h0, s0 = out['map']
    a1, r1, h1, s1 = out['per ray']
    assert N.array_equal(h0, h1) and s0 == s1 and N.allclose(a0, a1, rtol=1e-9, atol=1e-15)
    assert h0[:nf].sum() > 0.5 * n and h0[nf] > 0.15 * n and (h0[:nf] > 0).sum() > 0.4 * nf
    lazy = sources.buie_sunshape(300000, center, direction, 12., 0.05, flux=1., seed=24)
    eng.reset_tallies(); asm.reset_all_optics()
    eng.ray_tracer(sources.buie_sunshape(300000, center, direction, 12., 0.05, flux=1., seed=24), reps=6, min_energy=1e-10, tree=False, accel=True, seed=24)
    a2, r2, h2 = [x.copy() for x in eng.get_tallies()]
    given = RayBundle(vertices=lazy.get_vertices(), directions=lazy.get_directions(), energy=lazy.get_energy())
    eng.reset_tallies(); asm.reset_all_optics()
    eng.ray_tracer(given, reps=6, min_energy=1e-10, tree=False, accel=True, seed=24)
    a3, r3, h3 = eng.get_tallies()
    assert N.array_equal(h3, h2) and N.allclose(a3, a2, rtol=1e-9, atol=1e-15)
    # energy: what the source sent lands on the mesh (20 % absorbed per touch) or on the lid, or leaves sideways
    e_src = N.pi * 12. ** 2
    assert a0.sum() <= e_src * (1 + 1e-9) and a0.sum() > 0.3 * e_src

    # (3) through an STL file
    V, F = _height_field(230)
    path = str(tmp_path / 'relief.stl')
    stl_utils.make_stl(V, F, path)
    obj = stl_utils.load_stl_into_tracer(path, opt.Reflective, dict(absorptivity=0.2), option='triangle')
    assert len(obj.get_surfaces()) == nf
    lid = AssembledObject(surfs=[Surface(RectPlateGM(90., 90.), opt.LambertianReceiver(1.))], transform=N.dot(translate(0., 0., 50.), rotx(N.pi)))
    eng2 = TracerEngine(Assembly(objects=[obj, lid]))
    eng2.ray_tracer(sources.buie_sunshape(n, center, direction, 12., 0.05, flux=1., seed=23), reps=6, min_energy=1e-10, tree=False, accel=True, seed=23)
    a4, r4, h4 = eng2.get_tallies()
    assert abs(a4[:nf].sum() / a0[:nf].sum() - 1.) < 1e-3 and abs(a4[nf] / a0[nf] - 1.) < 1e-3


def test_large_grid_with_plates_and_a_closed_mesh(ctx):
    """
    The large grid (scenes beyond what LDS holds) with what the relief of test_mesh_of_1e5_triangles does not have: surfaces that
    are not triangles -- listed with their boxes instead of corner and edges -- and a closed mesh, whose far side a ray that runs a
    stage ahead of its exact tests (k_s_bounce_coop) meets too.  12 000 small mirror plates in random poses over a sphere of 20 480
    triangles, a black floor: the shared tests of k_s_bounce_coop == a lane per ray (k_s_bounce<2>) == every box tested (accel=False)
    == the megakernel on a sample == the oracle on the same Philox streams.
    """
    from tracer_amd import sources
    from tracer_amd.assembly import Assembly
    from tracer_amd.object import AssembledObject
    from tracer_amd.surface import Surface
    from tracer_amd.flat_surface import RectPlateGM
    from tracer_amd.models.triangulated_surface import TriangulatedSurface
    from tracer_amd import optics_callables as opt
    from tracer_amd.spatial_geometry import translate, general_axis_rotation
    from tracer_amd.scene import compile_scene
    from oracle import engine as oracle_engine
    rng = N.random.default_rng(12)
    # an icosphere: 20 faces split five times
    t = (1. + 5. ** 0.5) / 2.
    V = N.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t], [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], float)
    F = N.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
                 [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]])
    V /= N.linalg.norm(V, axis=1)[:, None]
    for _ in range(5):
        mid = {}
        V = list(V)
        def m(a, b):
            k = (min(a, b), max(a, b))
            if k not in mid:
                p = (V[a] + V[b]) / 2.
                V.append(p / N.linalg.norm(p))
                mid[k] = len(V) - 1
            return mid[k]
        F = N.array([f for a, b, c in F for f in ([a, m(a, b), m(c, a)], [b, m(b, c), m(a, b)], [c, m(c, a), m(b, c)], [m(a, b), m(b, c), m(c, a)])])
        V = N.array(V)
    assert len(F) == 20480
    ball = TriangulatedSurface(3. * V, F, opt.Reflective(0.3), transform=translate(0., 0., 4.))
    plates = []
    for k in range(12000):
        axis = rng.normal(size=3)
        rot = general_axis_rotation(axis / N.linalg.norm(axis), rng.uniform(0., N.pi))
        tr = N.eye(4)
        tr[:3, :3] = rot
        tr[:3, 3] = (rng.uniform(-9., 9.), rng.uniform(-9., 9.), rng.uniform(8., 14.))
        plates.append(AssembledObject(surfs=[Surface(RectPlateGM(0.25, 0.15), opt.Reflective(0.1))], transform=tr))
    floor = AssembledObject(surfs=[Surface(RectPlateGM(60., 60.), opt.LambertianReceiver(1.))], transform=translate(0., 0., -0.5))
    asm = Assembly(objects=plates + [ball, floor])
    cs = compile_scene(asm)
    assert cs.n_surf == 12000 + 20480 + 1
    direction = N.r_[0.05, 0.1, -1.] / N.linalg.norm([0.05, 0.1, -1.])
    n = 300000
    bundle = lambda: sources.buie_sunshape(n, N.c_[-30. * direction + N.r_[0., 0., 4.]], direction, 10., 0.05, flux=1., seed=5)
    ref = _trace(ctx, cs, bundle, reps=8)
    assert ref['launches'] > 1 and ref['h'][:12000].sum() > 0.03 * n and ref['h'][12000:-1].sum() > 0.05 * n and ref['h'][-1] > 0.3 * n
    assert ref['segments'] > 1.1 * n
    _same(ref, _trace(ctx, cs, bundle, reps=8, TRC_STREAM_COOP=0), 'a lane per ray against the shared tests')
    small = lambda: sources.buie_sunshape(15000, N.c_[-30. * direction + N.r_[0., 0., 4.]], direction, 10., 0.05, flux=1., seed=6)
    brute = _trace(ctx, cs, small, reps=8, accel=False, TRC_STREAM_FRESH=0, TRC_STREAM_BOUNCE=0)
    _same(brute, _trace(ctx, cs, small, reps=8), 'the large grid against every box')
    _same(brute, _trace(ctx, cs, small, reps=8, TRC_STREAM_COOP=0), 'the large grid, a lane per ray, against every box')
    _same(brute, _trace(ctx, cs, small, reps=8, stream=False, accel=False), 'megakernel, brute force')
    with N.errstate(all='ignore'):
        o = oracle_engine.trace_from_compiled(cs, small().source_args(), reps=8, min_energy=1e-10)
    assert N.array_equal(o['hits'], brute['h']) and o['segments'] == brute['segments']
    assert N.allclose(o['absorbed'], brute['a'], rtol=1e-9, atol=1e-12)
    # the ordered engine (tree=True, the reference's default) walks the large grid too (trc_nearest_grid32): the tree equals the
    # oracle's level by level, the tallies those of the fast engine
    from tracer_amd.tracer_engine import TracerEngine
    eng = TracerEngine(asm)
    eng.ray_tracer(small(), reps=8, min_energy=1e-10, tree=True, accel=True, seed=6)
    assert eng.stats['engine'] == 'ordered' and eng.Kd_Tree is None
    a, r, h = eng.get_tallies()
    assert N.array_equal(h, brute['h']) and N.allclose(a, brute['a'], rtol=1e-9, atol=1e-12)
    assert eng.tree.num_bunds() == len(o['levels'])
    for k in range(1, eng.tree.num_bunds()):
        B, Lo = eng.tree[k], o['levels'][k]
        assert N.array_equal(B.get_parents(), Lo['parents']), k
        assert N.allclose(B.get_vertices(), Lo['vertices'], rtol=1e-9, atol=1e-9) and N.allclose(B.get_energy(), Lo['energy'], rtol=1e-9, atol=1e-15), k


def test_scattering_slab_vs_oracle(ctx):
    """
    SURVEY 8(f)2, participating media: RefractiveScatteringHomogenous (optics_callables.py:1350-1376 on Scattering :946-1036,
    Henyey-Greenstein sampling.py:150-168).  A slab of a scattering glass (n = 1.5, s_c = 2.5 / m, g = 0.6) in air between two
    large plates, a black floor under it, rays from above:
      * fast engine (streaming form and megakernel) == oracle on the same Philox streams: per-surface hit counts, interactions
        (surface hits + scattering events) and segments exactly, energies to 1e-9; scattered rays leave no trace on the surfaces;
      * ordered engine == oracle level by level (vertices of the scattering events inside the slab, directions, parents);
      * known answer: of the rays refracted into the slab at normal incidence, the share that crosses its 0.8 m unscattered is
        exp(-s_c L) (3 sigma).
    """
    from tracer_amd import sources, optics_callables as opt
    from tracer_amd.assembly import Assembly
    from tracer_amd.object import AssembledObject
    from tracer_amd.surface import Surface
    from tracer_amd.flat_surface import RectPlateGM
    from tracer_amd.spatial_geometry import translate
    from tracer_amd.scene import compile_scene, DeviceScene
    from tracer_amd.ray_bundle import RayBundle
    from oracle import engine as oracle_engine
    s_c, g, L = 2.5, 0.6, 0.8
    mk = lambda: opt.RefractiveScatteringHomogenous(1., 1.5, 0., s_c, 0., g)
    top = AssembledObject(surfs=[Surface(RectPlateGM(40., 40.), mk())], transform=translate(0., 0., L))
    bottom = AssembledObject(surfs=[Surface(RectPlateGM(40., 40.), mk())], transform=translate(0., 0., 0.))
    floor = AssembledObject(surfs=[Surface(RectPlateGM(60., 60.), opt.LambertianReceiver(1.))], transform=translate(0., 0., -1.))
    cs = compile_scene(Assembly(objects=[top, bottom, floor]))
    n = 200000
    rng = N.random.RandomState(4)
    v = N.vstack((rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), N.full(n, 3.)))
    d = N.tile(N.c_[[0., 0., -1.]], (1, n))
    e = N.ones(n) / n
    reps = 30
    with N.errstate(all='ignore'):
        o = oracle_engine.trace_bundle(cs, v, d, e, reps, 1e-10, 11)
    assert o['events'] > o['hits'].sum() > 2 * n            # scattering events on top of the surface hits
    for stream in (True, False):
        dev = DeviceScene(cs, ctx)
        st, _ = dev.trace_fast(RayBundle(vertices=v, directions=d, energy=e, ref_index=N.ones(n)), reps, 1e-10, 11, stream=stream)
        a, r, h = dev.get_tallies()
        dev.close()
        assert N.array_equal(h, o['hits']) and st.segments == o['segments'] and st.hits == o['events'], stream
        assert N.allclose(a, o['absorbed'], rtol=1e-9, atol=1e-12) and N.allclose(r, o['received'], rtol=1e-9, atol=1e-12), stream
    # the ordered engine's recorded tree
    m = 20000
    with N.errstate(all='ignore'):
        o2 = oracle_engine.trace_bundle(cs, v[:, :m], d[:, :m], e[:m], reps, 1e-10, 11)
    dev = DeviceScene(cs, ctx)
    res, st = dev.trace_ordered(RayBundle(vertices=v[:, :m], directions=d[:, :m], energy=e[:m], ref_index=N.ones(m)), reps, 1e-10, 11)
    levels = [res.level(k) for k in range(res.num_levels())]
    res.close(); dev.close()
    assert len(levels) == len(o2['levels'])
    inside = 0
    for k in range(1, len(levels)):
        Ld, Lo = levels[k], o2['levels'][k]
        assert N.array_equal(Ld['parents'], Lo['parents']) and N.array_equal(Ld['surf'], Lo['surf']), k
        assert N.allclose(Ld['vertices'], Lo['vertices'], rtol=1e-9, atol=1e-9) and N.allclose(Ld['directions'], Lo['directions'], rtol=1e-9, atol=1e-9), k
        assert N.allclose(Ld['energy'], Lo['energy'], rtol=1e-12)
        z = Ld['vertices'][2]
        inside += int(((z > 1e-6) & (z < L - 1e-6)).sum())
    assert inside > m                                        # scattering events start inside the slab, not on a surface
    # Beer-Lambert: level 1 = the first interaction (all at the top plate, z = L): reflected or refracted; the refracted ones
    # (directions still (0, 0, -1)) either scatter inside (level 2 vertex with 0 < z < L) or reach the bottom plate (z = 0)
    L1, L2 = o2['levels'][1], o2['levels'][2]
    went_in = N.nonzero(L1['directions'][2] < 0)[0]
    par = L2['parents']
    from_in = N.isin(par, went_in)
    crossed = from_in & (N.abs(L2['vertices'][2]) < 1e-9)
    share = crossed.sum() / float(from_in.sum())
    sigma = N.sqrt(N.exp(-s_c * L) * (1 - N.exp(-s_c * L)) / from_in.sum())
    assert abs(share - N.exp(-s_c * L)) < 3.5 * sigma, (share, N.exp(-s_c * L), sigma)
    # the levels say which rays were scattered in the medium (TRC_LEVEL_VOLUME): rays that start inside the slab, nothing else
    for k in range(1, len(levels)):
        vol = levels[k]['volume']
        z = levels[k]['vertices'][2]
        mid = (z > 1e-6) & (z < L - 1e-6)
        assert (vol is None and not mid.any()) or N.array_equal(vol, mid), k
    # accountants on the scattering plates see the hits on the plates and not the scattering events in front of them, through the
    # ordered engine (which files those rays under the plate they were heading for) as through the fast one
    from tracer_amd.tracer_engine import TracerEngine
    mk_acc = lambda: opt.RefractiveScatteringHomogenousDetector(1., 1.5, 0., s_c, 0., g)
    plates = [Surface(RectPlateGM(40., 40.), mk_acc()), Surface(RectPlateGM(40., 40.), mk_acc())]
    asm = Assembly(objects=[AssembledObject(surfs=[plates[0]], transform=translate(0., 0., L)),
                            AssembledObject(surfs=[plates[1]], transform=translate(0., 0., 0.)),
                            AssembledObject(surfs=[Surface(RectPlateGM(60., 60.), opt.LambertianReceiver(1.))], transform=translate(0., 0., -1.))])
    got = {}
    for tree in (True, False):
        asm.reset_all_optics()
        eng = TracerEngine(asm)
        eng.ray_tracer(RayBundle(vertices=v[:, :m], directions=d[:, :m], energy=e[:m], ref_index=N.ones(m)), reps, 1e-10, tree=tree, seed=11)
        assert eng.stats['engine'] == ('ordered' if tree else 'fast')
        got[tree] = [p.get_optics_manager().get_all_hits() for p in plates]
    for pi in range(2):
        (e_o, loc_o, dir_o), (e_f, loc_f, dir_f) = got[True][pi], got[False][pi]
        assert len(e_o) == len(e_f) == o2['hits'][pi], pi
        assert N.allclose(N.abs(loc_o[2] - (L if pi == 0 else 0.)), 0., atol=1e-9), "every hit the accountants hold lies on the plate"
        ko, kf = N.lexsort((loc_o[1], loc_o[0], e_o)), N.lexsort((loc_f[1], loc_f[0], e_f))
        assert N.allclose(loc_o[:, ko], loc_f[:, kf], rtol=1e-9, atol=1e-9) and N.allclose(dir_o[:, ko], dir_f[:, kf], rtol=1e-9, atol=1e-9)
