"""
SURVEY 8(f)2, rest -- what rays carry beyond position, direction and energy:
  * complex, wavelength-dependent refractive indices of tabulated materials (Refractive, optics_callables.py:726-858) and the
    attenuation that follows from their imaginary part (RefractiveAbsorbant :908-944 on Absorbant.attenuate :874-889);
  * polychromatic bundles: a spectrum per ray (`spectra` over `wavelengths`, both (W,N)), integrated by the polychromatic wall
    (Lambertian_directional_axisymmetric_piecewise_Polychromatic :393-425) and scaled by the classes that have the line
    `outg._spectra *= ...`.
Per-call parity with the reference's own outputs is in test_gpu_parity.test_optics_vs_reference_and_oracle; here whole traces through
TracerEngine.ray_tracer are compared with the oracle level by level, and with known answers.
"""
import numpy as N
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    from tracer_amd import _cabi
    return _cabi.get_context()


def _materials():
    from tracer_amd import optics_callables as opt
    tl = N.linspace(0.3e-6, 2.5e-6, 6)
    air = opt.TabulatedMaterial(tl, N.ones(6), N.zeros(6))
    glass = opt.TabulatedMaterial(tl, [1.55, 1.53, 1.51, 1.50, 1.49, 1.47], [3e-8, 2e-8, 1e-8, 5e-8, 2e-7, 6e-7])
    return air, glass


def _slab_scene(single_ray, absorb=True):
    from tracer_amd import optics_callables as opt
    from tracer_amd.assembly import Assembly
    from tracer_amd.object import AssembledObject
    from tracer_amd.surface import Surface
    from tracer_amd.flat_surface import RectPlateGM
    from tracer_amd.spatial_geometry import translate
    air, glass = _materials()
    cls = opt.RefractiveAbsorbant if absorb else opt.Refractive
    kw = dict(attenuation_coefficient_1=1.) if absorb else {}
    top = AssembledObject(surfs=[Surface(RectPlateGM(40., 40.), cls(air, glass, single_ray=single_ray, **kw))], transform=translate(0., 0., 0.5))
    bottom = AssembledObject(surfs=[Surface(RectPlateGM(40., 40.), cls(air, glass, single_ray=single_ray, **kw))], transform=translate(0., 0., 0.))
    floor = AssembledObject(surfs=[Surface(RectPlateGM(60., 60.), opt.LambertianReceiver(1.))], transform=translate(0., 0., -1.))
    return Assembly(objects=[top, bottom, floor]), air, glass


def _bundle(n, air, seed=3, tilt=0.3):
    from tracer_amd.ray_bundle import RayBundle
    rng = N.random.RandomState(seed)
    v = N.vstack((rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), N.full(n, 3.)))
    d = N.vstack((rng.uniform(-tilt, tilt, n), rng.uniform(-tilt, tilt, n), -N.ones(n)))
    d /= N.sqrt(N.sum(d ** 2, axis=0))
    wl = rng.uniform(0.4e-6, 2.4e-6, n)
    return RayBundle(vertices=v, directions=d, energy=N.ones(n) / n, ref_index=air.m(wl), wavelengths=wl), wl


def _compare_levels(tree, o, complex_index=True, spectra=False):
    for k in range(1, min(tree.num_bunds(), len(o['levels']))):
        B, Lo = tree[k], o['levels'][k]
        assert N.array_equal(B.get_parents(), Lo['parents']), k
        assert N.allclose(B.get_vertices(), Lo['vertices'], rtol=1e-9, atol=1e-9), k
        assert N.allclose(B.get_directions(), Lo['directions'], rtol=1e-9, atol=1e-9), k
        assert N.allclose(B.get_energy(), Lo['energy'], rtol=1e-9, atol=1e-15), k
        if complex_index:
            assert N.iscomplexobj(B.get_ref_index()) and N.allclose(B.get_ref_index(), Lo['ref'], rtol=1e-12, atol=0), k
        if spectra:
            assert N.allclose(B.get_spectra(), Lo['spectra'], rtol=1e-12, atol=0), k
            assert N.array_equal(B.get_wavelengths(), Lo['swl']), k
    assert tree.num_bunds() == len(o['levels'])


def test_absorbing_slab_between_tabulated_materials(ctx):
    """
    A glass slab (complex index tabulated over the wavelength) in air, rays of mixed wavelengths from above, a black floor below.
    ray_tracer picks the ordered engine by itself (the rays carry complex indices); every level of the tree equals the oracle's,
    complex indices included, for ray splitting and for one ray per hit; the energy arriving on the floor through the slab at normal
    incidence follows Fresnel twice and Beer-Lambert with 4 pi k / lambda once.
    """
    from tracer_amd.tracer_engine import TracerEngine
    from tracer_amd.scene import compile_scene
    from oracle import engine as oracle_engine
    n = 4000
    for single in (False, True):
        asm, air, glass = _slab_scene(single)
        b, wl = _bundle(n, air)
        eng = TracerEngine(asm)
        eng.ray_tracer(b, reps=6, min_energy=1e-9, tree=True, seed=21)
        assert eng.stats['engine'] == 'ordered'
        cs = compile_scene(asm)
        assert len(cs.materials) == 2 and cs.carries
        with N.errstate(all='ignore'):
            o = oracle_engine.trace_bundle(cs, b.get_vertices(), b.get_directions(), b.get_energy(), 6, 1e-9, 21,
                                           ref_index=b.get_ref_index(), wavelengths=wl)
        _compare_levels(eng.tree, o)
        # rays are in glass between the plates and nowhere else
        B2 = eng.tree[2]
        z0 = eng.tree[1].get_vertices()[2][B2.get_parents()]
        going_down_inside = (N.abs(z0 - 0.5) < 1e-9) & (eng.tree[1].get_directions()[2][B2.get_parents()] < 0) & (N.abs(B2.get_vertices()[2]) < 1e-9)
        assert going_down_inside.sum() > n // 2
    # known answer at normal incidence, one wavelength, splitting optics, no second-order paths (reps = 3: top, bottom, floor)
    asm, air, glass = _slab_scene(False)
    from tracer_amd.ray_bundle import RayBundle
    m = 64
    lam = 2.0e-6
    wl = N.full(m, lam)
    b = RayBundle(vertices=N.vstack((N.linspace(-1, 1, m), N.zeros(m), N.full(m, 3.))), directions=N.tile(N.c_[[0., 0., -1.]], (1, m)),
                  energy=N.ones(m), ref_index=air.m(wl), wavelengths=wl)
    eng = TracerEngine(asm)
    eng.ray_tracer(b, reps=3, min_energy=1e-12, tree=True, seed=5)
    mg = glass.m(N.array([lam]))[0]
    R = ((1. - mg) / (1. + mg)) ** 2
    R = R.real                                   # what the reference keeps (optics_callables.py:838-840)
    # RefractiveAbsorbant takes k from the index of the OUTGOING ray (:882): the refracted ray leaving the slab is in air (k = 0),
    # the one entering it pays for the 2.5 m of air above with the glass's k -- the reference's behaviour, reproduced.
    T_in = N.exp(-4. * N.pi * 2.5 * mg.imag / lam)
    B2 = eng.tree[2]
    leaving = (N.abs(B2.get_vertices()[2]) < 1e-9) & (B2.get_directions()[2] < 0)       # out of the slab's underside, towards the floor
    assert leaving.sum() == m and N.allclose(B2.get_energy()[leaving], (1. - R) * T_in * (1. - R), rtol=1e-12)
    assert N.allclose(B2.get_ref_index()[leaving], 1.) and T_in < 0.5
    absorbed, hits = asm.get_surfaces()[2].get_optics_manager().get_all_hits()
    assert N.allclose(N.sort(absorbed)[-m:], (1. - R) * T_in * (1. - R), rtol=1e-12) and N.allclose(hits[2], -1.)


def test_engines_refuse_what_they_do_not_carry(ctx):
    from tracer_amd.tracer_engine import TracerEngine
    from tracer_amd._cabi import TracerAmdError
    asm, air, glass = _slab_scene(True)
    b, wl = _bundle(100, air)
    eng = TracerEngine(asm)
    with pytest.raises(TracerAmdError) as err:          # the single persistent kernel of the fast engine carries neither
        eng.ray_tracer(b, reps=3, min_energy=1e-9, tree=False, engine='fast', fast_kernel='megakernel')
    assert 'trc_trace_ordered' in str(err.value)
    few = _bundle(40, air)[0]
    with pytest.raises(TracerAmdError) as err:          # ... and the streaming form starts at 64 rays
        eng.ray_tracer(few, reps=3, min_energy=1e-9, tree=False, engine='fast')
    assert 'trc_trace_ordered' in str(err.value)
    eng.ray_tracer(few, reps=3, min_energy=1e-9, tree=False)
    assert eng.stats['engine'] == 'ordered'
    from tracer_amd.ray_bundle import RayBundle
    plain = RayBundle(vertices=b.get_vertices(), directions=b.get_directions(), energy=b.get_energy(), ref_index=N.ones(100))
    with pytest.raises(ValueError):                      # no wavelengths: the materials cannot be evaluated
        eng.ray_tracer(plain, reps=3, min_energy=1e-9)


def _poly_scene():
    from tracer_amd import optics_callables as opt
    from tracer_amd.assembly import Assembly
    from tracer_amd.object import AssembledObject
    from tracer_amd.surface import Surface
    from tracer_amd.flat_surface import RectPlateGM
    from tracer_amd.spatial_geometry import translate, rotx
    ths = N.linspace(0., N.pi / 2., 7)
    wls = N.linspace(0.25e-6, 2.6e-6, 5)
    grid = 0.2 + 0.7 * N.outer(N.cos(ths) ** 0.5, 1. / (1. + (wls * 1e6 - 1.) ** 2))
    wall = lambda: opt.Lambertian_directional_axisymmetric_piecewise_PolychromaticAbsorberPolychromatic(ths, grid, wls)
    floor = AssembledObject(surfs=[Surface(RectPlateGM(4., 4.), wall())], transform=translate(0., 0., 0.))
    roof = AssembledObject(surfs=[Surface(RectPlateGM(4., 4.), opt.Reflective(0.1))], transform=N.dot(translate(0., 0., 1.5), rotx(N.pi)))
    side = AssembledObject(surfs=[Surface(RectPlateGM(4., 1.5), opt.Lambertian(0.3))], transform=N.dot(translate(0., 2., 0.75), rotx(N.pi / 2.)))
    return Assembly(objects=[floor, roof, side]), (ths, wls, grid)


def test_polychromatic_bundle_through_a_cavity(ctx):
    """
    Rays carrying spectra bounce between a polychromatic wall (floor), a mirror (roof) and a diffuse side wall.  Every level of the
    ordered engine's tree equals the oracle's: spectra (W,n), their wavelength grids, energies.  Energy bookkeeping: after the
    polychromatic wall a ray's energy IS the integral of its spectrum; the mirror scales both by 0.9; the accountant of the wall
    collects incident minus outgoing spectra.
    """
    from tracer_amd.tracer_engine import TracerEngine
    from tracer_amd.scene import compile_scene
    from tracer_amd.ray_bundle import RayBundle
    from oracle import engine as oracle_engine
    asm, (ths, wls, grid) = _poly_scene()
    n, W = 3000, 9
    rng = N.random.RandomState(8)
    v = N.vstack((rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), N.full(n, 1.)))
    d = N.vstack((rng.uniform(-0.4, 0.4, n), rng.uniform(-0.4, 0.4, n), -N.ones(n)))
    d /= N.sqrt(N.sum(d ** 2, axis=0))
    swl = N.sort(rng.uniform(0.3e-6, 2.5e-6, size=(W, n)), axis=0)
    spec = rng.uniform(0.5, 2., size=(W, n)) * 1e6
    e = N.trapezoid(spec, swl, axis=0)
    b = RayBundle(vertices=v, directions=d, energy=e, spectra=spec, wavelengths=swl)
    eng = TracerEngine(asm)
    eng.ray_tracer(b, reps=5, min_energy=1e-9, tree=True, seed=33)
    assert eng.stats['engine'] == 'ordered'
    cs = compile_scene(asm)
    with N.errstate(all='ignore'):
        o = oracle_engine.trace_bundle(cs, v, d, e, 5, 1e-9, 33, wavelengths=swl, spectra=spec)
    _compare_levels(eng.tree, o, complex_index=False, spectra=True)
    # level 1: every ray met the floor; its energy is the integral of the spectrum it leaves with
    B1 = eng.tree[1]
    assert B1.get_num_rays() == n
    assert N.allclose(B1.get_energy(), N.trapezoid(B1.get_spectra(), B1.get_wavelengths(), axis=0), rtol=1e-12)
    assert (B1.get_spectra() < spec).all()
    # the mirror keeps the ratio
    B2 = eng.tree[2]
    on_roof = N.abs(B2.get_vertices()[2] - 1.5) < 1e-9
    assert on_roof.sum() > 100
    assert N.allclose(B2.get_spectra()[:, on_roof], 0.9 * B1.get_spectra()[:, B2.get_parents()[on_roof]], rtol=1e-12)
    assert N.allclose(B2.get_energy()[on_roof], 0.9 * B1.get_energy()[B2.get_parents()[on_roof]], rtol=1e-12)
    # accountants of the wall: absorbed energy per hit, then (wavelengths, absorbed spectra) -- the canonical order
    floor_opt = asm.get_surfaces()[0].get_optics_manager()
    absorbed, (hw, hs) = floor_opt.get_all_hits()
    assert hs.shape[0] == W and hs.shape == hw.shape and hs.shape[1] == len(absorbed) >= n
    assert N.allclose(hs[:, :n], spec - B1.get_spectra(), rtol=1e-12)
    assert N.allclose(absorbed[:n], e - B1.get_energy(), rtol=1e-9)

    # the per-surface protocol (the reference's own loop over Surface objects, optics still on the device) carries spectra too:
    # the first interaction is deterministic in energy and spectrum
    eng2 = TracerEngine(_poly_scene()[0])
    eng2.ray_tracer(RayBundle(vertices=v, directions=d, energy=e, spectra=spec.copy(), wavelengths=swl), reps=2, min_energy=1e-9, tree=True, engine='protocol')
    P1 = eng2.tree[1]
    assert N.array_equal(P1.get_parents(), B1.get_parents())
    assert N.allclose(P1.get_spectra(), B1.get_spectra(), rtol=1e-12) and N.allclose(P1.get_energy(), B1.get_energy(), rtol=1e-12)


def test_carried_indices_and_materials_on_the_fast_path(ctx):
    """
    VERDICT r2 item 7: complex indices and tabulated materials in the streaming engine (k_s_shade_x; Refractive / RefractiveAbsorbant,
    optics_callables.py:726-858, :908-944).  The glass slab of the test above with one ray per hit, tree=False: ray_tracer picks the
    fast engine by itself; per surface, hits and segments equal the oracle's on the same Philox streams, absorbed and incident
    energy to 1e-9; the ordered engine's accountants of the floor collect the same hits (as sets: the fast engine delivers them in
    the order they were made).  Room for every list forced small, the call is reported, not wrong.
    """
    from tracer_amd.tracer_engine import TracerEngine
    from tracer_amd.scene import compile_scene
    from oracle import engine as oracle_engine
    n = 60000
    for absorb in (True, False):
        asm, air, glass = _slab_scene(True, absorb=absorb)
        b, wl = _bundle(n, air, seed=11, tilt=0.8)
        eng = TracerEngine(asm)
        eng.ray_tracer(b, reps=7, min_energy=1e-9, tree=False, seed=21)
        assert eng.stats['engine'] == 'fast' and eng.stats['form'] == 'stream'
        a, r, h = [x.copy() for x in eng.get_tallies()]
        cs = compile_scene(asm)
        with N.errstate(all='ignore'):
            o = oracle_engine.trace_bundle(cs, b.get_vertices(), b.get_directions(), b.get_energy(), 7, 1e-9, 21,
                                           ref_index=b.get_ref_index(), wavelengths=wl)
        assert N.array_equal(h, o['hits']) and eng.stats['segments'] == o['segments']
        assert N.allclose(a, o['absorbed'], rtol=1e-9, atol=1e-15) and N.allclose(r, o['received'], rtol=1e-9, atol=1e-15)
        assert h[2] > 0.5 * n and (a[0] + a[1] > 0) == absorb        # the slab itself takes energy only when it attenuates
        e_fast, p_fast = asm.get_surfaces()[2].get_optics_manager().get_all_hits()
        # the ordered engine on the same bundle: the floor's accountant holds the same hits
        asm2, _, _ = _slab_scene(True, absorb=absorb)
        eng2 = TracerEngine(asm2)
        eng2.ray_tracer(_bundle(n, air, seed=11, tilt=0.8)[0], reps=7, min_energy=1e-9, tree=False, seed=21, engine='ordered')
        e_ord, p_ord = asm2.get_surfaces()[2].get_optics_manager().get_all_hits()
        assert len(e_fast) == len(e_ord) == h[2]
        k_f, k_o = N.lexsort((p_fast[1], p_fast[0])), N.lexsort((p_ord[1], p_ord[0]))
        assert N.allclose(p_fast[:, k_f], p_ord[:, k_o], rtol=1e-12, atol=1e-12) and N.allclose(e_fast[k_f], e_ord[k_o], rtol=1e-9, atol=1e-18)
        a2, r2, h2 = eng2.get_tallies()
        assert N.array_equal(h2, h) and N.allclose(a2, a, rtol=1e-9, atol=1e-15)


def test_polychromatic_bundle_on_the_fast_path(ctx):
    """
    VERDICT r2 item 7: spectra in the streaming engine -- one value per sample and slot beside the ray table, scaled per sample at
    the polychromatic wall (optics_callables.py:393-425), as a whole elsewhere.  The cavity of the test above with the wall's
    spectral accountant taken off (the fast engine captures no spectra per hit): tree=False goes to the fast engine; per surface,
    hits, segments and energies equal the oracle's and the ordered engine's.
    """
    from tracer_amd.tracer_engine import TracerEngine
    from tracer_amd.scene import compile_scene
    from tracer_amd.ray_bundle import RayBundle
    from tracer_amd import optics_callables as opt
    from oracle import engine as oracle_engine

    def scene():
        asm, _ = _poly_scene()
        wall = asm.get_surfaces()[0].get_optics_manager()
        wall.accountants = [a for a in wall.accountants if not isinstance(a, opt.PolychromaticAccountant)]
        return asm

    n, W = 50000, 9
    rng = N.random.RandomState(8)
    v = N.vstack((rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), N.full(n, 1.)))
    d = N.vstack((rng.uniform(-0.4, 0.4, n), rng.uniform(-0.4, 0.4, n), -N.ones(n)))
    d /= N.sqrt(N.sum(d ** 2, axis=0))
    swl = N.sort(rng.uniform(0.3e-6, 2.5e-6, size=(W, n)), axis=0)
    spec = rng.uniform(0.5, 2., size=(W, n)) * 1e6
    e = N.trapezoid(spec, swl, axis=0)
    mk = lambda: RayBundle(vertices=v, directions=d, energy=e, spectra=spec.copy(), wavelengths=swl)
    asm = scene()
    eng = TracerEngine(asm)
    eng.ray_tracer(mk(), reps=6, min_energy=1e-9, tree=False, seed=33)
    assert eng.stats['engine'] == 'fast' and eng.stats['form'] == 'stream'
    a, r, h = [x.copy() for x in eng.get_tallies()]
    cs = compile_scene(asm)
    with N.errstate(all='ignore'):
        o = oracle_engine.trace_bundle(cs, v, d, e, 6, 1e-9, 33, wavelengths=swl, spectra=spec)
    assert N.array_equal(h, o['hits']) and eng.stats['segments'] == o['segments']
    assert N.allclose(a, o['absorbed'], rtol=1e-9, atol=1e-12) and N.allclose(r, o['received'], rtol=1e-9, atol=1e-12)
    assert h[0] > n and h[1] > 0.2 * n and h[2] > 0          # rays come back to the wall with the spectrum they left it with
    eng2 = TracerEngine(scene())
    eng2.ray_tracer(mk(), reps=6, min_energy=1e-9, tree=False, seed=33, engine='ordered')
    a2, r2, h2 = eng2.get_tallies()
    assert N.array_equal(h2, h) and N.allclose(a2, a, rtol=1e-9, atol=1e-12)
    # with the wall's spectral accountant in place (PolychromaticAccountant, optics_callables.py:1825-1848): the captured hits keep
    # their sample wavelengths and their spectra before and after the wall -- the accountant holds what the ordered engine gives it
    # (the fast engine delivers the hits in the order they were made: compared hit by hit after sorting both by hit point)
    asm3, asm4 = _poly_scene()[0], _poly_scene()[0]
    eng3, eng4 = TracerEngine(asm3), TracerEngine(asm4)
    eng3.ray_tracer(mk(), reps=6, min_energy=1e-9, tree=False, seed=33)
    eng4.ray_tracer(mk(), reps=6, min_energy=1e-9, tree=False, seed=33, engine='ordered')
    assert eng3.stats['engine'] == 'fast' and eng3.stats['form'] == 'stream' and eng4.stats['engine'] == 'ordered'
    got = asm3.get_surfaces()[0].get_optics_manager().get_all_hits()
    ref = asm4.get_surfaces()[0].get_optics_manager().get_all_hits()
    (e3, (w3, s3)), (e4, (w4, s4)) = got[:2], ref[:2]
    assert len(e3) == len(e4) == h[0] and s3.shape == s4.shape == (W, h[0]) and w3.shape == w4.shape
    k3, k4 = N.lexsort((s3[1], s3[0], e3)), N.lexsort((s4[1], s4[0], e4))
    assert N.allclose(e3[k3], e4[k4], rtol=1e-9, atol=1e-9)
    assert N.allclose(s3[:, k3], s4[:, k4], rtol=1e-9, atol=1e-6) and N.array_equal(w3[:, k3], w4[:, k4])
    assert N.allclose(N.trapezoid(s3, w3, axis=0), e3, rtol=1e-9)          # absorbed energy = integral of the absorbed spectrum
    # a second call: its hits and their spectra are appended behind those of the first
    eng3.ray_tracer(mk(), reps=6, min_energy=1e-9, tree=False, seed=34)
    e5, (w5, s5) = asm3.get_surfaces()[0].get_optics_manager().get_all_hits()[:2]
    assert len(e5) > 1.9 * h[0] and s5.shape[1] == len(e5) and N.allclose(s5[:, :h[0]][:, k3], s4[:, k4], rtol=1e-9, atol=1e-6)


def test_polychromatic_wall_without_spectra_is_an_error(ctx):
    from tracer_amd.tracer_engine import TracerEngine
    from tracer_amd._cabi import TracerAmdError
    from tracer_amd.ray_bundle import RayBundle
    asm, _ = _poly_scene()
    b = RayBundle(vertices=N.c_[[0., 0., 1.]], directions=N.c_[[0., 0., -1.]], energy=N.ones(1))
    with pytest.raises(TracerAmdError) as err:
        TracerEngine(asm).ray_tracer(b, reps=2, min_energy=1e-9)
    assert 'spectra' in str(err.value)


def test_periodic_cell_vs_oracle_and_known_answer(ctx):
    """
    PeriodicBoundary as a native optics kind (optics_callables.py:690-723): a cell 2 m wide between two
    periodic walls at x = -1 and x = +1, a diffuse floor and a black ceiling that captures its hits.  Rays cross the walls,
    re-enter one period along the wall's normal and go on.  The ordered engine records the reference's bundle -- per wall
    the stubs of energy 0 (culled: behind the live rays) and the moved rays -- level by level as the oracle does; both forms
    of the fast engine follow the moved ray and end with the oracle's tallies; a ray aimed by hand lands where the
    periodic image says.
    """
    from tracer_amd import optics_callables as opt
    from tracer_amd.assembly import Assembly
    from tracer_amd.object import AssembledObject
    from tracer_amd.surface import Surface
    from tracer_amd.flat_surface import RectPlateGM
    from tracer_amd.spatial_geometry import roty, rotx, translate
    from tracer_amd.tracer_engine import TracerEngine
    from tracer_amd.ray_bundle import RayBundle
    from tracer_amd.scene import compile_scene, DeviceScene
    from oracle import engine as oracle_engine

    def cell():
        left = Surface(RectPlateGM(8., 4.), opt.PeriodicBoundary(2.))      # plates in the y-z plane: local z along global x
        right = Surface(RectPlateGM(8., 4.), opt.PeriodicBoundary(2.))
        floor = Surface(RectPlateGM(2., 4.), opt.Lambertian(0.3))
        top = Surface(RectPlateGM(2., 4.), opt.LambertianReceiver(1.))
        objs = [AssembledObject(surfs=[left], transform=N.dot(translate(-1., 0., 2.), roty(N.pi / 2.))),
                AssembledObject(surfs=[right], transform=N.dot(translate(1., 0., 2.), roty(N.pi / 2.))),
                AssembledObject(surfs=[floor]),
                AssembledObject(surfs=[top], transform=N.dot(translate(0., 0., 3.), rotx(N.pi)))]
        return Assembly(objects=objs), top
    n = 6000
    rng = N.random.default_rng(11)
    v = N.vstack((rng.uniform(-0.9, 0.9, n), rng.uniform(-1.5, 1.5, n), N.full(n, 2.5)))
    d = N.vstack((rng.uniform(-1.5, 1.5, n), rng.uniform(-0.2, 0.2, n), -N.ones(n)))
    d /= N.sqrt(N.sum(d ** 2, axis=0))
    e = N.ones(n) / n
    asm, top = cell()
    cs = compile_scene(asm)
    from tracer_amd import _cabi
    assert [dsc.optics_kind for dsc in cs.descs][:2] == [_cabi.OPT_PERIODIC_BOUNDARY] * 2 and not cs.splits
    with N.errstate(all='ignore'):
        o = oracle_engine.trace_bundle(cs, v, d, e, 8, 1e-9, 33)
    # ordered engine: the tree, level by level
    eng = TracerEngine(asm)
    eng.ray_tracer(RayBundle(vertices=v, directions=d, energy=e), reps=8, min_energy=1e-9, tree=True, seed=33)
    assert eng.stats['engine'] == 'ordered'
    _compare_levels(eng.tree, o, complex_index=False)
    L1 = eng.tree[1]
    crossed = int((o['levels'][1]['surf'] <= 1).sum()) // 2
    assert crossed > n // 4, "a good share of the rays crosses a wall before the floor"
    assert (L1.get_energy() == 0).sum() == crossed and (L1.get_energy()[-crossed:] == 0).all(), "the stubs: energy 0, behind the live rays"
    a_ord, r_ord, h_ord = eng.get_tallies()
    assert N.array_equal(h_ord, o['hits']) and N.allclose(a_ord, o['absorbed'], rtol=1e-9, atol=1e-12)
    assert a_ord[0] == 0. and a_ord[1] == 0., "a periodic boundary keeps nothing"
    # the fast engine, both forms: the moved ray is the ray
    for stream in (False, True):
        dev = DeviceScene(cs, ctx)
        st, _ = dev.trace_fast(RayBundle(vertices=v, directions=d, energy=e), 8, 1e-9, 33, stream=stream)
        a, r, h = dev.get_tallies()
        dev.close()
        assert N.array_equal(h, o['hits']) and N.allclose(a, o['absorbed'], rtol=1e-9, atol=1e-12), stream
        assert st.segments == o['segments']
    # known answer: from (0.5, 0, 1) along (+1, 0, +1) / sqrt 2 the ray meets the right wall at (1, 0, 1.5), re-enters at
    # (-1, 0, 1.5) -- one period along the normal that faces it -- and reaches the ceiling z = 3 at x = -1 + 1.5 = 0.5
    asm, top = cell()
    eng = TracerEngine(asm)
    eng.ray_tracer(RayBundle(vertices=N.c_[[0.5, 0., 1.]], directions=N.c_[[1., 0., 1.]] / N.sqrt(2.), energy=N.r_[1.]), reps=5,
                   min_energy=1e-9, tree=True, seed=1)
    absorbed, hits = top.get_optics_manager().get_all_hits()
    assert N.allclose(absorbed, [1.]) and N.allclose(hits[:, 0], [0.5, 0., 3.], atol=1e-12)
    assert N.allclose(eng.tree[1].get_vertices(), N.c_[[-1., 0., 1.5], [1., 0., 1.5]], atol=1e-12) and N.allclose(eng.tree[1].get_energy(), [1., 0.])
