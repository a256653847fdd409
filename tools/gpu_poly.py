"""Rays that carry spectra and complex indices, fast engine (streaming form, k_s_shade_x) against the ordered engine, tree=False:
(1) a box of a polychromatic wall, a mirror and a diffuse wall (tests/test_gpu_media.py), W samples per ray; (2) the glass slab between
tabulated materials, one ray per hit.  usage: gpu_poly.py [rays, default 2e6] [W, default 16]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as N
from tracer_amd import optics_callables as opt
from tracer_amd.assembly import Assembly
from tracer_amd.object import AssembledObject
from tracer_amd.surface import Surface
from tracer_amd.flat_surface import RectPlateGM
from tracer_amd.spatial_geometry import translate, rotx
from tracer_amd.ray_bundle import RayBundle
from tracer_amd.tracer_engine import TracerEngine

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2000000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rng = N.random.RandomState(8)


def poly_scene():
    ths = N.linspace(0., N.pi / 2., 7)
    wls = N.linspace(0.25e-6, 2.6e-6, 5)
    grid = 0.2 + 0.7 * N.outer(N.cos(ths) ** 0.5, 1. / (1. + (wls * 1e6 - 1.) ** 2))
    wall = opt.Lambertian_directional_axisymmetric_piecewise_PolychromaticAbsorberPolychromatic(ths, grid, wls)      # (absorbed energy and absorbed spectrum per hit)
    floor = AssembledObject(surfs=[Surface(RectPlateGM(4., 4.), wall)], transform=translate(0., 0., 0.))
    roof = AssembledObject(surfs=[Surface(RectPlateGM(4., 4.), opt.Reflective(0.1))], transform=N.dot(translate(0., 0., 1.5), rotx(N.pi)))
    side = AssembledObject(surfs=[Surface(RectPlateGM(4., 1.5), opt.Lambertian(0.3))], transform=N.dot(translate(0., 2., 0.75), rotx(N.pi / 2.)))
    return Assembly(objects=[floor, roof, side])


def slab_scene():
    tl = N.linspace(0.3e-6, 2.5e-6, 6)
    air = opt.TabulatedMaterial(tl, N.ones(6), N.zeros(6))
    glass = opt.TabulatedMaterial(tl, [1.55, 1.53, 1.51, 1.50, 1.49, 1.47], [3e-8, 2e-8, 1e-8, 5e-8, 2e-7, 6e-7])
    mk = lambda: opt.RefractiveAbsorbant(air, glass, single_ray=True, attenuation_coefficient_1=1.)
    top = AssembledObject(surfs=[Surface(RectPlateGM(40., 40.), mk())], transform=translate(0., 0., 0.5))
    bottom = AssembledObject(surfs=[Surface(RectPlateGM(40., 40.), mk())], transform=translate(0., 0., 0.))
    floor = AssembledObject(surfs=[Surface(RectPlateGM(60., 60.), opt.LambertianReceiver(1.))], transform=translate(0., 0., -1.))
    return Assembly(objects=[top, bottom, floor]), air


v = N.vstack((rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), N.full(n, 1.)))
d = N.vstack((rng.uniform(-0.4, 0.4, n), rng.uniform(-0.4, 0.4, n), -N.ones(n)))
d /= N.sqrt(N.sum(d ** 2, axis=0))
swl = N.sort(rng.uniform(0.3e-6, 2.5e-6, size=(W, n)), axis=0)
spec = rng.uniform(0.5, 2., size=(W, n)) * 1e6
e = N.trapezoid(spec, swl, axis=0)
for engine in ('fast', 'ordered'):
    eng = TracerEngine(poly_scene())
    for r in range(2):
        b = RayBundle(vertices=v, directions=d, energy=e, spectra=spec, wavelengths=swl)
        t0 = time.time()
        eng.ray_tracer(b, reps=6, min_energy=1e-9, tree=False, seed=33 + r, engine=engine)
        wall = time.time() - t0
        t0 = time.time()
        absorbed, (hw, hs) = eng._asm.get_surfaces()[0].get_optics_manager().get_all_hits()
        read = time.time() - t0
        eng._asm.reset_all_optics()
        print('polychromatic box, %d rays x %d samples, %s engine, run %d: wall %.1f ms, kernels %.2f ms, %d segments (%.0f M segments/s by kernel time); '
              'reading the wall\'s %d hits with their spectra %.1f ms'
              % (n, W, engine, r, wall * 1e3, eng.stats['kernel_ms'], eng.stats['segments'], eng.stats['segments'] / eng.stats['kernel_ms'] / 1e3, len(absorbed), read * 1e3), flush=True)
asm, air = slab_scene()
v[2] = 3.
wl = rng.uniform(0.4e-6, 2.4e-6, n)
for engine in ('fast', 'ordered'):
    eng = TracerEngine(asm)
    for r in range(2):
        b = RayBundle(vertices=v, directions=d, energy=N.ones(n) / n, ref_index=air.m(wl), wavelengths=wl)
        t0 = time.time()
        eng.ray_tracer(b, reps=7, min_energy=1e-9, tree=False, seed=21 + r, engine=engine)
        wall = time.time() - t0
        print('glass slab, %d rays, %s engine, run %d: wall %.1f ms, kernels %.2f ms, %d segments (%.0f M segments/s by kernel time)'
              % (n, engine, r, wall * 1e3, eng.stats['kernel_ms'], eng.stats['segments'], eng.stats['segments'] / eng.stats['kernel_ms'] / 1e3), flush=True)
