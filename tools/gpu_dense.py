"""Dense scenes of overlapping curved shapes: `shapes` plates, discs, spheres, hemispheres, cylinders and dishes
thrown into a 12 m cube, a pillbox disc of rays, up to 50 interactions.  Kernel time of the megakernel (brute force), of the streaming
form on the LDS-sized grid (a lane per ray) and -- TRC_GRID_FORCE32=1 -- on the large grid, where the lanes of a wave share the tests
(k_s_bounce_coop).  usage: gpu_dense.py [rays, default 1e7] [shapes, default 150]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as N
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

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10000000
shapes = int(sys.argv[2]) if len(sys.argv) > 2 else 150
sun = N.r_[0.2, -0.1, -1.] / N.linalg.norm([0.2, -0.1, -1.])
rng = N.random.RandomState(5)
objs = []
for _ in range(shapes):
    kind, s = rng.randint(0, 6), rng.uniform(0.2, 1.5)
    gm = (RectPlateGM(2 * s, s), RoundPlateGM(s), SphericalGM(s), HemisphereGM(s), FiniteCylinder(2 * s, 3 * s), ParabolicDishGM(2 * s, rng.uniform(0.5, 2.)))[kind]
    o = AssembledObject(surfs=[Surface(gm, RealReflective(0.2, 2e-3) if kind % 2 else Reflective(0.2))])
    ax = rng.normal(size=3)
    o.set_transform(generate_transform(ax / N.linalg.norm(ax), rng.uniform(0, 2 * N.pi), rng.uniform(-6., 6., 3)[:, None]))
    objs.append(o)
eng = TracerEngine(Assembly(objects=objs))
for form, accel in (('megakernel', False), ('stream', True)):
    for r in range(2):
        eng.reset_tallies()
        eng.ray_tracer(disk_bundle(n, N.c_[-12. * sun], sun, 9., 4.65e-3, flux=1., seed=9 + r), 50, 1e-8, accel=accel, seed=9 + r, tree=False, fast_kernel=form)
    a, rr, h = eng.get_tallies()
    print('%s%s: %d rays, %d shapes, kernels %.2f ms, %d segments (%.0f M segments/s), hits %d' %
          (form, ' (large grid)' if os.environ.get('TRC_GRID_FORCE32') and form == 'stream' else '', n, shapes, eng.stats['kernel_ms'], eng.stats['segments'],
           eng.stats['segments'] / eng.stats['kernel_ms'] / 1e3, int(h.sum())), flush=True)
