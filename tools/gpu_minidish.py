"""The only scene the reference publishes a timing for (user-doc/screenshots.html:17-20: "as low as 1.7 seconds ... to do 100,000 rays",
one core, about 2010; 1.08 s in this build's container with NumPy 2.2): examples/test_case.py -- a 5 m dish of f = 6.25 m tilted by 45
degrees, a 0.4 m square receiver behind a 0.7 m homogenizer, both 90 % reflective, under a 3 m pillbox disc source of 5 mrad, 100
iterations, min_energy 1e-6.  Traced through TracerEngine.ray_tracer at the published size and at the sizes a GPU is for.
usage: gpu_minidish.py [largest ray count, default 1e8]"""
import sys, os, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as N
from tracer_amd.tracer_engine import TracerEngine
from tracer_amd.sources import solar_disk_bundle
from tracer_amd.spatial_geometry import rotx
from tracer_amd.models.tau_minidish import MiniDish

top = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000000
focus, h_depth, side = 6.25, 0.7, 0.4
x = -1 / math.sqrt(2)
sizes = [n for n in (100000, 1000000, 10000000, 100000000) if n <= top]
edges = N.linspace(-side / 2., side / 2., 21)
power_in = 1000. * math.pi * 9.


def run(n, flow):
    dish = MiniDish(5., focus, 0.9, focus + h_depth, side, h_depth, 0.9)
    dish.set_transform(rotx(-N.pi / 4))
    plate = dish.get_receiver_surf().get_surfaces()[0]
    t0 = time.time()
    sun = solar_disk_bundle(n, N.c_[[0, 7., 7.]], N.array([0, x, x]), 3., 0.005, flux=1000., seed=5)
    engine = TracerEngine(dish)
    if flow == 'device':    # the flux map is binned by the kernels as rays land (O8); hits stay on the device, accountants are not fed
        engine.set_fluxmap(plate, edges, edges)
        engine.ray_tracer(sun, 100, 1e-6, tree=False, feed=False)
        H = engine.get_fluxmap(plate)
    elif flow == 'no tree':  # every hit of the plate and of the four detector walls comes back to the host, the ray tree is not kept
        engine.ray_tracer(sun, 100, 1e-6, tree=False)
        H = dish.histogram_hits(bins=20)[0]
    else:                   # the call of the script, word for word: the whole ray tree comes back as well (ordered engine)
        engine.ray_tracer(sun, 100, 1e-6)
        H = dish.histogram_hits(bins=20)[0]
    wall = time.time() - t0
    return wall, engine.stats, H


run(100000, 'as written')                         # context, library and buffers come up here
names = {'as written': 'ray tree + histogram_hits (the script as written)', 'no tree': 'tree=False + histogram_hits', 'device': 'flux map on the device'}
for n in sizes:
    for flow in ('as written', 'no tree', 'device'):
        if flow == 'as written' and n > 10000000:
            continue                              # a tree of 1e8 rays is 20 GB of host arrays
        wall, st, H = min((run(n, flow) for _ in range(5 if n <= 1000000 else 1)), key=lambda r: r[0])     # small sizes: best of five
        print('%9d rays, %s: %8.1f ms from bundle to flux map (kernels %.2f ms), %d segments, %.1f M segments/s by wall; receiver '
              '%.1f W of %.1f W (%.4f; the reference got 0.6010 at 1e5 rays), peak %.0f suns' %
              (n, names[flow], wall * 1e3, st['kernel_ms'], st['segments'],
               st['segments'] / wall / 1e6, H.sum(), power_in, H.sum() / power_in, H.max() / (side / 20) ** 2 / 1000.), flush=True)
