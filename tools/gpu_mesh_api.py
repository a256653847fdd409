"""The relief of 105 800 triangles through TracerEngine.ray_tracer as a script calls it: tree=True (the reference's default: the
ordered engine) and tree=False (the fast engine), accel=True.  usage: gpu_mesh_api.py [rays, default 1e6]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as N
from tracer_amd import scenes, sources
from tracer_amd.tracer_engine import TracerEngine

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000
t0 = time.time()
asm, nf, (center, direction, radius, csr) = scenes.relief_mesh()
eng = TracerEngine(asm)
print('%d faces, assembly built in %.2f s' % (nf, time.time() - t0), flush=True)
for tree in (False, True):
    for r in range(3):
        b = sources.buie_sunshape(n, center, direction, radius, csr, flux=1., seed=23 + r)
        t0 = time.time()
        eng.ray_tracer(b, reps=6, min_energy=1e-10, tree=tree, accel=True, seed=23 + r)
        wall = time.time() - t0
        print('tree=%s run %d: %d rays, engine %s, wall %.1f ms, kernels %.2f ms, %d segments' %
              (tree, r, n, eng.stats['engine'], wall * 1e3, eng.stats['kernel_ms'], eng.stats['segments']), flush=True)
