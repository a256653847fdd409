"""A Monte-Carlo loop as scripts of the reference write it: many calls of TracerEngine.ray_tracer(tree=False) on modest bundles.
Wall time per call at 1e5 and 1e6 NSTTF rays, with the receiver's accountants fed and without.  usage: api_loop.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as N
from tracer_amd import scenes
from tracer_amd.tracer_engine import TracerEngine
plant, field, rec, src = scenes.nsttf_field()
eng = TracerEngine(plant)
for n in (100000, 1000000):
    for feed, unchanged in ((True, False), (False, False), (True, True), (False, True)):
        ts = []
        for r in range(8):
            b = scenes.nsttf_source(n, src, seed=3, ray_offset=r * n)
            t0 = time.perf_counter()
            eng.ray_tracer(b, reps=100, min_energy=1e-10, tree=False, accel=True, seed=3, feed=feed, scene_unchanged=unchanged)
            ts.append((time.perf_counter() - t0) * 1e3)
        print('n %8d feed=%-5s scene_unchanged=%-5s wall ms per call: %s   (kernels %.3f ms)' % (n, feed, unchanged, ' '.join('%.2f' % t for t in ts), eng.stats['kernel_ms']), flush=True)
if len(sys.argv) > 1:
    import cProfile, pstats, io
    pr = cProfile.Profile(); pr.enable()
    for r in range(20):
        eng.ray_tracer(scenes.nsttf_source(100000, src, seed=3, ray_offset=r), reps=100, min_energy=1e-10, tree=False, accel=True, seed=3)
    pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(14); print(s.getvalue())
