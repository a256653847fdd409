"""The scenes BASELINE.md section 2 times the reference on (1e5 source rays, one CPU core: flat pair 0.80 s, dish 2.63 s, NSTTF 4.63 s
brute force / 2.32 s with its Kd-tree), traced the way a script of the reference does it -- a new engine, ray_tracer with the ray
tree kept (ordered engine), the receiver's accountant read back -- and the same with tree=False.  Wall time per run, best of five.
usage: api_reference_sizes.py [rays, default 1e5]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as N
from tracer_amd import scenes
from tracer_amd.tracer_engine import TracerEngine

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000
ref = {'flat pair': 0.80, 'dish': 2.63, 'NSTTF': 4.63, 'NSTTF accel=True': 2.32}


def flat():
    asm, mirror, rec, src = scenes.flat_pair()
    return asm, rec, lambda seed: scenes.flat_pair_source(n, src, seed=seed), dict(reps=10, min_energy=1e-10)


def dish():
    asm, dish_s, rec_s, src = scenes.dish()
    return asm, rec_s, lambda seed: scenes.dish_source(n, src, seed=seed), dict(reps=10, min_energy=1e-10)


def nsttf(accel):
    def make():
        plant, field, rec, src = scenes.nsttf_field()
        return plant, rec.get_surfaces()[0], lambda seed: scenes.nsttf_source(n, src, seed=seed), dict(reps=100, min_energy=1e-10, accel=accel)
    return make


TracerEngine(scenes.flat_pair()[0]).ray_tracer(scenes.flat_pair_source(1000, scenes.flat_pair()[3], seed=1), 2, 1e-10)     # context up
for name, make in (('flat pair', flat), ('dish', dish), ('NSTTF', nsttf(False)), ('NSTTF accel=True', nsttf(True))):
    out = []
    for tree in (True, False):
        best = None
        for k in range(5):
            asm, rec, source, kw = make()
            t0 = time.time()
            eng = TracerEngine(asm)
            eng.ray_tracer(source(10 + k), tree=tree, seed=10 + k, **kw)
            power = rec.get_optics_manager().get_all_hits()[0].sum()
            wall = time.time() - t0
            best = wall if best is None else min(best, wall)
        out.append((best, power, eng.stats['engine']))
    print('%-17s %d rays: reference %.2f s | tree kept (%s engine) %.1f ms = %.0fx | tree=False (%s engine) %.1f ms = %.0fx | receiver %.4g W' %
          (name, n, ref[name], out[0][2], out[0][0] * 1e3, ref[name] / out[0][0], out[1][2], out[1][0] * 1e3, ref[name] / out[1][0], out[1][1]), flush=True)
