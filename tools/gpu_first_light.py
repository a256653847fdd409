"""First-light script (not a test): run the benchmark scenes on the GPU and print sanity anchors."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as N
from tracer_amd import _cabi, scenes
from tracer_amd.tracer_engine import TracerEngine

ctx = _cabi.get_context(0)
print('device:', ctx.device_name(), flush=True)

# dish: intercept factor anchor 0.9346 (reference, 1e5 rays)
asm, dish_s, rec_s, src = scenes.dish()
eng = TracerEngine(asm)
for n in (100000, 1000000):
    b = scenes.dish_source(n, src, seed=42)
    t = time.time()
    eng.reset_tallies(); asm.reset_all_optics()
    eng.ray_tracer(b, reps=10, tree=False, seed=42)
    a, r, h = eng.get_tallies()
    etot = 1000. * N.pi * 2.5 ** 2
    print('dish fast n=%d wall %.3fs stats %s' % (n, time.time() - t, eng.stats), flush=True)
    print('  absorbed', a, 'hits', h, 'intercept factor %.4f' % (a[1] / (etot * 0.94)), flush=True)
    hits = rec_s.get_optics_manager().get_all_hits()
    print('  receiver accountant: n=%d sumE=%.3f' % (len(hits[0]), hits[0].sum()), flush=True)
b = scenes.dish_source(100000, src, seed=42)
eng.reset_tallies(); asm.reset_all_optics()
eng.ray_tracer(b, reps=10, tree=True, seed=42)
print('dish ordered sizes', [eng.tree[k].get_num_rays() for k in range(eng.tree.num_bunds())], eng.stats, flush=True)
a2, _, h2 = eng.get_tallies()
print('  absorbed', a2, h2, flush=True)

# NSTTF: bundle sizes ~ N*[1, .065, .064]; receiver ~5.19 MW
plant, field, rec, src = scenes.nsttf_field()
eng = TracerEngine(plant)
ue, ve = scenes.nsttf_fluxmap_edges()
eng.set_fluxmap(218, ue, ve)
b = scenes.nsttf_source(100000, src, seed=7)
eng.ray_tracer(b, reps=100, tree=True, seed=7)
print('nsttf ordered sizes', [eng.tree[k].get_num_rays() for k in range(eng.tree.num_bunds())], eng.stats, flush=True)
a, r, h = eng.get_tallies()
print('  receiver kW %.3f hits %d ; fluxmap sum %.3f' % (a[218] / 1e3, h[218], eng.get_fluxmap(218).sum() / 1e3), flush=True)
for accel in (False, True):
    for n in (100000, 10000000):
        eng.reset_tallies(); plant.reset_all_optics()
        b = scenes.nsttf_source(n, src, seed=7)
        t = time.time()
        eng.ray_tracer(b, reps=100, tree=False, accel=accel, seed=7)
        a, r, h = eng.get_tallies()
        st = eng.stats
        print('nsttf fast accel=%s n=%d wall %.3fs kernel %.2f ms  %.1f Mseg/s  receiver kW %.3f hits %d' % (
            accel, n, time.time() - t, st['kernel_ms'], st['segments'] / st['kernel_ms'] / 1e3, a[218] / 1e3 * (1e5 / n) * (n / 1e5), h[218]), flush=True)
