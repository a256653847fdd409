"""Where the time of one TracerEngine.ray_tracer(tree=False, accel=True) call on the bench workload goes (host side): cProfile,
top entries by cumulative time.  usage: api_profile.py [rays] [read]"""
import cProfile, pstats, sys, os, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as N
from tracer_amd import scenes
from tracer_amd.tracer_engine import TracerEngine
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000000
plant, field, rec, src = scenes.nsttf_field()
eng = TracerEngine(plant)
ue, ve = scenes.nsttf_fluxmap_edges()
eng.set_fluxmap(218, ue, ve)
mk = lambda k: scenes.nsttf_source(n, src, seed=2024, ray_offset=k * n)
eng.ray_tracer(mk(0), reps=100, min_energy=1e-10, tree=False, accel=True, seed=2024)
plant.reset_all_optics()
rec_opt = plant.get_surfaces()[218].get_optics_manager()
rec_opt.get_all_hits()
plant.reset_all_optics()
read = len(sys.argv) > 2        # any second argument: the receiver's accountants are read after the call (the hits come off the device)
import time
for k in range(3):
    t0 = time.time()
    eng.ray_tracer(mk(10 + k), reps=100, min_energy=1e-10, tree=False, accel=True, seed=2024)
    t1 = time.time()
    if read:
        got = len(rec_opt.get_all_hits()[0])
    t2 = time.time()
    plant.reset_all_optics()
    t3 = time.time()
    print('call %.2f ms  read %.2f ms  reset %.2f ms' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3), flush=True)
pr = cProfile.Profile()
pr.enable()
eng.ray_tracer(mk(1), reps=100, min_energy=1e-10, tree=False, accel=True, seed=2024)
if read:
    rec_opt.get_all_hits()
    plant.reset_all_optics()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(18)
print(s.getvalue())
