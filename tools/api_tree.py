"""Default call of the public entry point -- TracerEngine.ray_tracer(bundle) with tree=True: the ordered engine, every level of the
RayTree recorded and copied to the host -- on the NSTTF field.  usage: api_tree.py [rays, default 1e7]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as N
from tracer_amd import scenes
from tracer_amd.tracer_engine import TracerEngine
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10000000
plant, field, rec, src = scenes.nsttf_field()
eng = TracerEngine(plant)
for host_bundle, read in ((True, False), (False, False), (False, True)):
    for r in range(3):
        b = scenes.nsttf_source(n, src, seed=5, ray_offset=r * n)
        if host_bundle:
            b.get_vertices()            # the bundle on the host, as a script of the reference has it
        plant.reset_all_optics()
        t0 = time.time()
        eng.ray_tracer(b, reps=100, min_energy=1e-10, tree=True, accel=True, seed=5)
        wall = time.time() - t0
        st = eng.stats
        t1 = time.time()
        if read:
            e_last = eng.tree[-1].get_energy().sum()           # the last level comes off the device now
            rec_hits = plant.get_surfaces()[218].get_optics_manager().get_all_hits()      # ... and the accountants are fed now
        t_read = time.time() - t1
        print('   trace call %.1f ms; before it: %s' % (st['wall_s'] * 1e3, ', '.join('%s %.1f ms' % (k, v * 1e3) for k, v in st['host_s'].items())))
        print('%s run %d: %d rays, wall %.1f ms, kernels %.1f ms, %d segments -> %.0f M segments/s end to end; reading the tree and the '
              'receiver afterwards %.1f ms; levels %s' %
              ('bundle on the host' if host_bundle else 'bundle from its descriptor', r, n, wall * 1e3, st['kernel_ms'], st['segments'],
               st['segments'] / wall / 1e6, t_read * 1e3, [eng.tree[k].get_num_rays() for k in range(eng.tree.num_bunds())]), flush=True)
if len(sys.argv) > 2:
    import cProfile, pstats, io
    b = scenes.nsttf_source(n, src, seed=5, ray_offset=7 * n)
    if sys.argv[2] != 'd':          # 'd': the bundle stays a descriptor (generated on the device)
        b.get_vertices()
    plant.reset_all_optics()
    pr = cProfile.Profile(); pr.enable()
    eng.ray_tracer(b, reps=100, min_energy=1e-10, tree=True, accel=True, seed=5)
    pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(12); print(s.getvalue())
