"""A day of sun positions over the NSTTF field as a script of the reference would run it: field.track_sun(azimuth, zenith), a new
source bundle, TracerEngine.ray_tracer(tree=False, accel=True) per position.  Wall time per position.  usage: api_sun_sweep.py [rays] [p]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as N
from tracer_amd import scenes, sources
from tracer_amd.tracer_engine import TracerEngine
from tracer_amd.models.heliostat_field import solar_vector
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000
plant, field, rec, src = scenes.nsttf_field()
pos = scenes.nsttf_positions()
aim = N.tile(N.array([0., 0., 60.]), (pos.shape[0], 1))
centre = src['center'][:, 0] - 300. * solar_vector(0., 35.05 * N.pi / 180.)
eng = TracerEngine(plant)


def one(k):
    az, zen = (-60. + 8. * k) * N.pi / 180., (60. - 2.5 * min(k, 15 - k)) * N.pi / 180.
    t0 = time.perf_counter()
    field.track_sun(az, zen, aim_points=aim)
    t1 = time.perf_counter()
    sun = solar_vector(az, zen)
    b = sources.buie_sunshape(n, N.vstack(300. * sun + centre), -sun, src['radius'], 0.01, flux=1000., pre_process_CSR=False, seed=k)
    plant.reset_all_optics()
    eng.ray_tracer(b, reps=100, min_energy=1e-10, tree=False, accel=True, seed=k)
    t2 = time.perf_counter()
    return (t1 - t0) * 1e3, (t2 - t1) * 1e3


ts = [one(k) for k in range(16)]
print('%d rays per position; per position: track_sun %s ms' % (n, ' '.join('%.1f' % a for a, b in ts[:8])))
print('                          ray_tracer (new poses -> device, new footprint map, trace, accountants) %s ms' % ' '.join('%.1f' % b for a, b in ts))
print('receiver power of the last position: %.1f kW' % (eng.get_tallies()[0][218] / 1e3))
if len(sys.argv) > 2:
    import cProfile, pstats, io
    pr = cProfile.Profile(); pr.enable()
    for k in range(16, 24):
        one(k)
    pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(14); print(s.getvalue())
