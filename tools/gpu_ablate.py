"""Ablation timings of k_trace_fast on the NSTTF workload (not a test)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as N
from tracer_amd import _cabi, scenes
from tracer_amd.scene import compile_scene, DeviceScene, TableScene
from tracer_amd.accel_tree import KdTree
from tracer_amd.ray_bundle import RayBundle

ctx = _cabi.get_context(0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20000000
plant, field, rec, src = scenes.nsttf_field()
cs = compile_scene(plant)
kd = KdTree(plant, 8 + 1.3 * N.log(cs.n_surf), min_leaf=1)


def run(name, dev, bundle_fn, accel, reps=3):
    best = 1e9
    for r in range(reps):
        st, _ = dev.trace_fast(bundle_fn(), 100, 1e-10, 7, accel=accel)
        best = min(best, st.kernel_ms)
    print('%-44s %9.3f ms  %8.1f Mseg/s  segs %d hits %d' % (name, best, st.segments / best / 1e3, st.segments, st.hits), flush=True)
    return best


def srcb():
    return scenes.nsttf_source(n, src, seed=7)


# a: one far-away dummy surface: source generation + one plane test
dummy = TableScene([_cabi.GM_RECT], [_cabi.OPT_REFLECTIVE], [N.eye(4) + N.diag([0, 0, 0, 0.])], N.array([[1e-3, 1e-3] + [0.] * 14]),
                   N.zeros((1, 8)), N.zeros(0), [-1], [0])
dummy.descs[0].frame[3] = 1e6
d0 = DeviceScene(dummy, ctx)
run('a. source gen + 1 dummy surface', d0, srcb, False)
dev = DeviceScene(cs, ctx)
dev.set_kdtree(kd)
run('b. nsttf brute, fused source', dev, srcb, False)
run('c. nsttf kd, fused source', dev, srcb, True)
# given bundle (materialise once on host, then upload each call: upload is not in kernel_ms)
m = min(n, 10000000)
b = scenes.nsttf_source(m, src, seed=7)
v, d, e = b.get_vertices(), b.get_directions(), b.get_energy()
hb = RayBundle(vertices=v, directions=d, energy=e)
run('d. nsttf kd, given bundle (%d)' % m, dev, lambda: hb, True)
run('e. nsttf brute, given bundle (%d)' % m, dev, lambda: hb, False)
run('f. dummy, given bundle (%d)' % m, d0, lambda: hb, False)
# with flux map + hit capture like the bench
ue, ve = scenes.nsttf_fluxmap_edges()
dev2 = DeviceScene(cs, ctx)
dev2.set_kdtree(kd)
dev2.set_fluxmap(218, ue, ve)
dev2.set_hit_capacity(int(0.08 * n * 4) + 4096)
run('g. nsttf kd, fused, fluxmap + hit capture', dev2, srcb, True)
dev3 = DeviceScene(cs, ctx)
dev3.set_kdtree(kd)
dev3.set_fluxmap(218, ue, ve)
run('h. nsttf kd, fused, fluxmap only', dev3, srcb, True)
