import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as N
from tracer_amd import _cabi, scenes
from tracer_amd.scene import compile_scene, DeviceScene
from tracer_amd.accel_tree import KdTree
ctx = _cabi.get_context(0)
n = int(float(sys.argv[1]))
plant, field, rec, src = scenes.nsttf_field()
cs = compile_scene(plant)
kd = KdTree(plant, 8 + 1.3 * N.log(cs.n_surf), min_leaf=1)
dev = DeviceScene(cs, ctx); dev.set_kdtree(kd)
for accel in (True, False):
    best = 1e9
    for r in range(3):
        st, _ = dev.trace_fast(scenes.nsttf_source(n, src, seed=7), 100, 1e-10, 7, accel=accel)
        best = min(best, st.kernel_ms)
    print('accel=%s %8.3f ms %8.1f Mseg/s segs %d hits %d' % (accel, best, st.segments / best / 1e3, st.segments, st.hits), flush=True)
