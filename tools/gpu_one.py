import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as N
from tracer_amd import _cabi, scenes
from tracer_amd.scene import compile_scene, DeviceScene
from tracer_amd.accel_tree import KdTree
ctx = _cabi.get_context(0)
n = int(float(sys.argv[1])); accel = sys.argv[2] == 'kd'
plant, field, rec, src = scenes.nsttf_field()
cs = compile_scene(plant)
dev = DeviceScene(cs, ctx)
if accel:
    dev.set_kdtree(KdTree(plant, 8 + 1.3 * N.log(cs.n_surf), min_leaf=1))
for r in range(2):
    st, _ = dev.trace_fast(scenes.nsttf_source(n, src, seed=7), 100, 1e-10, 7, accel=accel)
print('accel=%s %8.3f ms %8.1f Mseg/s segs %d' % (accel, st.kernel_ms, st.segments / st.kernel_ms / 1e3, st.segments), flush=True)
