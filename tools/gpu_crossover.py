"""wall time of trc_trace_fast in its two forms over the call size (NSTTF, accel): where should the default switch?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as N
from tracer_amd import _cabi, scenes
from tracer_amd.scene import compile_scene, DeviceScene
from tracer_amd.accel_tree import KdTree
ctx = _cabi.get_context(0)
plant, field, rec, src = scenes.nsttf_field()
cs = compile_scene(plant)
dev = DeviceScene(cs, ctx)
dev.set_kdtree(KdTree(plant, 8 + 1.3 * N.log(cs.n_surf), min_leaf=1))
for n in (1000, 10000, 30000, 100000, 300000, 1000000, 3000000):
    row = []
    for stream in (False, True):
        best = 1e9
        for r in range(4):
            t0 = time.time()
            st, _ = dev.trace_fast(scenes.nsttf_source(n, src, seed=7), 100, 1e-10, 7, accel=True, stream=stream)
            best = min(best, time.time() - t0)
        row.append((best * 1e3, st.kernel_ms))
    print('n %8d  megakernel wall %8.3f ms (kernel %7.3f)   streaming wall %8.3f ms (kernels %7.3f)' % (n, row[0][0], row[0][1], row[1][0], row[1][1]), flush=True)
