import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as N
from tracer_amd import _cabi, scenes
from tracer_amd.scene import compile_scene, DeviceScene
ctx = _cabi.get_context(0)
plant, field, rec, src = scenes.nsttf_field()
cs = compile_scene(plant)
dev = DeviceScene(cs, ctx)
for n in (10000, 100000, 1000000):
    for stream in (False, True):
        ts = []
        for r in range(6):
            b = scenes.nsttf_source(n, src, seed=3, ray_offset=r * n)
            t0 = time.perf_counter()
            st, _ = dev.trace_fast(b, 100, 1e-10, 3, accel=False, stream=stream)
            ts.append((time.perf_counter() - t0) * 1e3)
        print('n %8d %-10s wall ms %s kernel_ms %.3f' % (n, 'stream' if stream else 'mega', ' '.join('%.2f' % t for t in ts), st.kernel_ms), flush=True)
