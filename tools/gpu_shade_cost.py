"""What the receiver's flux map and hit capture cost on the bench workload: kernel time of 1e8 NSTTF rays with neither, either, both."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as N
from tracer_amd import _cabi, scenes
from tracer_amd.scene import compile_scene, DeviceScene

ctx = _cabi.get_context(0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000000
plant, field, rec, src = scenes.nsttf_field()
cs = compile_scene(plant)
ue, ve = scenes.nsttf_fluxmap_edges()
for name, fm, cap in (('neither', False, False), ('flux map', True, False), ('hit capture', False, True), ('both (the bench)', True, True)):
    dev = DeviceScene(cs, ctx)
    if fm:
        dev.set_fluxmap(218, ue, ve)
    if cap:
        dev.set_hit_capacity(int(0.08 * n) + 4096)
    best = 1e9
    for r in range(4):
        if cap:
            dev.lib.trc_scene_clear_hits(dev.handle)
        st, _ = dev.trace_fast(scenes.nsttf_source(n, src, seed=7 + r), 100, 1e-10, 7 + r, accel=True)
        best = min(best, st.kernel_ms)
    print('%-20s %8.3f ms  %8.1f Mseg/s  dropped %d' % (name, best, st.segments / best / 1e3, st.hits_dropped), flush=True)
    dev.close()
