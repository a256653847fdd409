"""configs[1]: parabolic dish + round receiver, Buie sunshape, 1e7 rays -- both forms of the fast engine."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as N
from tracer_amd import _cabi, scenes
from tracer_amd.scene import compile_scene, DeviceScene
ctx = _cabi.get_context(0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10 ** 7
asm, dish_s, rec_s, src = scenes.dish()
cs = compile_scene(asm)
for stream in (False, True):
    dev = DeviceScene(cs, ctx)
    for r in range(3):
        t0 = time.time()
        st, _ = dev.trace_fast(scenes.dish_source(n, src, seed=3), 100, 1e-10, 3, accel=True, stream=stream)
        wall = time.time() - t0
    a, rcv, h = dev.get_tallies()
    print('%-10s kernel %8.3f ms  wall %8.3f ms  %8.1f Mseg/s  segments %d  receiver %.2f W (3 runs)' %
          ('streaming' if stream else 'megakernel', st.kernel_ms, wall * 1e3, st.segments / st.kernel_ms / 1e3, st.segments, a[1]), flush=True)
    dev.close()
