"""configs[4] on one rank: dish with slope error -> spectral cavity, rays with wavelengths handed over as a bundle (host arrays).
usage: gpu_cavity.py [rays, default 1.25e8: a rank's share of 1e9 over 8 GPUs]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as N
from tracer_amd import _cabi, scenes
from tracer_amd.scene import DeviceScene
from tracer_amd.ray_bundle import RayBundle
ctx = _cabi.get_context(0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 125000000
ts, src = scenes.dish_cavity()
t0 = time.time()
b0 = scenes.dish_source(n, src, seed=9)
v, d, e = N.asarray(b0.get_vertices()), N.asarray(b0.get_directions()), N.asarray(b0.get_energy())
wl = N.random.default_rng(4).uniform(0.3e-6, 2.5e-6, n)
print('bundle of %d rays on the host (generated on the device, wavelengths drawn here): %.1f s' % (n, time.time() - t0), flush=True)
dev = DeviceScene(ts, ctx)
for r in range(2):
    dev.reset_tallies()
    t0 = time.time()
    st, _ = dev.trace_fast(RayBundle(vertices=v, directions=d, energy=e, wavelengths=wl), 12, 1e-3 * e[0], 31, stream=os.environ.get("CAVITY_MEGAKERNEL", "0") != "1")
    wall = time.time() - t0
    a, rcv, h = dev.get_tallies()
    print('run %d: kernels %8.2f ms  wall (incl. 7 columns over PCIe) %8.1f ms  %8.1f Mseg/s by kernel time  segments %d  hits %s  absorbed share %.4f  launches %d' %
          (r, st.kernel_ms, wall * 1e3, st.segments / st.kernel_ms / 1e3, st.segments, h.tolist(), a.sum() / e.sum(), st.launches), flush=True)
dev.close()
