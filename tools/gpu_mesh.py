"""a mesh as the reference builds it -- one Surface per face (models/triangulated_surface.py) -- beyond LDS: a relief of
105 800 triangles with a lid above it, Buie sun, mirror faces.  Times trc_trace_fast on it.  usage: gpu_mesh.py [rays, default 2e7]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as N
from tracer_amd import _cabi, sources
from tracer_amd import optics_callables as opt
from tracer_amd.assembly import Assembly
from tracer_amd.object import AssembledObject
from tracer_amd.surface import Surface
from tracer_amd.flat_surface import RectPlateGM
from tracer_amd.models.triangulated_surface import TriangulatedSurface
from tracer_amd.spatial_geometry import translate, rotx
from tracer_amd.scene import compile_scene, DeviceScene


def height_field(m, extent=10., amp=0.8):
    """m x m quads over [-extent, extent]^2 on a relief steep enough for second and third bounces, two triangles each:
    vertices (n, 3), faces (2 m^2, 3) -- the scene of tests/test_gpu_stream.py::test_mesh_of_1e5_triangles"""
    x, y = N.meshgrid(N.linspace(-extent, extent, m + 1), N.linspace(-extent, extent, m + 1), indexing='ij')
    z = amp * N.sin(1.3 * x) * N.cos(1.1 * y)
    V = N.c_[x.ravel(), y.ravel(), z.ravel()]
    i, j = N.meshgrid(N.arange(m), N.arange(m), indexing='ij')
    a, b, c, d = (i * (m + 1) + j).ravel(), ((i + 1) * (m + 1) + j).ravel(), ((i + 1) * (m + 1) + j + 1).ravel(), (i * (m + 1) + j + 1).ravel()
    return V, N.vstack((N.c_[a, b, c], N.c_[a, c, d]))


n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20000000
ctx = _cabi.get_context(0)
t0 = time.time()
V, F = height_field(int(os.environ.get("MESH_M", "230")))
mesh = TriangulatedSurface(V, F, opt.Reflective(0.2))
lid = AssembledObject(surfs=[Surface(RectPlateGM(90., 90.), opt.LambertianReceiver(1.))], transform=N.dot(translate(0., 0., 50.), rotx(N.pi)))
cs = compile_scene(Assembly(objects=[mesh, lid]))
t1 = time.time()
dev = DeviceScene(cs, ctx)
print('%d faces: %.2f s to build the mesh object and compile it, %.2f s to upload (boxes, 32-bit grid)' % (len(F), t1 - t0, time.time() - t1), flush=True)
direction = N.r_[0.1, -0.05, -1.] / N.linalg.norm([0.1, -0.05, -1.])
for r in range(3):
    b = sources.buie_sunshape(n, N.c_[-40. * direction], direction, float(os.environ.get("MESH_SRC_RADIUS", "12.")), 0.05, flux=1., seed=23 + r)
    t0 = time.time()
    st, _ = dev.trace_fast(b, 6, 1e-10, 23 + r, accel=True)
    wall = time.time() - t0
    a, rc, h = dev.get_tallies()
    print('run %d: %d rays, kernels %.2f ms, wall %.1f ms, %d segments (%.0f M segments/s by kernel time), %d launches; faces hit %d, lid hits %d' %
          (r, n, st.kernel_ms, wall * 1e3, st.segments, st.segments / st.kernel_ms / 1e3, st.launches, int((h[:-1] > 0).sum()), int(h[-1])), flush=True)
