"""What coherent rays are worth on the mesh of tools/gpu_mesh.py: the same bundle given as arrays, once in the order of the source's
streams and once sorted by the cell of a 128 x 128 raster over the start points.  usage: gpu_mesh_sorted.py [rays, default 1e7]"""
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
from tracer_amd.ray_bundle import RayBundle


def height_field(m, extent=10., amp=0.8):
    """the relief of tools/gpu_mesh.py"""
    x, y = N.meshgrid(N.linspace(-extent, extent, m + 1), N.linspace(-extent, extent, m + 1), indexing='ij')
    z = amp * N.sin(1.3 * x) * N.cos(1.1 * y)
    V = N.c_[x.ravel(), y.ravel(), z.ravel()]
    i, j = N.meshgrid(N.arange(m), N.arange(m), indexing='ij')
    a, b, c, d = (i * (m + 1) + j).ravel(), ((i + 1) * (m + 1) + j).ravel(), ((i + 1) * (m + 1) + j + 1).ravel(), (i * (m + 1) + j + 1).ravel()
    return V, N.vstack((N.c_[a, b, c], N.c_[a, c, d]))


n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10000000
ctx = _cabi.get_context(0)
V, F = height_field(230)
mesh = TriangulatedSurface(V, F, opt.Reflective(0.2))
lid = AssembledObject(surfs=[Surface(RectPlateGM(90., 90.), opt.LambertianReceiver(1.))], transform=N.dot(translate(0., 0., 50.), rotx(N.pi)))
cs = compile_scene(Assembly(objects=[mesh, lid]))
dev = DeviceScene(cs, ctx)
direction = N.r_[0.1, -0.05, -1.] / N.linalg.norm([0.1, -0.05, -1.])
b = sources.buie_sunshape(n, N.c_[-40. * direction], direction, 12., 0.05, flux=1., seed=23)
v, d, e = N.array(b.get_vertices()), N.array(b.get_directions()), N.array(b.get_energy())
for name in ('stream order', 'sorted by start cell'):
    if name.startswith('sorted'):
        key = (N.clip(((v[0] + 16.) / 32. * 128).astype(int), 0, 127) * 128 + N.clip(((v[1] + 16.) / 32. * 128).astype(int), 0, 127))
        o = N.argsort(key, kind='stable')
        v, d, e = N.ascontiguousarray(v[:, o]), N.ascontiguousarray(d[:, o]), N.ascontiguousarray(e[o])
    for r in range(2):
        dev.reset_tallies()
        st, _ = dev.trace_fast(RayBundle(vertices=v, directions=d, energy=e), 6, 1e-10, 23, accel=True)
        a, rc, h = dev.get_tallies()
        print('%s run %d: kernels %.2f ms, %d segments (%.0f M segments/s), lid hits %d' % (name, r, st.kernel_ms, st.segments, st.segments / st.kernel_ms / 1e3, int(h[-1])), flush=True)
