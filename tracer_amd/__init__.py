"""
tracer_amd -- MI355X-native Monte-Carlo ray-tracing core behind Tracer's Python plugin API.

The hot path TracerEngine.ray_tracer() (reference: tracer/tracer_engine.py:124-295) runs in
hand-written HIP kernels for gfx950 reached through the C-ABI of include/tracer_amd.h; this package
is the host-side mirror of the reference's Assembly / Surface / GeometryManager / optics-callable
interface.  `tracer_amd.compat.install()` makes `import tracer.<module>` resolve to these modules
so existing scene scripts run unchanged.
"""
from .rng import seed  # noqa: F401

__version__ = '0.1.0'
