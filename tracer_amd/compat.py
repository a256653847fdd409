"""
Drop-in aliasing: after `tracer_amd.compat.install()`, `import tracer.surface`, `from
tracer.models.heliostat_field import HeliostatField`, ... resolve to the tracer_amd modules of the
same name, so scene scripts written against casselineau/Tracer run unchanged on the GPU engine.
"""
import importlib
import sys
import types

_MODULES = ['assembly', 'object', 'surface', 'has_frame', 'geometry_manager', 'flat_surface', 'triangular_face',
            'quadric', 'paraboloid', 'sphere_surface', 'cylinder', 'cone', 'quadratic_surface', 'ellipsoid',
            'optics', 'optics_callables', 'ray_bundle', 'trace_tree', 'tracer_engine', 'sources',
            'spatial_geometry', 'boundary_shape', 'accel_tree', 'polygon', 'tracer_engine_mp', 'models',
            'models.one_sided_mirror', 'models.heliostat_field', 'models.homogenizer', 'models.spherical_lens',
            'models.triangulated_surface', 'models.homogenized_local_receiver', 'models.tau_minidish', 'models.PETAL_dish',
            'models.SG4']


def install(force=False):
    if 'tracer' in sys.modules and not force and not getattr(sys.modules['tracer'], '_tracer_amd_alias', False):
        raise RuntimeError("a different `tracer` package is already imported")
    root = importlib.import_module('tracer_amd')
    alias = types.ModuleType('tracer')
    alias.__path__ = []
    alias._tracer_amd_alias = True
    sys.modules['tracer'] = alias
    for name in _MODULES:
        mod = importlib.import_module('tracer_amd.' + name)
        sys.modules['tracer.' + name] = mod
        parent = alias if '.' not in name else sys.modules['tracer.' + name.rsplit('.', 1)[0]]
        setattr(parent, name.rsplit('.', 1)[-1], mod)
    # the top-level `emissive_losses` package of the reference (view factors, radiosity)
    for name in ('emissive_losses', 'emissive_losses.emissive_losses', 'emissive_losses.view_factors_3D'):
        sys.modules.setdefault(name, importlib.import_module('tracer_amd.' + name))
    # scene scripts import the Coin3D viewer with a star import; rendering is outside this package: a Renderer that says so
    coin = types.ModuleType('tracer.CoIn_rendering')
    coin.__path__ = []
    rendering = types.ModuleType('tracer.CoIn_rendering.rendering')

    class Renderer(object):
        def __init__(self, *args, **kwargs):
            raise NotImplementedError('tracer_amd has no Coin3D viewer: trace with it, render with the reference')
    rendering.Renderer = Renderer
    rendering.__all__ = ['Renderer']
    coin.rendering = rendering
    alias.CoIn_rendering = coin
    sys.modules['tracer.CoIn_rendering'] = coin
    sys.modules['tracer.CoIn_rendering.rendering'] = rendering
    rtu = types.ModuleType('ray_trace_utils')
    rtu.__path__ = []
    rtu.vector_manipulations = importlib.import_module('tracer_amd.vector_manipulations')
    sys.modules.setdefault('ray_trace_utils', rtu)
    sys.modules.setdefault('ray_trace_utils.vector_manipulations', rtu.vector_manipulations)
    rtu.stl_utils = importlib.import_module('tracer_amd.stl_utils')
    sys.modules.setdefault('ray_trace_utils.stl_utils', rtu.stl_utils)
    rtu.estimator = importlib.import_module('tracer_amd.estimator')
    sys.modules.setdefault('ray_trace_utils.estimator', rtu.estimator)
    return root
