"""
ctypes binding of include/tracer_amd.h.

The library is the HIP build only (tracer_amd/lib/libtracer_amd.so, produced by `make` or
__graft_entry__.build()).  There is no CPU implementation behind this module: if the library is
missing, or no GPU is present when a context is created, the caller gets an exception.
"""
import ctypes as C
import weakref
import os
import threading

import numpy as N

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TRACER_AMD_LIB") or os.path.join(_HERE, "lib", "libtracer_amd.so")

TRC_BUIE_NELEM = 210
BUIE_LEN = 3 * (TRC_BUIE_NELEM + 1) + 6

# trace flags
# trc_status
OK, ERR_INVALID, ERR_DEVICE, ERR_UNSUPPORTED, ERR_CAPACITY, ERR_NOMEM = 0, -1, -2, -3, -4, -5
TRACE_ACCEL = 0x1
TRACE_KEEP_LAST = 0x2
BIN_ANGLE, BIN_HEIGHT, BIN_RADIUS, BIN_ROUND9, BIN_RADIUS_HALF_OPEN = 0x1, 0x2, 0x4, 0x8, 0x10      # trc_scene_bin_hits modes
TRACE_STREAM = 0x4
TRACE_MEGAKERNEL = 0x8
# surface flags
SURF_CAPTURE_HITS = 0x1
SURF_CAPTURE_LEAN = 0x2
LEVEL_VOLUME = 0x40000000        # TRC_LEVEL_VOLUME: bit of a level's surf[] entries (trc_result_level_get)

# enum trc_gm_kind
(GM_FLAT_INF, GM_RECT, GM_RECT_EXTRUDED, GM_RECT_PERFORATED, GM_ROUND, GM_ROUND_CUT, GM_TRIANGLE,
 GM_PARABOLOID, GM_PARAB_DISH, GM_PARAB_HEX, GM_PARAB_RECT, GM_PARAB_RECT_OFFAXIS, GM_PARAB_CYL,
 GM_PARAB_TROUGH, GM_SPHERE, GM_HEMISPHERE, GM_SPHERE_RECT, GM_CYL_INF, GM_CYL_FINITE, GM_CYL_RECTCUT,
 GM_CONE_INF, GM_CONE_FINITE, GM_FRUSTUM, GM_FRUSTUM_RECTCUT, GM_QUADRATIC, GM_QUADRATIC_RECT,
 GM_ELLIPSOID, GM_ELLIPSOID_CUT, GM_SPHERE_CUT, GM_POLYGON) = range(30)

# enum trc_optics_kind
(OPT_TRANSPARENT, OPT_REFLECTIVE, OPT_ONE_SIDED_REFLECTIVE, OPT_REAL_REFLECTIVE,
 OPT_ONE_SIDED_REAL_REFLECTIVE, OPT_LAMBERTIAN, OPT_LAMBERTIAN_SPECULAR, OPT_REFRACTIVE_HOMOGENOUS,
 OPT_REFLECTIVE_SPECTRAL, OPT_LAMBERTIAN_DIRECTIONAL, OPT_LAMBERTIAN_DIRECTIONAL_SPECTRAL,
 OPT_FRESNEL_CONDUCTOR, OPT_SEMI_LAMBERTIAN, OPT_REFRACTIVE_SCATTERING, OPT_REFRACTIVE_MATERIAL,
 OPT_LAMBERTIAN_POLYCHROMATIC, OPT_PERIODIC_BOUNDARY) = range(17)

# enum trc_source_kind
SRC_PILLBOX_DISK, SRC_PILLBOX_RECT, SRC_BUIE_DISK, SRC_BUIE_RECT, SRC_PILLBOX_TRIANGLE, SRC_VF_CYLINDER, SRC_VF_FRUSTUM = range(7)

_p_f64 = C.POINTER(C.c_double)
_p_i64 = C.POINTER(C.c_int64)
_p_u64 = C.POINTER(C.c_uint64)
_p_i32 = C.POINTER(C.c_int32)


class SurfaceDesc(C.Structure):
    _fields_ = [('gm_kind', C.c_int32), ('optics_kind', C.c_int32), ('flags', C.c_int32),
                ('extra_off', C.c_int32), ('extra_len', C.c_int32), ('reserved', C.c_int32),
                ('frame', C.c_double * 12), ('gm', C.c_double * 16), ('opt', C.c_double * 8)]


class Rays(C.Structure):
    _fields_ = [('n', C.c_int64), ('on_device', C.c_int32), ('n_spec', C.c_int32),
                ('x', _p_f64), ('y', _p_f64), ('z', _p_f64),
                ('dx', _p_f64), ('dy', _p_f64), ('dz', _p_f64), ('e', _p_f64),
                ('parent', _p_i64), ('ref_index', _p_f64), ('wavelength', _p_f64), ('rid', _p_u64),
                ('ref_index_im', _p_f64), ('spec_wl', _p_f64), ('spectra', _p_f64), ('n_mat', C.c_int64), ('mat', _p_f64)]


class SourceDesc(C.Structure):
    _fields_ = [('kind', C.c_int32), ('reserved', C.c_int32), ('center', C.c_double * 3),
                ('rot_pos', C.c_double * 9), ('rot_dir', C.c_double * 9), ('p', C.c_double * 8),
                ('energy', C.c_double), ('buie', C.c_double * BUIE_LEN)]


class KdTreeDesc(C.Structure):
    _fields_ = [('n_nodes', C.c_int32), ('n_leaf_surfs', C.c_int32), ('n_always', C.c_int32),
                ('reserved', C.c_int32), ('flag', _p_i32), ('split', _p_f64), ('child', _p_i32),
                ('leaf_off', _p_i32), ('leaf_cnt', _p_i32), ('leaf_surfs', _p_i32),
                ('always_relevant', _p_i32), ('bounds', C.c_double * 6)]


class TraceStats(C.Structure):
    _fields_ = [('segments', C.c_int64), ('hits', C.c_int64), ('rays_left', C.c_int64),
                ('hits_dropped', C.c_int64), ('energy_left', C.c_double), ('kernel_ms', C.c_double),
                ('bounces', C.c_int32), ('launches', C.c_int32)]


# every symbol include/tracer_amd.h declares: name -> (restype, argtypes)
_vp = C.c_void_p
_pvp = C.POINTER(C.c_void_p)
SIGNATURES = {
    'trc_last_error': (C.c_char_p, []),
    'trc_abi_version': (C.c_int, []),
    'trc_ctx_create': (C.c_int, [C.c_int, _pvp]),
    'trc_ctx_destroy': (C.c_int, [_vp]),
    'trc_ctx_synchronize': (C.c_int, [_vp]),
    'trc_ctx_device_name': (C.c_int, [_vp, C.c_char_p, C.c_int]),
    'trc_scene_create': (C.c_int, [_vp, C.c_int32, C.POINTER(SurfaceDesc), C.c_int32, _p_f64, _pvp]),
    'trc_scene_destroy': (C.c_int, [_vp]),
    'trc_scene_update_frames': (C.c_int, [_vp, C.c_int32, _p_f64]),
    'trc_scene_set_kdtree': (C.c_int, [_vp, C.POINTER(KdTreeDesc)]),
    'trc_scene_set_fluxmap': (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, _p_f64, _p_f64, _p_f64]),
    'trc_scene_set_hit_capacity': (C.c_int, [_vp, C.c_int64]),
    'trc_scene_clear_hits': (C.c_int, [_vp]),
    'trc_scene_reserve_hits': (C.c_int, [_vp, C.c_int64]),
    'trc_scene_hits_reserved': (C.c_int, [_vp, _p_i64, _p_i64]),
    'trc_host_alloc': (C.c_int, [C.c_int64, _pvp]),
    'trc_host_free': (C.c_int, [_vp]),
    'trc_scene_reset_tallies': (C.c_int, [_vp]),
    'trc_scene_get_tallies': (C.c_int, [_vp, _p_f64, _p_f64, _p_i64]),
    'trc_scene_get_fluxmap': (C.c_int, [_vp, C.c_int32, _p_f64]),
    'trc_scene_get_hits': (C.c_int, [_vp, _p_i64, _p_i32] + [_p_f64] * 8),
    'trc_scene_get_hits_x': (C.c_int, [_vp, _p_i64, _p_i32] + [_p_f64] * 8 + [C.c_int32, _p_f64]),
    'trc_scene_hit_spectral_columns': (C.c_int, [_vp, C.POINTER(C.c_int32)]),
    'trc_scene_bin_hits': (C.c_int, [_vp, C.c_int32, _p_i32, _p_i32, _p_f64, _p_i32, _p_f64]),
    'trc_scene_enable_transfer': (C.c_int, [_vp, C.c_int32]),
    'trc_scene_get_transfer': (C.c_int, [_vp, _p_f64]),
    'trc_scene_tally_size': (C.c_int, [_vp, _p_i64]),
    'trc_scene_export_tallies': (C.c_int, [_vp, _vp, C.c_int32]),
    'trc_scene_import_tallies': (C.c_int, [_vp, _vp, C.c_int32]),
    'trc_trace_fast': (C.c_int, [_vp, C.POINTER(Rays), C.POINTER(SourceDesc), C.c_int64, C.c_int32, C.c_double,
                                 C.c_uint64, C.c_uint64, C.c_int32, C.POINTER(Rays), C.POINTER(TraceStats)]),
    'trc_trace_ordered': (C.c_int, [_vp, C.POINTER(Rays), C.POINTER(SourceDesc), C.c_int64, C.c_int32, C.c_double,
                                    C.c_uint64, C.c_uint64, C.c_int32, _pvp, C.POINTER(TraceStats)]),
    'trc_result_num_levels': (C.c_int, [_vp, _p_i32]),
    'trc_result_level_size': (C.c_int, [_vp, C.c_int32, _p_i64, _p_i64]),
    'trc_result_level_get': (C.c_int, [_vp, C.c_int32, C.POINTER(Rays), _p_i32]),
    'trc_result_destroy': (C.c_int, [_vp]),
    'trc_source_generate': (C.c_int, [_vp, C.POINTER(SourceDesc), C.c_int64, C.c_uint64, C.c_uint64, C.POINTER(Rays)]),
    'trc_source_start32': (C.c_int, [_vp, C.POINTER(SourceDesc), C.c_int64, C.c_uint64, C.c_uint64, C.POINTER(C.c_float),
                                     C.POINTER(C.c_float), _p_f64]),
    'trc_gm_find_intersections': (C.c_int, [_vp, C.POINTER(SurfaceDesc), C.c_int32, _p_f64, C.POINTER(Rays), _p_f64,
                                            _p_f64, _p_f64, _p_f64]),
    'trc_kdtree_traversal': (C.c_int, [_vp, C.POINTER(KdTreeDesc), C.c_int32, C.POINTER(Rays), C.c_int64, C.POINTER(C.c_uint8),
                                       C.POINTER(C.c_int32)]),
    'trc_gm_get_normals': (C.c_int, [_vp, C.POINTER(SurfaceDesc), C.c_int64] + [_p_f64] * 9),
    'trc_optics_apply': (C.c_int, [_vp, C.POINTER(SurfaceDesc), C.c_int32, _p_f64, C.POINTER(Rays)] + [_p_f64] * 6 +
                         [C.c_uint64, C.c_int32, C.POINTER(Rays)]),
    'trc_optics_fresnel_attenuating': (C.c_int, [_vp, C.c_int64, C.c_double] + [_p_f64] * 6),
}


class TracerAmdError(RuntimeError):
    """A C-ABI call failed.  `status` is the trc_status code."""
    def __init__(self, status, message):
        RuntimeError.__init__(self, message)
        self.status = status


class NativeLibraryMissing(TracerAmdError):
    pass


_lib = None
_lib_lock = threading.Lock()


def load_library(path=None):
    """
    Load the HIP library and bind every entry point of the header.  Raises NativeLibraryMissing if
    the shared object has not been built -- deliberately no other implementation is tried.
    """
    global _lib
    with _lib_lock:
        if _lib is not None and path is None:
            return _lib
        p = path or LIB_PATH
        if not os.path.exists(p):
            raise NativeLibraryMissing(-2, "%s not found: build the HIP extension first (`make` at the repository "
                                       "root or __graft_entry__.build()); tracer_amd has no CPU path" % p)
        lib = C.CDLL(p)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)   # AttributeError here means the .so is stale vs the header
            fn.restype = res
            fn.argtypes = args
        if lib.trc_abi_version() != 2:
            raise TracerAmdError(-1, "ABI version mismatch: library %d, binding 2" % lib.trc_abi_version())
        if path is None:
            _lib = lib
        return lib


def check(status):
    if status != 0:
        msg = load_library().trc_last_error()
        msg = msg.decode('utf-8', 'replace') if msg else 'status %d' % status
        if status == -1:
            raise TracerAmdValueError(status, msg)
        if status == -3:
            raise TracerAmdUnsupported(status, msg)
        raise TracerAmdError(status, msg)


class TracerAmdValueError(TracerAmdError, ValueError):
    """Bad argument: the reference raises ValueError for these (e.g. flat_surface.py:192-195)."""


class TracerAmdUnsupported(TracerAmdError, NotImplementedError):
    """A geometry/optics/source kind that is not in the native table."""


# ------------------------------------------------------------------------------------------------
# array helpers
# ------------------------------------------------------------------------------------------------
def f64(a):
    """C-contiguous float64 view/copy of a."""
    return N.ascontiguousarray(a, dtype=N.float64)


PINNED_FROM = 1 << 20       # bytes from which a result array is placed in page-locked memory (pinned_empty)


def pinned_empty(shape, dtype=N.float64):
    """
    An uninitialised array for a large result of the library -- a hit list, a level of the ray tree -- in page-locked host
    memory (trc_host_alloc): the device-to-host copy into it runs at the rate of the link instead of through the driver's
    bounce buffers.  The block goes back to the library's pool when the last view of the array is gone.  Small arrays
    are ordinary numpy arrays.
    """
    dtype = N.dtype(dtype)
    shape = (shape,) if N.isscalar(shape) else tuple(shape)
    nbytes = int(N.prod(shape, dtype=N.int64)) * dtype.itemsize
    if nbytes < PINNED_FROM:
        return N.empty(shape, dtype=dtype)
    lib = load_library()
    p = C.c_void_p()
    check(lib.trc_host_alloc(nbytes, C.byref(p)))
    buf = (C.c_char * nbytes).from_address(p.value)
    weakref.finalize(buf, lib.trc_host_free, C.c_void_p(p.value))
    return N.frombuffer(buf, dtype=dtype).reshape(shape)


def ptr(a, typ=_p_f64):
    if a is None:
        return typ()
    return a.ctypes.data_as(typ)


def make_rays(n, x=None, y=None, z=None, dx=None, dy=None, dz=None, e=None, parent=None, ref_index=None,
              wavelength=None, rid=None, ref_index_im=None, spec_wl=None, spectra=None, mat=None):
    """Fill a Rays struct from 1-D contiguous arrays (the caller keeps them alive); spec_wl, spectra: C-contiguous (W, n)."""
    r = Rays()
    r.n = n
    r.on_device = 0
    r.x, r.y, r.z = ptr(x), ptr(y), ptr(z)
    r.dx, r.dy, r.dz = ptr(dx), ptr(dy), ptr(dz)
    r.e = ptr(e)
    r.parent = ptr(parent, _p_i64)
    r.ref_index = ptr(ref_index)
    r.wavelength = ptr(wavelength)
    r.rid = ptr(rid, _p_u64)
    r.ref_index_im = ptr(ref_index_im)
    if spectra is not None:
        assert spec_wl is not None and spec_wl.shape == spectra.shape and spectra.ndim == 2 and spectra.shape[1] >= n
        assert spectra.flags.c_contiguous and spec_wl.flags.c_contiguous
        r.n_spec = spectra.shape[0]
        r.spec_wl, r.spectra = ptr(spec_wl), ptr(spectra)
    if mat is not None:
        assert mat.ndim == 2 and mat.shape[0] % 2 == 0 and mat.shape[1] >= n and mat.flags.c_contiguous
        r.n_mat = mat.shape[0] // 2
        r.mat = ptr(mat)
    return r


# ------------------------------------------------------------------------------------------------
# context: one per process and GPU, created lazily
# ------------------------------------------------------------------------------------------------
class Context(object):
    def __init__(self, device_id=0):
        self.lib = load_library()
        h = C.c_void_p()
        check(self.lib.trc_ctx_create(int(device_id), C.byref(h)))
        self.handle = h
        self.device_id = device_id

    def device_name(self):
        buf = C.create_string_buffer(256)
        check(self.lib.trc_ctx_device_name(self.handle, buf, 256))
        return buf.value.decode()

    def synchronize(self):
        check(self.lib.trc_ctx_synchronize(self.handle))

    def close(self):
        if self.handle is not None and self.handle.value:
            self.lib.trc_ctx_destroy(self.handle)
            self.handle = None


_contexts = {}
_default_device = None


def set_default_device(device_id):
    global _default_device
    _default_device = int(device_id)


def default_device():
    if _default_device is not None:
        return _default_device
    return int(os.environ.get('LOCAL_RANK', '0'))


def _torch_first():
    """
    PyTorch-ROCm brings its own copy of the HIP runtime.  Whichever of the two runtimes in the process is initialised second
    sees the GPUs only if it is torch's that came first; the other way round torch reports "No HIP GPUs are available"
    (INTEGRATION.md).  So when torch is already imported -- a job that will exchange tallies through torch.distributed --
    its runtime is initialised here, before the library's context exists.
    """
    import sys
    torch = sys.modules.get('torch')
    if torch is None or _contexts:
        return
    try:
        cuda = torch.cuda
        if cuda.is_available() and not cuda.is_initialized():
            cuda.init()
    except Exception:       # a CPU-only torch, or no GPU: nothing to order
        pass


def check_torch_order():
    """
    Called before a device tensor of torch meets the library (distributed.reduce_scene_tallies, nccl): a loud error instead of
    torch's "No HIP GPUs are available" when torch was imported after the library's context had been created.
    """
    import sys
    torch = sys.modules.get('torch')
    if torch is None:
        return
    ok = False
    try:
        ok = torch.cuda.is_available()
    except Exception:
        ok = False
    if not ok and _contexts:
        raise RuntimeError(
            "torch cannot see the GPU: the HIP runtime of libtracer_amd.so was initialised before torch's own copy of it. "
            "Import torch (and call torch.cuda.init() or torch.cuda.set_device(...)) before the first tracer_amd context is "
            "created -- tracer_amd._cabi.get_context() does this by itself when torch is already imported -- or exchange the "
            "tallies through the host (backend 'gloo').")


def get_context(device_id=None):
    if device_id is None:
        device_id = default_device()
    ctx = _contexts.get(device_id)
    if ctx is None:
        _torch_first()
        ctx = Context(device_id)
        _contexts[device_id] = ctx
    return ctx
