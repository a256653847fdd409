"""
Scene compiler and device-scene handle.

compile_scene() flattens Assembly.get_surfaces() (reference ordering contract assembly.py:60-77)
into the trc_surface_desc table of include/tracer_amd.h: one row per Surface with its global frame
(Surface._temp_frame), its geometry kind/parameters and its optics kind/parameters.  DeviceScene
owns the trc_scene handle and wraps the tally / flux-map / hit-buffer calls.

A surface whose geometry manager or optics callable is not in the native table makes
compile_scene raise NotNativeError; TracerEngine then drives such scenes through the per-surface
protocol (engine='protocol'), in which native kinds still run on the GPU.
"""
import ctypes as C

import numpy as N

from . import _cabi
from .geometry_manager import NativeGeometryManager, fill_desc
from .deferred import Delivery
from .face_set import FaceSet, SurfaceSeq
from .optics_callables import OpticsCallable, native_optics_of, LocationAccountant, DirectionAccountant, \
    AbsorptionAccountant, ReceptionAccountant, ScatteringAccountant, NormalAccountant


_DESC_DTYPE = N.dtype([('ints', N.int32, 6), ('frame', N.float64, 12), ('gm', N.float64, 16), ('opt', N.float64, 8)])
assert _DESC_DTYPE.itemsize == C.sizeof(_cabi.SurfaceDesc)


def material_rows(materials, wavelengths):
    """trc_rays.mat: (2K, n), rows 2k, 2k+1 = Re, Im of materials[k].m(wavelengths)"""
    wl = N.asarray(wavelengths, dtype=float)
    out = N.empty((2 * len(materials), len(wl)))
    for k, mat in enumerate(materials):
        with N.errstate(all='ignore'):
            m = N.asarray(mat.m(wl), dtype=complex)
        out[2 * k], out[2 * k + 1] = m.real, m.imag
    return out


class NotNativeError(NotImplementedError):
    """The scene holds a geometry/optics plug-in that only exists in Python."""


class CompiledScene(object):
    def __init__(self, surfaces):
        # (the faces of a mesh kept as arrays stay arrays: face_set.SurfaceSeq -- everything else is a list of Surfaces)
        self.surfaces = surfaces if isinstance(surfaces, SurfaceSeq) else list(surfaces)
        n = len(self.surfaces)
        if n == 0:
            raise ValueError("the assembly has no surfaces")
        # the table is filled row by row into one array laid out like trc_surface_desc (6 int32, then frame 12, gm 16, opt 8 doubles)
        # and copied over the ctypes array at the end: this runs once per call of ray_tracer, to see whether the scene changed
        rows = N.zeros(n, dtype=_DESC_DTYPE)
        self._extra = []
        self.splits = False
        self.carries = False        # optics that read what only rays of the ordered engine carry (complex indices, spectra)
        self.materials = []         # materials of the Refractive surfaces; row k of trc_rays.mat is materials[k].m(wavelengths)
        self._sig_parts = []        # (first row, rows, key) of the parts whose rows a short key stands for (the faces of a mesh)
        self.capture = [False] * n
        self.optics = []            # the distinct optics managers of the scene, in surface order; those that capture hits
        self.capturing_optics = []
        self._seen = set()
        segments = self.surfaces.segments() if isinstance(self.surfaces, SurfaceSeq) else [(0, self.surfaces)]
        for first, part in segments:
            if isinstance(part, FaceSet):           # all faces of a mesh at once: one optics (or one recipe), one kind of geometry
                m = len(part)
                okind, opar, sflags = self._optics_row(first, part.optics_template(), no_tables=True)
                if part.optics is None and (sflags & _cabi.SURF_CAPTURE_HITS):
                    # every face its own accountants: they have to exist before hits are promised to them
                    for s in part:
                        self._note_optics(s.get_optics_manager(), True)
                # the rows of the faces as they stand -- pose of the object, parameters of the optics -- are kept by the FaceSet: a
                # script that traces the same mesh again and again pays one copy per call, and the table's signature a few bytes
                key = (part._parent.tobytes(), int(okind), tuple(float(x) for x in opar), int(sflags))
                block = part.compiled_rows(key)
                if block is None:
                    gkind, gpar = part.gm_rows()
                    block = N.zeros(m, dtype=_DESC_DTYPE)
                    block['ints'] = (gkind, okind, sflags, -1, 0, 0)
                    block['frame'] = part.global_frames12()
                    block['gm'][:, :gpar.shape[1]] = gpar
                    if len(opar):
                        block['opt'][:, :len(opar)] = opar
                    part.keep_compiled_rows(key, block)
                rows[first:first + m] = block
                self._sig_parts.append((first, m, ('faces', id(part), part.stamp) + key))
                if sflags & _cabi.SURF_CAPTURE_HITS:
                    self.capture[first:first + m] = [True] * m
                continue
            for i, s in enumerate(part, first):
                gm = s.get_geometry_manager()
                if not isinstance(gm, NativeGeometryManager):
                    raise NotNativeError("surface %d: geometry manager %s is not in the native table" % (i, type(gm).__name__))
                gkind, gpar, gextra = gm._native()
                okind, opar, sflags, oextra = self._optics_row(i, s.get_optics_manager())
                if len(gextra) and len(oextra):
                    raise NotNativeError("surface %d: both geometry and optics carry tables" % i)
                ex = list(gextra) if len(gextra) else list(oextra)
                off = len(self._extra) if len(ex) else -1
                self._extra.extend(ex)
                self.capture[i] = bool(sflags & _cabi.SURF_CAPTURE_HITS)
                row = rows[i]
                row['ints'] = (gkind, okind, sflags, off, len(ex), 0)
                row['frame'] = N.asarray(s._temp_frame, dtype=float)[:3].ravel()
                if len(gpar):
                    row['gm'][:len(gpar)] = gpar
                if len(opar):
                    row['opt'][:len(opar)] = opar
        self._rows = rows                                   # (the ctypes table is a view of this array: no second copy of 38 MB for a mesh)
        self.descs = (_cabi.SurfaceDesc * n).from_buffer(rows)
        self.extra = _cabi.f64(self._extra)
        del self._extra, self._seen
        self.n_surf = n

    def _optics_row(self, i, opt, no_tables=False):
        """(kind, parameters, surface flags[, table]) of an optics manager, and what it means for the scene as a whole"""
        nat = native_optics_of(opt)
        if nat is None:
            raise NotNativeError("surface %d: optics %s is not in the native table" % (i, type(opt).__name__))
        try:
            okind, opar, oextra = nat._native()
        except NotImplementedError as err:
            raise NotNativeError("surface %d: %s" % (i, err))
        if okind == _cabi.OPT_REFRACTIVE_MATERIAL:      # opt[4], opt[5]: the scene's rows of this surface's two materials
            opar = list(opar)
            for j, mat in enumerate(nat._materials):
                known = [k for k, m in enumerate(self.materials) if m is mat]
                if not known:
                    self.materials.append(mat)
                    known = [len(self.materials) - 1]
                opar[4 + j] = float(known[0])
        if okind in (_cabi.OPT_REFRACTIVE_MATERIAL, _cabi.OPT_LAMBERTIAN_POLYCHROMATIC):
            self.carries = True
        if getattr(nat, '_splits', False):
            self.splits = True
        wants_hits = isinstance(opt, OpticsCallable) and len(opt.accountants) > 0
        self._note_optics(opt, wants_hits)
        sflags = _cabi.SURF_CAPTURE_HITS if wants_hits else 0
        # "Receiver" classes (absorbed energy + hit points): the device leaves incident energy and direction of their hits out
        if wants_hits and all(type(a) in (AbsorptionAccountant, LocationAccountant) for a in opt.accountants):
            sflags |= _cabi.SURF_CAPTURE_LEAN
        if no_tables:
            if len(oextra):
                raise NotNativeError("surface %d: the faces of a mesh share optics that carry a table (%s)" % (i, type(opt).__name__))
            return okind, opar, sflags
        return okind, opar, sflags, oextra

    def _note_optics(self, opt, captures):
        if id(opt) not in self._seen:
            self._seen.add(id(opt))
            self.optics.append(opt)
            if captures:
                self.capturing_optics.append(opt)

    def _sig(self, rows_bytes):
        """what identifies the table: the rows of the surfaces compiled one by one as bytes, a short key for the faces of a mesh"""
        out, at = [], 0
        for first, m, key in self._sig_parts:
            if first > at:
                out.append(rows_bytes(at, first))
            out.append(key)
            at = first + m
        if at < self.n_surf:
            out.append(rows_bytes(at, self.n_surf))
        out.append(self.extra.tobytes())
        return tuple(out)

    def signature(self):
        """What identifies everything uploaded to the device (compared with ==)."""
        return self._sig(lambda a, b: self._rows[a:b].tobytes())

    def signature_without_frames(self):
        """The same with the surface frames left out: equal for two states of a scene that only moved (a heliostat field
        following the sun) -- then DeviceScene.update_frames is enough."""
        def part(a, b):
            rows = self._rows[a:b]
            return rows['ints'][:, :5].tobytes() + rows['gm'].tobytes() + rows['opt'].tobytes()
        # (the key of a mesh's faces holds the pose of their object: left out here)
        sig = self._sig(part)
        return tuple(k[:3] + k[4:] if isinstance(k, tuple) and k and k[0] == 'faces' else k for k in sig)

    def frames12(self):
        fr = N.empty((self.n_surf, 12))
        segments = self.surfaces.segments() if isinstance(self.surfaces, SurfaceSeq) else [(0, self.surfaces)]
        for first, part in segments:
            if isinstance(part, FaceSet):
                fr[first:first + len(part)] = part.global_frames12()
            else:
                for i, s in enumerate(part, first):
                    fr[i] = N.asarray(s._temp_frame, dtype=float)[:3].ravel()
        return fr


class TableScene(object):
    """
    A scene given directly as arrays (kinds, frames, parameters) instead of Surface objects -- the form
    in which the golden fixtures store scenes.  Same attributes as CompiledScene where the engines need them.
    """
    def __init__(self, gm_kind, optics_kind, frames, gm, opt, extra, extra_off, extra_len, flags=None):
        n = len(gm_kind)
        self.n_surf = n
        self.surfaces = None
        self.descs = (_cabi.SurfaceDesc * n)()
        self.extra = _cabi.f64(extra)
        self.splits = False
        self.carries = False
        self.materials = []
        self.capture = [False] * n
        self.optics, self.capturing_optics = [], []
        for i in range(n):
            fill_desc(self.descs[i], N.asarray(frames[i]), int(gm_kind[i]), list(gm[i]), int(optics_kind[i]), list(opt[i]),
                      flags=0 if flags is None else int(flags[i]), extra_off=int(extra_off[i]), extra_len=int(extra_len[i]))
            if optics_kind[i] == _cabi.OPT_REFRACTIVE_HOMOGENOUS and opt[i][2] == 0.:
                self.splits = True

    def signature(self):
        return bytes(self.descs) + self.extra.tobytes()


def scene_arrays(compiled):
    """dict of plain arrays describing a CompiledScene / TableScene (what fixtures store)."""
    d = compiled.descs
    n = compiled.n_surf
    return dict(gm_kind=N.array([d[i].gm_kind for i in range(n)], dtype=N.int32),
                optics_kind=N.array([d[i].optics_kind for i in range(n)], dtype=N.int32),
                frames=N.array([N.vstack((N.array(list(d[i].frame)).reshape(3, 4), [0., 0., 0., 1.])) for i in range(n)]),
                gm=N.array([list(d[i].gm) for i in range(n)]), opt=N.array([list(d[i].opt) for i in range(n)]),
                extra=N.array(compiled.extra, dtype=float),
                extra_off=N.array([d[i].extra_off for i in range(n)], dtype=N.int32),
                extra_len=N.array([d[i].extra_len for i in range(n)], dtype=N.int32))


def compile_scene(assembly_or_surfaces):
    surfaces = assembly_or_surfaces.get_surfaces() if hasattr(assembly_or_surfaces, 'get_surfaces') \
        else assembly_or_surfaces
    return CompiledScene(surfaces)


class DeviceScene(object):
    """trc_scene handle."""
    def __init__(self, compiled, ctx=None):
        self.ctx = ctx or _cabi.get_context()
        self.lib = self.ctx.lib
        self.compiled = compiled
        h = C.c_void_p()
        _cabi.check(self.lib.trc_scene_create(self.ctx.handle, compiled.n_surf, compiled.descs, len(compiled.extra),
                                              _cabi.ptr(compiled.extra) if len(compiled.extra) else None, C.byref(h)))
        self.handle = h
        self.n_surf = compiled.n_surf
        self.fluxmaps = {}
        self.hit_capacity = 0
        self.form_rate = {}         # segments per ms of kernel time seen from each form of the fast engine on this scene (TracerEngine)
        self._kd_keep = None
        self.pending_hits = None    # PendingHits: what the hit buffer holds for accountants that have not read it yet
        self._has_scattering = None
        self.capture_rate = None    # largest share of captured hits per ray of a fast trace seen on the scene in its present poses

    def close(self):
        if self.handle is not None and self.handle.value:
            try:
                self.settle_pending()
            finally:
                self.lib.trc_scene_destroy(self.handle)
                self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def update_frames(self, compiled):
        """New poses for the same surfaces (trc_scene_update_frames): boxes and grid are rebuilt by the library, a Kd-tree
        set before is dropped (it described the old poses); tallies, flux maps and the hit buffer stay."""
        self.settle_pending()       # (normals of hits that wait for a NormalAccountant are those of the poses they were made in)
        self.capture_rate = None
        fr = _cabi.f64(compiled.frames12())
        _cabi.check(self.lib.trc_scene_update_frames(self.handle, compiled.n_surf, _cabi.ptr(fr)))
        self.compiled = compiled
        self._kd_keep = None

    # -- acceleration ---------------------------------------------------------------------------
    def set_kdtree(self, tree):
        """tree: accel_tree.KdTree or None."""
        if tree is None:
            _cabi.check(self.lib.trc_scene_set_kdtree(self.handle, None))
            self._kd_keep = None
            return
        f = tree.flat()
        d = _cabi.KdTreeDesc()
        d.n_nodes = len(f['flag'])
        d.n_leaf_surfs = len(f['leaf_surfs'])
        d.n_always = len(f['always_relevant'])
        i32 = C.POINTER(C.c_int32)
        d.flag = f['flag'].ctypes.data_as(i32)
        d.split = _cabi.ptr(f['split'])
        d.child = f['child'].ctypes.data_as(i32)
        d.leaf_off = f['leaf_off'].ctypes.data_as(i32)
        d.leaf_cnt = f['leaf_cnt'].ctypes.data_as(i32)
        d.leaf_surfs = f['leaf_surfs'].ctypes.data_as(i32)
        d.always_relevant = f['always_relevant'].ctypes.data_as(i32)
        for i in range(6):
            d.bounds[i] = f['bounds'][i]
        self._kd_keep = f
        _cabi.check(self.lib.trc_scene_set_kdtree(self.handle, C.byref(d)))

    # -- tallies ----------------------------------------------------------------------------------
    def set_fluxmap(self, surf_index, u_edges, v_edges, proj=None):
        """
        Accumulate absorbed energy of surface `surf_index` on the grid u_edges x v_edges of its local
        x, y.  proj defaults to round(inv(frame), 9) (Surface.global_to_local, surface.py:125).
        """
        u = _cabi.f64(u_edges)
        v = _cabi.f64(v_edges)
        if proj is None:
            proj = N.round(N.linalg.inv(self.compiled.surfaces[surf_index]._temp_frame), decimals=9)
        p = _cabi.f64(N.asarray(proj)[:3].ravel())
        _cabi.check(self.lib.trc_scene_set_fluxmap(self.handle, surf_index, len(u) - 1, len(v) - 1, _cabi.ptr(u),
                                                   _cabi.ptr(v), _cabi.ptr(p)))
        self.fluxmaps[surf_index] = (len(u) - 1, len(v) - 1)

    def get_fluxmap(self, surf_index):
        nu, nv = self.fluxmaps[surf_index]
        out = N.empty((nu, nv))
        _cabi.check(self.lib.trc_scene_get_fluxmap(self.handle, surf_index, _cabi.ptr(out)))
        return out

    def set_hit_capacity(self, capacity):
        """an empty hit buffer of this capacity (what it held is delivered first to the accountants that wait for it)"""
        capacity = int(capacity)
        self.settle_pending()
        if capacity != self.hit_capacity:
            _cabi.check(self.lib.trc_scene_set_hit_capacity(self.handle, capacity))
            self.hit_capacity = capacity

    def compiled_has_scattering(self):
        if self._has_scattering is None:
            self._has_scattering = bool((N.frombuffer(self.compiled.descs, dtype=_DESC_DTYPE)['ints'][:, 1] == _cabi.OPT_REFRACTIVE_SCATTERING).any())
        return self._has_scattering

    def hits_reserved(self):
        """(entries of the hit buffer reserved so far, capacity last asked for)"""
        a, b = C.c_int64(0), C.c_int64(0)
        _cabi.check(self.lib.trc_scene_hits_reserved(self.handle, C.byref(a), C.byref(b)))
        return a.value, b.value

    def reserve_hits(self, capacity):
        """room for `capacity` hits in all, keeping the hits the buffer holds"""
        _cabi.check(self.lib.trc_scene_reserve_hits(self.handle, int(capacity)))
        self.hit_capacity = self.hits_reserved()[1]

    def settle_pending(self):
        """hits the fast engine left in the device buffer for accountants that have not read them yet are delivered now
        (before the buffer is emptied, re-sized or the scene goes away)"""
        p = self.pending_hits
        self.pending_hits = None
        if p is not None:
            p.settle()

    def reset_tallies(self):
        self.settle_pending()           # (the reset empties the hit buffer)
        _cabi.check(self.lib.trc_scene_reset_tallies(self.handle))

    def get_tallies(self):
        a = N.empty(self.n_surf)
        r = N.empty(self.n_surf)
        h = N.empty(self.n_surf, dtype=N.int64)
        _cabi.check(self.lib.trc_scene_get_tallies(self.handle, _cabi.ptr(a), _cabi.ptr(r),
                                                   h.ctypes.data_as(C.POINTER(C.c_int64))))
        return a, r, h

    def get_hits(self):
        """dict of the captured hits: device arrival order for one capturing surface, surface by surface (arrival order inside) for several.  When every capturing surface is captured lean (Receiver accountants:
        absorbed energy + hit point) the columns the device did not write are not fetched either: `e_in` is the absorbed energy,
        `directions` is None."""
        n = C.c_int64(0)
        nul = C.POINTER(C.c_double)()
        # room for every entry reserved so far (the written hits and what the open chunks of the streaming engine leave unused):
        # one call packs, counts and copies, and the arrays are cut to the count afterwards
        k = self.hits_reserved()[0]
        fl = [self.compiled.descs[i].flags for i in range(self.n_surf)]
        lean = all((f & _cabi.SURF_CAPTURE_LEAN) for f in fl if (f & _cabi.SURF_CAPTURE_HITS)) and any(f & _cabi.SURF_CAPTURE_HITS for f in fl)
        # (large lists land in page-locked memory: 6.5e6 hits of 36 bytes cross the link in 5 ms, not 15)
        surf = _cabi.pinned_empty(k, dtype=N.int32)
        e_abs, points = _cabi.pinned_empty(k), _cabi.pinned_empty((3, k))
        e_in = e_abs if lean else _cabi.pinned_empty(k)
        directions = None if lean else _cabi.pinned_empty((3, k))
        cols = [e_abs, None if lean else e_in, points[0], points[1], points[2]] + ([None] * 3 if lean else [directions[0], directions[1], directions[2]])   # rows: no copy afterwards
        # hits of polychromatic rays bring 3 W more columns: sample wavelengths, the spectrum that arrived, the one that left
        nx = C.c_int32(0)
        _cabi.check(self.lib.trc_scene_hit_spectral_columns(self.handle, C.byref(nx)))
        nx = nx.value
        x = _cabi.pinned_empty((nx, k)) if (nx and k) else None
        if k and x is not None:
            _cabi.check(self.lib.trc_scene_get_hits_x(self.handle, C.byref(n), surf.ctypes.data_as(C.POINTER(C.c_int32)),
                                                      *([(_cabi.ptr(c) if c is not None else nul) for c in cols] + [nx, _cabi.ptr(x)])))
        elif k:
            _cabi.check(self.lib.trc_scene_get_hits(self.handle, C.byref(n), surf.ctypes.data_as(C.POINTER(C.c_int32)),
                                                    *[(_cabi.ptr(c) if c is not None else nul) for c in cols]))
        m = n.value
        surf, e_abs, points = surf[:m], e_abs[:m], points[:, :m]
        e_in = e_abs if lean else e_in[:m]
        directions = None if lean else directions[:, :m]
        out = dict(surf=surf, e_abs=e_abs, e_in=e_in, points=points, directions=directions)
        if x is not None:
            # (the library packs column k of the m hits at x[k * m + i]: the first nx * m doubles of the block)
            W = nx // 3
            xs = x.reshape(-1)[:nx * m].reshape(nx, m)
            out['spectra'] = (xs[W:2 * W], xs[2 * W:], xs[:W])         # (in, out, wavelengths): feed_accountants' order
        return out

    def bin_hits(self, surf_lo, surf_hi, ranges, mode):
        """
        Absorbed energy of the captured hits per view-factor element (trc_scene_bin_hits; the device form of
        emissive_losses/view_factors_3D.py `alloc_VF`).  surf_lo/surf_hi: inclusive surface index range per element;
        ranges (n, 6): ang0, ang1, h0, h1, r0, r1; mode: _cabi.BIN_* bits per element.
        """
        lo = N.ascontiguousarray(surf_lo, dtype=N.int32)
        hi = N.ascontiguousarray(surf_hi, dtype=N.int32)
        rng = N.ascontiguousarray(ranges, dtype=float).reshape(-1, 6)
        md = N.ascontiguousarray(mode, dtype=N.int32)
        n = len(lo)
        if not (len(hi) == n and len(rng) == n and len(md) == n):
            raise ValueError('bin_hits: the four element arrays must have the same length')
        out = N.zeros(n)
        i32 = C.POINTER(C.c_int32)
        _cabi.check(self.lib.trc_scene_bin_hits(self.handle, n, lo.ctypes.data_as(i32), hi.ctypes.data_as(i32), _cabi.ptr(rng),
                                                md.ctypes.data_as(i32), _cabi.ptr(out)))
        return out

    def enable_transfer(self, on=True):
        """keep (or drop) the surface-to-surface transfer matrix of the fast engine; resets the tallies"""
        self.settle_pending()
        _cabi.check(self.lib.trc_scene_enable_transfer(self.handle, 1 if on else 0))

    def get_transfer(self):
        """(n_surf + 1, n_surf): energy carried from surface `row` (last row: the source) to surface `column`"""
        out = N.zeros((self.n_surf + 1, self.n_surf))
        _cabi.check(self.lib.trc_scene_get_transfer(self.handle, _cabi.ptr(out)))
        return out

    def tally_size(self):
        n = C.c_int64(0)
        _cabi.check(self.lib.trc_scene_tally_size(self.handle, C.byref(n)))
        return n.value

    def export_tallies(self, out=None):
        """Packed tally buffer as a host array, or into a device pointer (int) when `out` is given."""
        if out is None:
            buf = N.empty(self.tally_size())
            _cabi.check(self.lib.trc_scene_export_tallies(self.handle, buf.ctypes.data_as(C.c_void_p), 0))
            return buf
        _cabi.check(self.lib.trc_scene_export_tallies(self.handle, C.c_void_p(int(out)), 1))
        return None

    def import_tallies(self, src):
        if isinstance(src, N.ndarray):
            src = _cabi.f64(src)
            _cabi.check(self.lib.trc_scene_import_tallies(self.handle, src.ctypes.data_as(C.c_void_p), 0))
        else:
            _cabi.check(self.lib.trc_scene_import_tallies(self.handle, C.c_void_p(int(src)), 1))

    # -- tracing --------------------------------------------------------------------------------
    def _bundle_args(self, bundle):
        """(rays struct or None, source desc or None, n, seed override, offset, keepalive)"""
        from .sources import LazySourceBundle
        if isinstance(bundle, LazySourceBundle) and bundle.is_pending():
            desc, n, seed, off = bundle.source_args()
            return None, desc, n, seed, off, None
        cols = bundle.columns_soa()
        n = cols['x'].shape[0]
        mats = getattr(self.compiled, 'materials', [])
        if mats:
            # the materials' own m(lambda) at every ray's wavelength (children inherit the wavelength): tables, files and analytic
            # models alike, and the device's comparison `index == material_1's` (optics_callables.py:750) stays exact
            if 'wavelength' not in cols:
                raise ValueError("the scene refracts between materials: the bundle needs a `wavelengths` column")
            cols['mat'] = material_rows(mats, cols['wavelength'])
        rays = _cabi.make_rays(n, cols['x'], cols['y'], cols['z'], cols['dx'], cols['dy'], cols['dz'], cols['e'],
                               ref_index=cols.get('ref_index'), wavelength=cols.get('wavelength'),
                               ref_index_im=cols.get('ref_index_im'), spec_wl=cols.get('spec_wl'), spectra=cols.get('spectra'),
                               mat=cols.get('mat'))
        return rays, None, n, None, 0, cols

    def trace_fast(self, bundle, reps, min_energy, seed, accel=False, keep_last=False, stream=None, last_capacity=None):
        """last_capacity: room for the rays still alive after `reps` interactions (keep_last); None = one per ray.  More rays
        left than that is a TracerAmdError with status ERR_CAPACITY."""
        rays, src, n, src_seed, off, keep = self._bundle_args(bundle)
        if src_seed is not None:
            seed = src_seed
        flags = (_cabi.TRACE_ACCEL if accel else 0) | (_cabi.TRACE_KEEP_LAST if keep_last else 0) | \
            (0 if stream is None else (_cabi.TRACE_STREAM if stream else _cabi.TRACE_MEGAKERNEL))
        stats = _cabi.TraceStats()
        last = None
        last_cols = None
        if keep_last:
            m = n if last_capacity is None else max(1, min(int(last_capacity), n))
            last_cols = [N.empty(m) for _ in range(7)]
            last = _cabi.make_rays(m, *last_cols)
        _cabi.check(self.lib.trc_trace_fast(self.handle, C.byref(rays) if rays is not None else None,
                                            C.byref(src) if src is not None else None, n, int(reps), float(min_energy),
                                            int(seed), int(off), flags, C.byref(last) if last is not None else None,
                                            C.byref(stats)))
        if keep_last:
            m = last.n
            last_cols = [c[:m] for c in last_cols]
        return stats, last_cols

    def trace_ordered(self, bundle, reps, min_energy, seed, accel=False):
        rays, src, n, src_seed, off, keep = self._bundle_args(bundle)
        if src_seed is not None:
            seed = src_seed
        flags = _cabi.TRACE_ACCEL if accel else 0
        stats = _cabi.TraceStats()
        res = C.c_void_p()
        _cabi.check(self.lib.trc_trace_ordered(self.handle, C.byref(rays) if rays is not None else None,
                                               C.byref(src) if src is not None else None, n, int(reps),
                                               float(min_energy), int(seed), int(off), flags, C.byref(res),
                                               C.byref(stats)))
        return OrderedResult(self, res), stats


class OrderedResult(object):
    """trc_result handle: the RayTree levels of an ordered trace, fetched level by level."""
    def __init__(self, scene, handle):
        self.scene = scene
        self.lib = scene.lib
        self.handle = handle

    def close(self):
        if self.handle is not None and self.handle.value:
            self.lib.trc_result_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def num_levels(self):
        n = C.c_int32(0)
        _cabi.check(self.lib.trc_result_num_levels(self.handle, C.byref(n)))
        return n.value

    def level_size(self, level):
        a, b = C.c_int64(0), C.c_int64(0)
        _cabi.check(self.lib.trc_result_level_size(self.handle, level, C.byref(a), C.byref(b)))
        return a.value, b.value

    def level(self, level, with_ref_index=True, with_wavelength=False, complex_index=False, n_spec=0):
        """dict: vertices (3,n), directions (3,n), energy, parents, surf, ref_index[, wavelengths], n_live.
        complex_index: ref_index comes back complex; n_spec = W > 0: `wavelengths` and `spectra` are the (W, n) columns of a
        polychromatic bundle."""
        n, n_live = self.level_size(level)
        v = N.empty((3, n))
        d = N.empty((3, n))
        e = N.empty(n)
        par = N.empty(n, dtype=N.int64)
        ri = N.empty(n) if with_ref_index else None
        wl = N.empty(n) if with_wavelength else None
        surf = N.empty(n, dtype=N.int32)
        im = N.empty(n) if (complex_index and with_ref_index) else None
        swl = N.empty((n_spec, n)) if n_spec else None
        sp = N.empty((n_spec, n)) if n_spec else None
        rays = _cabi.make_rays(n, v[0], v[1], v[2], d[0], d[1], d[2], e, parent=par, ref_index=ri, wavelength=wl,
                               ref_index_im=im, spec_wl=swl, spectra=sp)
        _cabi.check(self.lib.trc_result_level_get(self.handle, level, C.byref(rays),
                                                  surf.ctypes.data_as(C.POINTER(C.c_int32))))
        volume = None
        if self.scene.compiled_has_scattering():
            volume = (surf & _cabi.LEVEL_VOLUME) != 0          # rays scattered in the medium in front of `surf` (never reached it)
            surf &= ~N.int32(_cabi.LEVEL_VOLUME)
            if not volume.any():
                volume = None
        out = dict(vertices=v, directions=d, energy=e, parents=par, surf=surf, n_live=n_live, volume=volume)
        if ri is not None:
            out['ref_index'] = ri if im is None else ri + 1j * im
        if wl is not None:
            out['wavelengths'] = wl
        if n_spec:
            out['wavelengths'] = swl
            out['spectra'] = sp
        return out


class PendingHits(Delivery):
    """
    The hits a DeviceScene's buffer holds for the accountants of its capturing surfaces (fast engine): fetched and fed when
    the first of them is read, or when the buffer has to be emptied.  Traces that follow each other without a read in
    between go on filling the same buffer and are delivered together.
    """
    def __init__(self, dev):
        Delivery.__init__(self)
        self.dev = dev
        self.surfaces = dev.compiled.surfaces

    def deliver(self, holders):
        dev = self.dev
        if dev.handle is None:
            return
        h = dev.get_hits()
        cap = [i for i, c in enumerate(dev.compiled.capture) if c]
        one = cap[0] if len(cap) == 1 else None
        sp = h.get('spectra')               # polychromatic hits: (spectra in, spectra out, sample wavelengths), (W, hits) each
        if h['directions'] is None:         # Receiver accountants only: absorbed energy and hit points
            feed_accountants(self.surfaces, h['surf'], h['e_in'], None, h['points'], None, e_abs=h['e_abs'], only=holders, one_surface=one, spectra=sp)
        else:
            feed_accountants(self.surfaces, h['surf'], h['e_in'], h['e_in'] - h['e_abs'], h['points'], h['directions'], only=holders,
                             one_surface=one, spectra=sp)

    def release(self):
        if self.dev.pending_hits is self:
            self.dev.pending_hits = None
        self.dev = None
        self.surfaces = None


def feed_accountants(surfaces, surf_ids, e_in, e_out, points, directions, wavelengths=None, spectra=None, e_abs=None, only=None,
                     one_surface=None):
    """
    Hand per-hit data to the accountants of each surface's optics (fused engines).  Inside one call
    the hits of a surface are kept in the order given.  only: ids of the accountants to feed (a delivery settled late,
    deferred.py: the others were reset since the trace), None = all.
    """
    surf_ids = N.asarray(surf_ids)
    if len(surf_ids) == 0:
        return
    if one_surface is not None:
        # the caller knows that every hit is on this surface (the only one that captures): 6.5e6 comparisons saved
        order, uniq, start, stop = None, [int(one_surface)], [0], [len(surf_ids)]
    elif surf_ids[0] == surf_ids[-1] and (surf_ids == surf_ids[0]).all():
        # one capturing surface (the receiver of a field): no sorting, no gathering of 1e7 hits
        order, uniq, start, stop = None, [int(surf_ids[0])], [0], [len(surf_ids)]
    elif (surf_ids[1:] >= surf_ids[:-1]).all():
        # surface by surface already (the ordered engine's levels): runs are slices, nothing is gathered
        order = None
        start = N.r_[0, N.nonzero(surf_ids[1:] != surf_ids[:-1])[0] + 1]
        uniq = surf_ids[start]
        stop = list(start[1:]) + [len(surf_ids)]
    else:
        # device arrival order (fast engine)
        counts = N.bincount(surf_ids)
        present = N.nonzero(counts)[0]
        if len(present) <= 32:
            # a few capturing surfaces: one linear pass each finds a surface's hits, in order (a sort of 1e6 keys costs ten of them)
            order, uniq, start, stop = present, present, [None] * len(present), [None] * len(present)
        else:
            order = N.argsort(surf_ids, kind='stable')
            sorted_ids = surf_ids[order]
            start = N.r_[0, N.nonzero(sorted_ids[1:] != sorted_ids[:-1])[0] + 1]
            uniq = sorted_ids[start]
            stop = list(start[1:]) + [len(order)]
    for s, a, b in zip(uniq, start, stop):
        opt = surfaces[s].get_optics_manager()
        if not isinstance(opt, OpticsCallable) or not opt.accountants:
            continue
        idx = slice(int(a), int(b)) if order is None else (N.nonzero(surf_ids == s)[0] if a is None else order[a:b])
        surf = surfaces[s]
        pts = points[:, idx]
        dirs = None if directions is None else directions[:, idx]

        def normals(surf=surf, pts=pts, dirs=dirs):
            gm = surf.get_geometry_manager()
            desc, _ = gm._desc(surf._temp_frame)
            ctx = _cabi.get_context()
            h = _cabi.f64(pts)
            d = _cabi.f64(dirs)
            out = N.empty_like(h)
            _cabi.check(ctx.lib.trc_gm_get_normals(ctx.handle, C.byref(desc), h.shape[1], _cabi.ptr(h[0]), _cabi.ptr(h[1]),
                                                   _cabi.ptr(h[2]), _cabi.ptr(d[0]), _cabi.ptr(d[1]), _cabi.ptr(d[2]),
                                                   _cabi.ptr(out[0]), _cabi.ptr(out[1]), _cabi.ptr(out[2])))
            return out
        hit = dict(e_in=e_in[idx], e_out=None if e_out is None else e_out[idx], points=pts, directions=dirs, normals=normals,
                   wavelengths=None if wavelengths is None else wavelengths[idx])
        if e_abs is not None:       # (given instead of e_out by the fast engine's lean capture: no array of differences is made)
            hit['e_abs'] = e_abs[idx]
        if spectra is not None:     # polychromatic bundles: (spectra of the incident rays, of the outgoing ones, their wavelength grids)
            hit['spectra_in'], hit['spectra_out'], hit['wavelengths'] = spectra[0][:, idx], spectra[1][:, idx], spectra[2][:, idx]
        for acc in opt.accountants:
            if only is None or id(acc) in only:
                acc.feed(hit)
