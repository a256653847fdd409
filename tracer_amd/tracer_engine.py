"""
TracerEngine: the entry point of the hot path.

Signature and side effects follow the reference's tracer/tracer_engine.py:16-25, :124-295:
`ray_tracer(bundle, reps, min_energy, tree, accel, Kd_Tree, **kwargs)` returns the vertices and
directions of the rays still alive, fills `self.tree` (RayTree) and the accountants of the
surfaces' optics.  What differs is where the work happens:

  engine='ordered'  (default when tree=True)  one device launch set per bounce; bundle ordering,
                    parents and culled-rays-last layout identical to the reference (trc_trace_ordered).
  engine='fast'     (default when tree=False) persistent-wavefront kernel, rays stay in registers for
                    all bounces, tallies / flux maps / hit buffer on the device (trc_trace_fast);
                    `self.tree` stays empty.
  engine='protocol' host loop over surfaces calling register_incoming / select_rays / get_outgoing
                    (the reference's four-step protocol) for scenes with user-defined Python geometry
                    or optics; native kinds inside such a scene still run their device kernels.

There is no CPU implementation of the native kinds: without the HIP library or a GPU every path
raises.
"""
import logging
import time
import weakref

import numpy as N

from . import rng
from .accel_tree import KdTree
from .ray_bundle import RayBundle, concatenate_rays
from .scene import compile_scene, DeviceScene, NotNativeError, feed_accountants, PendingHits
from .optics_callables import OpticsCallable
from .trace_tree import RayTree
from .ordered_levels import OrderedLevels, LazyLevelBundle, PendingLevels, level_columns


class TracerEngine(object):
    def __init__(self, parent_assembly, loglevel=logging.DEBUG):
        self._asm = parent_assembly
        self.loglevel = loglevel
        self._dev = None
        self._dev_sig = None
        self._dev_static_sig = None
        self._dev_frames = None
        self._fluxmap_requests = {}
        self._transfer = False
        self._transfer_host = None      # contribution of ordered-engine runs (from the tree's parents)
        self._kd_on_device = None
        self._auto_kd = None
        self._ordered_pending = []      # weak references to what ordered traces still owe to accountants (ordered_levels.PendingLevels)
        self._transfer_pending = []     # those of them that carry a contribution to the transfer matrix (kept alive until it is read)
        self.stats = {}

    # -- the Kd-tree of the last accelerated call (tracer_engine.py:171-185) -------------------------------
    @property
    def Kd_Tree(self):
        lazy = getattr(self, '_kd_lazy', None)
        if lazy is not None:
            key, max_depth, fast, kw = lazy
            if self._auto_kd is not None and self._auto_kd[0] == key:
                self._Kd_Tree = self._auto_kd[1]
            else:
                # built from the assembly as it stands NOW: kept for later traces only when that is still the scene the call
                # traced (a script may have re-posed the assembly between the trace and this read -- a tree of the new poses
                # under the old poses' key would be handed to the device the next time the old poses are traced)
                tree = KdTree(self._asm, max_depth, loglevel=self.loglevel, fast=fast, **kw)
                if compile_scene(self._asm).signature() == key[0]:
                    self._auto_kd = (key, tree)
                self._Kd_Tree = tree
            self._kd_lazy = None
        return getattr(self, '_Kd_Tree', None)

    @Kd_Tree.setter
    def Kd_Tree(self, value):
        self._Kd_Tree = value
        self._kd_lazy = None

    # -- device scene management ------------------------------------------------------------------
    def _device_scene(self, unchanged=False):
        if unchanged and self._dev is not None:      # the caller vouches for it (ray_tracer(scene_unchanged=True)): no re-compilation
            return self._dev
        compiled = compile_scene(self._asm)
        sig = compiled.signature()
        if self._dev is not None and sig == self._dev_sig:
            self._dev.compiled = compiled   # same numbers, fresh Surface objects
            return self._dev
        self._settle_engine_pendings()      # (what earlier traces owe to accountants is delivered in the poses it was made in)
        if self._dev is not None and compiled.signature_without_frames() == self._dev_static_sig and \
                self._fluxmaps_unmoved(compiled):
            # the scene only moved (a heliostat field following the sun): new frames for the scene already on the device
            self._dev.update_frames(compiled)
        else:
            if self._dev is not None:
                self._dev.close()
            self._dev = DeviceScene(compiled)
            self._dev_static_sig = compiled.signature_without_frames()
            for si, (u, v) in self._fluxmap_requests.items():
                self._dev.set_fluxmap(si, u, v)
            if self._transfer:
                self._dev.enable_transfer(True)
        self._dev_sig = sig
        self._dev_frames = compiled.frames12()      # a snapshot: the Surface objects themselves move with the scene
        self._kd_on_device = None
        self._auto_kd = None
        return self._dev

    def _fluxmaps_unmoved(self, compiled):
        """flux maps are binned in the frame their surface had when they were set: such a surface must not have moved"""
        new = compiled.frames12()
        return all(N.array_equal(self._dev_frames[si], new[si]) for si in self._fluxmap_requests)

    def set_fluxmap(self, surface, u_edges, v_edges):
        """
        Ask the device to bin the energy absorbed by `surface` (a Surface of the assembly, or its
        index) on u_edges x v_edges of its local x, y -- the on-device form of the caller-side
        numpy.histogram2d of the reference's examples.  Read it back with get_fluxmap().
        """
        si = surface if isinstance(surface, int) else self._asm.get_surfaces().index(surface)
        self._fluxmap_requests[si] = (N.asarray(u_edges, dtype=float), N.asarray(v_edges, dtype=float))
        if self._dev is not None:
            self._dev.close()
            self._dev = None

    def get_fluxmap(self, surface):
        si = surface if isinstance(surface, int) else self._asm.get_surfaces().index(surface)
        return self._dev.get_fluxmap(si)

    def get_tallies(self):
        """(absorbed, received, hits) per surface, accumulated on the device since the last reset."""
        return self._dev.get_tallies()

    def bin_hits(self, surf_lo, surf_hi, ranges, mode):
        """absorbed energy of the hits captured by the last fast trace per (surface range, azimuth, height, radius) element,
        binned on the device (DeviceScene.bin_hits; the view-factor allocation of emissive_losses)"""
        return self._dev.bin_hits(surf_lo, surf_hi, ranges, mode)

    def enable_transfer_matrix(self, on=True):
        """
        Keep the surface-to-surface energy transfer of the following traces: get_transfer_matrix()[i, j] is the energy
        carried by the ray segments that leave surface i (last row: the source bundle) and land on surface j.  This is
        what the blocking / shading post-process of the reference's NSTTF example (examples/Sandia_NSTTF_field
        example.py:229-290) recovers by matching hit coordinates on the host; here the fast engine accumulates it while
        shading (models.heliostat_field.field_losses reads the example's quantities off it).  Resets the tallies.
        """
        self._transfer = bool(on)
        self._transfer_host = None
        if self._dev is not None:
            self._dev.enable_transfer(self._transfer)

    def get_transfer_matrix(self):
        if not self._transfer:
            raise ValueError('call enable_transfer_matrix() before tracing')
        for p in self._transfer_pending:        # contributions of ordered traces whose levels are still on the device
            p.settle()
        self._transfer_pending = []
        n = len(self._asm.get_surfaces())
        T = self._dev.get_transfer() if self._dev is not None else N.zeros((n + 1, n))
        return T if self._transfer_host is None else T + self._transfer_host

    def reset_tallies(self):
        for p in self._transfer_pending:
            p.transfer = False
            p.always = False
        self._transfer_pending = []
        self._transfer_host = None
        if self._dev is not None:
            self._dev.reset_tallies()

    # -- the entry point ------------------------------------------------------------------------------
    KD_KEYWORDS = ('min_leaf', 't_trav', 't_isec', 'empty_bonus', 'debug', 'split_threshold')    # KdTree's own (accel_tree.py:42-60)
    KD_BUILD_MAX = 8192      # surfaces beyond which accel=True does not build the reference's Kd-tree for the fast engine
    KD_WORTH_IT = 2e9        # rays x surfaces from which the ordered engine gets the reference's Kd-tree built for accel=True
    TUNE_MIN_RAYS = 1 << 21  # calls from which the two forms of the fast engine are compared on a scene (fast_kernel='auto')
    TUNE_MAX_SURFACES = 512  # ... and the scene size up to which the megakernel is worth a try
    SLOW_STREAM = 1.5e6      # segments per ms of kernel time below which the streaming form counts as slow
    HITS_RESIDENT_MAX = 1 << 30     # entries up to which unread hits of successive calls are kept in the device's buffer (68 B each)
    LEVELS_RESIDENT_MAX = 16 << 30  # bytes of unread ray-tree levels of earlier calls that may stay on the device

    def ray_tracer(self, bundle, reps=100, min_energy=1e-10, tree=True, accel=False, Kd_Tree=None, **kwargs):
        """
        Trace `bundle` through the assembly for at most `reps` interactions per ray, dropping rays
        whose energy falls to `min_energy` or below.  accel: False, True, 'fast' or 'lightweight'
        (Kd-tree over the objects' BoundaryBoxes; 'lightweight' is accepted and gives the same results).
        Extra keywords: engine ('auto'|'ordered'|'fast'|'protocol'), seed, hit_capacity, last_capacity, scene_unchanged, and the
        KdTree keywords of the reference (min_leaf, t_trav, t_isec, empty_bonus).
        tree=False records the last bundle only, like the reference (tracer_engine.py:288-291): its rays are those the call
        returns, with their energies; the fast engine does not keep the bundle before it, so this record has no parents.
        """
        engine = kwargs.pop('engine', 'auto')
        seed = kwargs.pop('seed', None)
        hit_capacity = kwargs.pop('hit_capacity', None)
        fast_kernel = kwargs.pop('fast_kernel', 'auto')     # 'auto' | 'stream' | 'megakernel' (fast engine only)
        feed = kwargs.pop('feed', True)     # False: captured hits stay on the device (bin_hits), accountants are not fed
        last_capacity = kwargs.pop('last_capacity', None)   # fast engine: room for the rays still alive after `reps` (see _trace_fast)
        # Every call compiles the assembly into its table again to see whether anything changed since the last one -- poses, optics
        # parameters, surfaces -- which is 1.8 ms for the 219 surfaces of the NSTTF field, against 0.1 ms of tracing for 1e5 rays.
        # A Monte-Carlo loop that knows the scene stands still says so:
        scene_unchanged = kwargs.pop('scene_unchanged', False)
        if seed is None:
            seed = rng.next_seed()
        self.reps = reps
        self.minener = min_energy
        t_call = time.time()
        self.tree = RayTree()           # (the tree of the call before goes: levels nobody can read any more leave the device here)
        self._t_marks = [('old tree released', time.time() - t_call)]

        if engine == 'protocol':
            return self._trace_protocol(bundle, reps, min_energy, tree)
        try:
            t1 = time.time()
            dev = self._device_scene(unchanged=scene_unchanged)
            self._t_marks.append(('scene checked', time.time() - t1))
        except NotNativeError as err:
            if engine != 'auto':
                raise
            logging.log(self.loglevel, 'protocol engine: %s' % err)
            return self._trace_protocol(bundle, reps, min_energy, tree)

        if engine == 'auto':
            # Complex refractive indices, materials evaluated at the rays' wavelengths and spectra travel with the rays of the
            # ordered engine and of the streaming form of the fast engine (64 rays or more of a given bundle; k_s_shade_x).  A
            # captured hit of a polychromatic ray keeps its sample wavelengths and its spectrum before and after the surface (what
            # PolychromaticAccountant collects).
            carries = dev.compiled.carries or (not _pending(bundle) and (bundle.is_polychromatic() or bundle.has_complex_index()))
            if carries:
                carries = _pending(bundle) or bundle.get_num_rays() < 64
            engine = 'ordered' if (tree or dev.compiled.splits or carries) else 'fast'
        if accel and Kd_Tree is None and (engine == 'fast' or (engine == 'ordered' and (dev.n_surf > self.KD_BUILD_MAX or (
                dev.n_surf <= 65535 and bundle.get_num_rays() * dev.n_surf <= self.KD_WORTH_IT)))):
            # The fast engine does not walk the reference's Kd-tree: large calls search the library's own uniform grid over the same
            # geometry boxes (csrc/trc_bounds.h), small ones test the boxes themselves.  Building the tree -- the reference's SAH
            # build, Python: 38 ms for the 219 surfaces of the NSTTF field, minutes for a mesh of 1e5 faces -- before every trace
            # of a scene that moves (a day of sun positions) cost more than the traces.  engine.Kd_Tree builds it when it is read.
            # The ordered engine walks it, but below KD_WORTH_IT box tests per bounce (rays x surfaces: a millisecond of the GPU)
            # testing every surface's box costs less than the build: 1e5 rays on the NSTTF field took 30 ms with the tree, 3 without.
            # Beyond KD_BUILD_MAX surfaces (a mesh) the tree is never built for a trace: the scene stands on the library's large grid,
            # which the ordered engine walks too (trc_nearest_grid32).
            num_surfs = dev.n_surf
            kw = dict(kwargs)
            kw.setdefault('min_leaf', 1)
            unknown = [k for k in kw if k not in self.KD_KEYWORDS]       # (the tree is built later, if at all: complain now)
            if unknown:
                raise TypeError("ray_tracer() got unexpected keyword argument(s) %s" % ', '.join(repr(k) for k in unknown))
            self._Kd_Tree = None
            self._kd_lazy = None if num_surfs > self.KD_BUILD_MAX else \
                ((self._dev_sig, accel == 'fast', tuple(sorted(kw.items()))), 8 + 1.3 * N.log(num_surfs), accel == 'fast', kw)
            if self._kd_on_device is not None:
                dev.set_kdtree(None)
                self._kd_on_device = None
        elif accel:
            if Kd_Tree is None:
                num_surfs = dev.n_surf
                max_depth = 8 + 1.3 * N.log(num_surfs)
                logging.log(self.loglevel, 'Maximum Kd tree depth %i' % max_depth)
                kw = dict(kwargs)
                kw.setdefault('min_leaf', 1)
                # the tree only depends on the scene and on these arguments: repeated calls (Monte-Carlo loops) reuse it
                key = (self._dev_sig, accel == 'fast', tuple(sorted(kw.items())))
                if self._auto_kd is None or self._auto_kd[0] != key:
                    self._auto_kd = (key, KdTree(self._asm, max_depth, loglevel=self.loglevel, fast=(accel == 'fast'), **kw))
                self.Kd_Tree = self._auto_kd[1]
            else:
                self.Kd_Tree = Kd_Tree
            if self._kd_on_device is not self.Kd_Tree:
                dev.set_kdtree(self.Kd_Tree)
                self._kd_on_device = self.Kd_Tree

        if engine == 'fast':
            return self._trace_fast(dev, bundle, reps, min_energy, seed, bool(accel), hit_capacity, fast_kernel, feed, last_capacity)
        if engine == 'ordered':
            # (no tree on the device: the call was too small for one to pay, every surface's box is tested)
            return self._trace_ordered(dev, bundle, reps, min_energy, seed, bool(accel) and self._kd_on_device is not None, tree)
        raise ValueError("unknown engine %r" % (engine,))

    # -- fast engine --------------------------------------------------------------------------------
    def _trace_fast(self, dev, bundle, reps, min_energy, seed, accel, hit_capacity, fast_kernel='auto', feed=True, last_capacity=None):
        n = bundle.get_num_rays()
        capture = any(dev.compiled.capture)
        accs = []
        pend = None
        if capture:
            # The hits stay in the device's buffer until an accountant is read (scene.PendingHits).  A call that finds the hits
            # of the calls before still unread there -- every accountant concerned still holds its mark -- goes on filling the
            # same buffer, grown if need be; otherwise what it holds is delivered (or dropped, when nobody waits for it any more)
            # and the buffer emptied.  An explicit hit_capacity is taken literally: an empty buffer of that size.
            # Room for the hits of this call: two per ray unless the caller says otherwise -- or, once the scene in its present
            # poses has been traced, four times the share of captured hits per ray seen so far (a field sends 6 % of its rays to
            # the receiver: successive calls then fit the buffer of the first many times over before it has to grow).
            need = int(hit_capacity) if hit_capacity is not None else 2 * n + 1024
            if hit_capacity is None and dev.capture_rate is not None:
                need = min(need, int(4. * dev.capture_rate * n) + 65536)
            accs = [a for opt in dev.compiled.capturing_optics for a in opt.accountants]
            pend = dev.pending_hits
            keep = bool(feed and hit_capacity is None and pend is not None and pend.wanted() and accs and all(pend.holds_mark(a) for a in accs))
            if keep and not _pending(bundle) and bundle.is_polychromatic():
                keep = False        # (the spectra of captured hits live beside the buffer as it is: hits waiting there are delivered first)
            used = 0
            if keep:
                used = dev.hits_reserved()[0]
                keep = used + need <= self.HITS_RESIDENT_MAX
            if keep:
                dev.reserve_hits(used + need)
            else:
                pend = None
                used = 0
                if hit_capacity is None and dev.hit_capacity >= need:
                    dev.settle_pending()        # (a buffer that is large enough already is kept: freeing and allocating 15 GB costs 0.1 s)
                else:
                    dev.set_hit_capacity(need)
                dev.lib.trc_scene_clear_hits(dev.handle)
        t0 = time.time()
        stream = {'auto': None, 'stream': True, 'megakernel': False}[fast_kernel]
        if stream is None and accel and dev.n_surf > self.KD_BUILD_MAX:
            stream = True       # the streaming form has the grid for large scenes; the megakernel would test every box
        # Large calls go to the streaming form.  On a scene where it turns out slow -- every segment a hit on overlapping curved
        # shapes, several bounces deep: below SLOW_STREAM segments per ms, a fortieth of its rate on a heliostat field -- the next
        # large call tries the megakernel, whose rays stay in registers, and the faster of the two serves the scene from then on.
        # Both forms end every ray alike (test_forms_of_the_fast_engine_end_every_ray_alike).
        tuned = stream is None and n >= self.TUNE_MIN_RAYS and dev.n_surf <= self.TUNE_MAX_SURFACES
        if tuned:
            rate = dev.form_rate
            if 'stream' in rate and rate['stream'] < self.SLOW_STREAM:
                stream = False if ('megakernel' not in rate or rate['megakernel'] > rate['stream']) else True
        # Rays still alive after `reps` interactions come back as the call's result (tracer_engine.py:293-295).  Bundles beyond
        # 2^24 rays get room for 2^22 of them unless the caller says otherwise (last_capacity=...): 1e8 rays would cost 5.6 GB
        # of host arrays per call for a result that is empty in most scenes.  More rays left than room is an error of the
        # call (status ERR_CAPACITY), raised after the trace: tallies and flux maps of the scene then hold it, the accountants do not.
        cap = last_capacity if last_capacity is not None else (n if n <= (1 << 24) else (1 << 22))
        stats, last = dev.trace_fast(bundle, reps, min_energy, seed, accel=accel, keep_last=True, stream=stream, last_capacity=cap)
        wall = time.time() - t0
        self._set_stats(stats, wall, 'fast')
        self.stats['form'] = 'stream' if stats.launches > 1 else 'megakernel'
        if tuned and stats.kernel_ms > 0:
            dev.form_rate[self.stats['form']] = stats.segments / stats.kernel_ms
        if stats.hits_dropped:
            raise RuntimeError("%d hits were not captured: the hit buffer holds %d; pass hit_capacity=..."
                               % (stats.hits_dropped, dev.hit_capacity))
        if capture and n > 0:
            rate = max(dev.hits_reserved()[0] - used, 0) / float(n)
            dev.capture_rate = rate if dev.capture_rate is None else max(dev.capture_rate, rate)
        if capture and feed and accs:
            if pend is None:
                pend = dev.pending_hits = PendingHits(dev)
            pend.surfaces = dev.compiled.surfaces
            for a in accs:
                pend.give_mark(a)
        self._warn_left(stats.rays_left, stats.energy_left, bundle)
        vertices, directions = N.vstack(last[0:3]), N.vstack(last[3:6])
        # "otherwise only register the last bundle" (tracer_engine.py:288-291): scripts read engine.tree[-1] after tree=False
        self.tree.append(RayBundle(vertices=vertices, directions=directions, energy=N.asarray(last[6])))
        return vertices, directions

    # -- ordered engine -----------------------------------------------------------------------------
    def _trace_ordered(self, dev, bundle, reps, min_energy, seed, accel, tree):
        has_ref = bundle._has_column('ref_index') if not _pending(bundle) else False
        has_wl = bundle._has_column('wavelengths') if not _pending(bundle) else False
        n_spec = bundle.get_spectra().shape[0] if (not _pending(bundle) and bundle.is_polychromatic()) else 0
        cplx = bool(dev.compiled.materials) or (not _pending(bundle) and bundle.has_complex_index())
        if n_spec:
            has_wl = False          # `wavelengths` is the (W, N) grid of the spectra
        self._trim_resident_levels()
        t0 = time.time()
        res, stats = dev.trace_ordered(bundle, reps, min_energy, seed, accel=accel)
        wall = time.time() - t0
        self._set_stats(stats, wall, 'ordered')
        # Nothing is copied here: the levels stay on the device (ordered_levels.py).  engine.tree holds bundles whose columns
        # arrive when they are read, the accountants hold marks that the first get_data() / get_all_hits() settles.
        levels = OrderedLevels(res, has_wl, cplx, n_spec)
        nlev = levels.nlev
        names = level_columns(has_ref or dev.compiled.splits or _has_refractive(dev), has_wl, n_spec)
        if tree is True:
            self.tree.append(bundle)
        for lv in range(1, nlev):
            if tree is True or lv == nlev - 1:
                self.tree.append(LazyLevelBundle(levels, lv, names))
        surfaces = dev.compiled.surfaces
        accs = [a for opt in dev.compiled.optics if isinstance(opt, OpticsCallable) for a in opt.accountants] if surfaces is not None else []
        if nlev > 1 and (accs or self._transfer):
            pend = PendingLevels(self, surfaces, levels, bundle, dev.n_surf, self._transfer)
            for a in accs:
                pend.give_mark(a)
            self._ordered_pending.append(weakref.ref(pend))
            if self._transfer:
                self._transfer_pending.append(pend)
        if nlev <= 1 or stats.rays_left == 0:
            logging.log(self.loglevel, 'Ray bundle depleted')
            return N.zeros((3, 0)), N.zeros((3, 0))
        last = levels.level(nlev - 1)
        k = last['n_live']
        self._warn_left(stats.rays_left, stats.energy_left, bundle)
        return last['vertices'][:, :k], last['directions'][:, :k]

    def _trim_resident_levels(self):
        """unread levels of earlier ordered traces stay on the device up to LEVELS_RESIDENT_MAX bytes: beyond, the oldest are
        delivered to their accountants now"""
        live = [r() for r in self._ordered_pending]
        live = [p for p in live if p is not None and not p.settled and p.levels is not None]
        self._ordered_pending = [weakref.ref(p) for p in live]
        total = sum(p.levels.nbytes() for p in live)
        for p in live:
            if total <= self.LEVELS_RESIDENT_MAX:
                break
            total -= p.levels.nbytes()
            p.settle()

    def _settle_engine_pendings(self):
        for r in self._ordered_pending:
            p = r()
            if p is not None:
                p.settle()
        self._ordered_pending = []
        self._transfer_pending = []

    # -- protocol engine ----------------------------------------------------------------------------
    def intersect_ray(self, bundle, surfaces, surf_relevancy):
        """
        First surface hit by each ray, surfaces driven one at a time through register_incoming
        (tracer_engine.py:27-64): t == 0 is no hit, ties go to the lowest surface index.
        Returns (earliest_surf (-1 = none), surf_relevancy).
        """
        n = bundle.get_num_rays()
        best = N.full(n, N.inf)
        earliest = N.full(n, -1, dtype=int)
        for si, surf in enumerate(surfaces):
            rel = N.asarray(surf_relevancy[si], dtype=bool)
            if not rel.any():
                continue
            sub = bundle if rel.all() else bundle.inherit(rel)
            t = N.array(surf.register_incoming(sub), dtype=float)
            t[t == 0.] = N.inf
            idx = N.nonzero(rel)[0]
            closer = t < best[idx]
            best[idx[closer]] = t[closer]
            earliest[idx[closer]] = si
        return earliest, surf_relevancy

    def _trace_protocol(self, bundle, reps, min_energy, tree):
        surfaces = self._asm.get_surfaces()
        objects = self._asm.get_objects()
        S = len(surfaces)
        owner = N.repeat(N.arange(len(objects)), [len(o.get_surfaces()) for o in objects])
        first_of_obj = N.concatenate(([0], N.cumsum([len(o.get_surfaces()) for o in objects])))
        bund = bundle
        if tree is True:
            self.tree.append(bund)
        relevancy = N.ones((S, bund.get_num_rays()), dtype=bool)
        record = []
        for it in range(reps):
            front, _ = self.intersect_ray(bund, surfaces, relevancy)
            outg, record, weak, next_rel = [], [], [], []
            for si in range(S):
                rel_idx = N.nonzero(relevancy[si])[0]
                hit_here = front[rel_idx] == si
                if not hit_here.any():
                    surfaces[si].done()
                    continue
                surfaces[si].select_rays(N.nonzero(hit_here)[0])
                new = surfaces[si].get_outgoing()
                new.set_parents(rel_idx[new.get_parents()])        # index into the full bundle
                record.append(new)
                low = new.get_energy() <= min_energy
                weak.append(low)
                if low.any():
                    new = new.delete_rays(N.nonzero(low)[0])
                surfaces[si].done()
                outg.append(new)
                oi = owner[si]
                rel = N.ones((S, new.get_num_rays()), dtype=bool)
                rel[owner == oi] = objects[oi].surfaces_for_next_iteration(new, si - first_of_obj[oi])
                next_rel.append(rel)
            bund = concatenate_rays(outg)
            if tree:
                rec = concatenate_rays(record)
                if rec.get_num_rays() != 0:
                    self.tree.append(bund + rec.inherit(N.nonzero(N.hstack(weak))[0]))
            if bund.get_num_rays() == 0:
                logging.log(self.loglevel, 'Ray bundle depleted')
                break
            relevancy = N.hstack(next_rel)
        if not tree:
            self.tree.append(concatenate_rays(record))
        if bund.get_num_rays() != 0:
            self._warn_left(bund.get_num_rays(), N.sum(bund.get_energy()), bundle)
        return bund.get_vertices(), bund.get_directions()

    # -- helpers ------------------------------------------------------------------------------------
    def _set_stats(self, stats, wall, engine):
        self.stats = dict(engine=engine, segments=stats.segments, hits=stats.hits, rays_left=stats.rays_left,
                          energy_left=stats.energy_left, kernel_ms=stats.kernel_ms, bounces=stats.bounces,
                          launches=stats.launches, wall_s=wall, host_s=dict(getattr(self, '_t_marks', [])))
        logging.log(self.loglevel, 'trace time %s s' % wall)

    def _warn_left(self, rays_left, energy_left, bundle):
        if rays_left:
            logging.warning('%d rays left at the end of the simulation' % rays_left)
            try:
                tot = N.sum(bundle.get_energy()) if not _pending(bundle) else bundle.source_args()[0].energy * bundle.get_num_rays()
                logging.warning('Remaining energy in last bundle: %s%%' % (energy_left / tot * 100.))
            except Exception:
                pass


def _pending(bundle):
    return hasattr(bundle, 'is_pending') and bundle.is_pending()


def _has_refractive(dev):
    from . import _cabi
    return any(d.optics_kind == _cabi.OPT_REFRACTIVE_HOMOGENOUS for d in dev.compiled.descs)
