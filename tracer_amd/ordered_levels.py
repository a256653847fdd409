"""
The recorded levels of an ordered trace, left on the device until they are read.

`TracerEngine.ray_tracer(tree=True)` -- the reference's default -- records every bundle of the trace (tracer/tracer_engine.py
:268-274, tracer/trace_tree.py:6-55) and feeds the accountants of every surface hit (tracer/optics_callables.py:1949-1961).
The ordered engine produces all of that on the device; copying it to the host and pushing it through NumPy was 93 of the
100 ms of a 1e7-ray call.  Here the call returns once the kernels are done:

  OrderedLevels     the trc_result of the trace; a level is copied to the host (page-locked memory) the first time it is
                    asked for, once
  LazyLevelBundle   the RayBundle that engine.tree holds for a level: its columns appear on first access
  PendingLevels     what the trace owes to the accountants (deferred.Delivery): settled by the first get_data() /
                    get_all_hits() that meets one of its marks, with the same per-level feeding as before

The device memory of a trace is released when its levels have all been read or nobody can read them any more.
"""
import numpy as N

from .deferred import Delivery
from .optics_callables import OpticsCallable
from .ray_bundle import RayBundle
from .scene import feed_accountants


class OrderedLevels(object):
    def __init__(self, res, has_wl, cplx, n_spec):
        self.res = res
        self.has_wl, self.cplx, self.n_spec = has_wl, cplx, n_spec
        self.nlev = res.num_levels()
        self.sizes = [res.level_size(lv) for lv in range(self.nlev)]       # (rays recorded, rays that go on)
        self._cache = {}
        self._unread = set(range(1, self.nlev))

    def nbytes(self):
        """device bytes the levels hold (an estimate: 100 bytes per recorded ray and the spectra)"""
        return sum(n for n, _ in self.sizes) * (100 + 16 * self.n_spec)

    def level(self, lv):
        L = self._cache.get(lv)
        if L is None:
            if self.res is None:
                raise RuntimeError("the levels of this trace are gone from the device")
            L = self.res.level(lv, with_ref_index=True, with_wavelength=self.has_wl, complex_index=self.cplx, n_spec=self.n_spec)
            self._cache[lv] = L
        return L

    def forget(self, lv):
        """the level was handed to its (only) reader: the cache lets go of it; with the last level the device memory goes"""
        self._cache.pop(lv, None)
        self._unread.discard(lv)

    def close(self):
        if self.res is not None:
            self.res.close()
            self.res = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LazyLevelBundle(RayBundle):
    """level `lv` of an ordered trace as engine.tree holds it: generated columns arrive from the device on first access"""
    def __init__(self, levels, lv, names):
        RayBundle.__init__(self, **dict((k, None) for k in names))
        object.__setattr__(self, '_lv_src', (levels, lv, tuple(names)))
        object.__setattr__(self, '_lv_n', levels.sizes[lv][0])

    def get_num_rays(self):
        return self._lv_n

    def _materialize(self):
        src = self._lv_src
        if src is None:
            return
        object.__setattr__(self, '_lv_src', None)
        levels, lv, names = src
        L = levels.level(lv)
        for k in names:
            self._cols[k] = L[k]


def level_columns(has_ref, has_wl, n_spec):
    names = ['vertices', 'directions', 'energy', 'parents']
    if has_ref:
        names.append('ref_index')
    if has_wl or n_spec:
        names.append('wavelengths')
    if n_spec:
        names.append('spectra')
    return names


class PendingLevels(Delivery):
    """the accountant data (and, when the engine keeps one, the transfer-matrix contribution) of one ordered trace"""
    def __init__(self, engine, surfaces, levels, bundle, n_surf, transfer):
        Delivery.__init__(self)
        self.engine = engine
        self.surfaces = surfaces
        self.levels = levels
        self.bundle = bundle
        self.n_surf = n_surf
        self.transfer = transfer
        self.always = bool(transfer)

    def deliver(self, holders):
        lv_src, bundle, surfaces = self.levels, self.bundle, self.surfaces
        has_wl, n_spec = lv_src.has_wl, lv_src.n_spec
        acc_table = N.array([isinstance(sf.get_optics_manager(), OpticsCallable) and
                             any(id(a) in holders for a in sf.get_optics_manager().accountants) for sf in surfaces]) \
            if surfaces is not None else N.zeros(self.n_surf, dtype=bool)
        prev = dict(energy=N.asarray(bundle.get_energy()), directions=N.asarray(bundle.get_directions()),
                    wavelengths=bundle.get_wavelengths() if has_wl else None,
                    spectra=N.asarray(bundle.get_spectra()) if n_spec else None)
        prev_surf = None
        for lv in range(1, lv_src.nlev):
            L = lv_src.level(lv)
            # accountants: hits of a surface in the order the reference selects them (ascending parent).  Only the hits on
            # surfaces that have accountants are touched, and they are only sorted when the device's order -- (culled, surface,
            # block) with ascending parents inside -- is not that order already (no culled rays, one block: the usual case)
            par = L['parents']
            vol = L.get('volume')           # rays scattered in the medium in front of the surface they are filed under: not hits of it
            wanted = acc_table[L['surf']] if vol is None else (acc_table[L['surf']] & ~vol)
            order = N.nonzero(wanted)[0] if acc_table.any() else N.zeros(0, dtype=int)
            if len(order):
                whole = len(order) == len(par)           # every ray of the level ended on a surface with accountants: no gathering
                so, po = (L['surf'], par) if whole else (L['surf'][order], par[order])
                key = so.astype(N.int64) * (1 << 40) + po
                if not (key[1:] >= key[:-1]).all():
                    order = order[N.argsort(key, kind='stable')]
                    whole = False
                    so, po = L['surf'][order], par[order]
                sel = slice(None) if whole else order
                # (copies, not views, of what the recorded bundle of the tree holds: a script may edit engine.tree in place)
                feed_accountants(surfaces, so, prev['energy'][po], L['energy'].copy() if whole else L['energy'][order],
                                 L['vertices'].copy() if whole else L['vertices'][:, order], prev['directions'][:, po],
                                 None if prev['wavelengths'] is None else prev['wavelengths'][po],
                                 spectra=None if not n_spec else (prev['spectra'][:, po], L['spectra'][:, sel], L['wavelengths'][:, sel]),
                                 only=holders)
            if self.transfer and self.engine is not None:
                ns = self.n_surf
                left = N.full(len(prev['energy']), ns) if prev_surf is None else prev_surf
                if self.engine._transfer_host is None:
                    self.engine._transfer_host = N.zeros((ns + 1, ns))
                if vol is None:
                    N.add.at(self.engine._transfer_host, (left[par], L['surf']), prev['energy'][par])
                else:               # (a volume event lands nowhere, and the ray goes on from where it had left)
                    keep = ~vol
                    N.add.at(self.engine._transfer_host, (left[par][keep], L['surf'][keep]), prev['energy'][par][keep])
            prev_surf = L['surf'] if vol is None or not self.transfer else N.where(vol, (N.full(len(prev['energy']), self.n_surf) if prev_surf is None else prev_surf)[par], L['surf'])
            prev = dict(energy=L['energy'], directions=L['directions'], wavelengths=L.get('wavelengths') if has_wl else None,
                        spectra=L.get('spectra'))

    def release(self):
        self.levels = None
        self.bundle = None
        self.surfaces = None
        self.engine = None
