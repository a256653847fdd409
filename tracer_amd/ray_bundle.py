"""
RayBundle: the host-side view of a bundle of rays, one column per ray.

Public behaviour follows the reference's tracer/ray_bundle.py:6-223 (constructor keywords,
get_<prop>(selector) / set_<prop>(value, selector) accessors for the base properties and any extra
keyword property, inherit, +, delete_rays, empty_bund, concatenate_rays).  Storage differs: columns
live in one dict, accessors are synthesised on attribute lookup, and `columns_soa()` hands the rows
of the (3,N) arrays to the C-ABI as a structure-of-arrays without copying when they are contiguous.
A LazySourceBundle (sources.py) is a RayBundle whose columns are produced on the device on first
access, so that the engine can fuse source generation into the trace kernel.
"""
import numpy as N

_BASE = ('vertices', 'directions', 'energy', 'parents', 'ref_index')


class RayBundle(object):
    def __init__(self, vertices=None, directions=None, energy=None, parents=None, ref_index=None, **kwds):
        object.__setattr__(self, '_cols', {})
        object.__setattr__(self, '_check_attr', [])
        given = dict(zip(_BASE, (vertices, directions, energy, parents, ref_index)))
        given.update(kwds)
        for name, val in given.items():
            self._create_property(name, val)

    # -- dynamic accessors --------------------------------------------------------------------
    def _create_property(self, propname, init_val):
        attr = '_' + propname
        if attr not in self._check_attr:
            self._check_attr.append(attr)
        if init_val is not None:
            self._cols[propname] = init_val

    def _materialize(self):
        """Hook for lazily generated bundles."""
        return None

    def __getattr__(self, name):
        # only called when normal lookup fails
        if name.startswith('get_') and ('_' + name[4:]) in self._check_attr:
            prop = name[4:]

            def getter(selector=None):
                self._materialize()
                col = self._cols[prop]
                return col if selector is None else col[..., selector]
            return getter
        if name.startswith('set_') and ('_' + name[4:]) in self._check_attr:
            prop = name[4:]

            def setter(new_val, selector=None):
                self._materialize()
                if selector is None:
                    self._cols[prop] = new_val
                else:
                    self._cols[prop][..., selector] = new_val
            return setter
        if name.startswith('_') and name in self.__dict__.get('_check_attr', ()):
            self._materialize()
            try:
                return self._cols[name[1:]]
            except KeyError:
                raise AttributeError(name)
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name.startswith('_') and name in self._check_attr:
            self._cols[name[1:]] = value
        else:
            object.__setattr__(self, name, value)

    def _has_column(self, propname):
        self._materialize()
        return propname in self._cols

    def has_property(self, propname):
        return ('_' + propname) in self._check_attr

    def get_num_rays(self):
        return self.get_vertices().shape[1]

    # -- bundle algebra -------------------------------------------------------------------------
    def inherit(self, selector=N.s_[:], vertices=None, direction=None, energy=None, parents=None,
                ref_index=None, **kwds):
        """
        New bundle: the given properties, and every other property of this bundle taken at
        `selector` (ray_bundle.py:117-143).
        """
        vals = dict((a[1:], None) for a in self._check_attr)
        vals.update(vertices=vertices, directions=direction, energy=energy, parents=parents, ref_index=ref_index)
        vals.update(kwds)
        for prop in list(vals):
            if vals[prop] is None and self._has_column(prop):
                vals[prop] = self._cols[prop][..., selector]
        return RayBundle(**vals)

    def __add__(self, added):
        out = RayBundle()
        for attr in self._check_attr:
            prop = attr[1:]
            if self._has_column(prop) and added._has_column(prop):
                out._create_property(prop, N.hstack((self._cols[prop], added._cols[prop])))
        return out

    @staticmethod
    def empty_bund():
        z3 = N.zeros((3, 0))
        return RayBundle(vertices=z3, directions=z3.copy(), energy=N.array([]), parents=N.array([], dtype=int),
                         ref_index=N.array([]))

    def delete_rays(self, selector):
        n = self.get_num_rays()
        if selector is None:
            selector = N.arange(n)
        return self.inherit(N.delete(N.arange(n), selector))

    # -- C-ABI view -------------------------------------------------------------------------------
    def columns_soa(self, need_energy=True):
        """
        dict of contiguous 1-D float64 arrays x,y,z,dx,dy,dz[,e,ref_index,wavelength] for the C-ABI.
        Rows of C-contiguous (3,N) float64 arrays are passed as views (no copy).
        A complex `ref_index` (media that attenuate) adds ref_index_im; a polychromatic bundle -- `spectra` (W,N) over
        `wavelengths` (W,N), optics_callables.py:406-413 -- gives spec_wl and spectra instead of wavelength.
        """
        from ._cabi import f64
        v = f64(self.get_vertices())
        d = f64(self.get_directions())
        out = dict(x=v[0], y=v[1], z=v[2], dx=d[0], dy=d[1], dz=d[2], _keep=(v, d))
        if self._has_column('energy'):
            out['e'] = f64(self._cols['energy'])
        elif need_energy:
            raise ValueError("the bundle has no energy column")
        if self._has_column('ref_index'):
            ri = N.asarray(self._cols['ref_index'])
            if N.iscomplexobj(ri):
                out['ref_index'] = f64(ri.real)
                out['ref_index_im'] = f64(ri.imag)
            else:
                out['ref_index'] = f64(ri)
        if self._has_column('wavelengths'):
            wl = N.asarray(self._cols['wavelengths'])
            if wl.ndim == 2:
                out['spec_wl'] = f64(wl)
            else:
                out['wavelength'] = f64(wl)
        if self._has_column('spectra'):
            sp = f64(self._cols['spectra'])
            if 'spec_wl' not in out or out['spec_wl'].shape != sp.shape:
                raise ValueError("a polychromatic bundle carries `spectra` and `wavelengths` of the same (W, N) shape")
            out['spectra'] = sp
        return out

    def is_polychromatic(self):
        return self._has_column('spectra')

    def has_complex_index(self):
        return self._has_column('ref_index') and N.iscomplexobj(self._cols['ref_index'])


def concatenate_rays(bundles):
    """Merge bundles in order; properties are those set in the first one (ray_bundle.py:197-223)."""
    if len(bundles) == 0:
        return RayBundle.empty_bund()
    for b in bundles:              # source bundles still described by their generator are generated now
        if hasattr(b, '_materialize'):
            b._materialize()
    out = RayBundle()
    first = bundles[0]
    for attr in first._check_attr:
        prop = attr[1:]
        if first._has_column(prop):
            out._create_property(prop, N.hstack([b._cols[prop] for b in bundles]))
    return out
