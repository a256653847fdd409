"""
Optics managers: callables `optics(geometry, rays, selector) -> RayBundle` plus accountants.

Class names, constructor arguments and result conventions follow the reference's
tracer/optics_callables.py (citations per class).  A native optics class only holds parameters:
`_native()` gives its row in the device table (optics kind + up to 8 parameters + optional table),
and `__call__` runs that kind's device code on the selected hits through trc_optics_apply, so the
per-surface protocol and the fused engines share one implementation (csrc/trc_core.h, trc_shade).

Accountant-wrapped classes (`ReflectiveReceiver`, `OneSidedRealReflectiveDetector`,
`LambertianAbsorberLocationDirectional`, ...) are not generated eagerly like the reference does
(~20.7k classes at import, optics_callables.py:2043-2092) but synthesised on first attribute access
from the same naming algorithm (module __getattr__), giving the same name -> accountant-list map.
"""
import ctypes as C
from itertools import combinations
from copy import deepcopy

import numpy as N
from .deferred import settle_marks

from . import _cabi, rng
from .geometry_manager import fill_desc
from .ray_bundle import RayBundle


def copy_optical_manager(opt):
    """An optics instance must not be shared by two surfaces (optics_callables.py:82-91)."""
    newopt = deepcopy(opt)
    if hasattr(newopt, 'reset'):
        newopt.reset()
    return newopt


# --------------------------------------------------------------------------------------------------
# native optics
# --------------------------------------------------------------------------------------------------
class NativeOptics(object):
    """Base of the optics whose behaviour is implemented on the device."""
    _splits = False      # may emit two rays per hit

    def _native(self):
        """(optics_kind, params, extra)"""
        raise NotImplementedError

    def __call__(self, geometry, rays, selector):
        selector = N.asarray(selector)
        if len(selector) == 0:
            return RayBundle.empty_bund()
        ctx = _cabi.get_context()
        kind, params, extra = self._native()
        extra = _cabi.f64(extra)
        desc = _cabi.SurfaceDesc()
        fill_desc(desc, geometry._working_frame, _cabi.GM_FLAT_INF, [], kind, params,
                  extra_off=0 if len(extra) else -1, extra_len=len(extra))
        n = len(selector)
        d = _cabi.f64(rays.get_directions(selector))
        e = _cabi.f64(rays.get_energy(selector))
        ri = im = wl = swl = sp = mat = None
        if rays._has_column('ref_index'):
            ri = N.asarray(rays.get_ref_index(selector))
            if N.iscomplexobj(ri):
                ri, im = _cabi.f64(ri.real), _cabi.f64(ri.imag)
            else:
                ri = _cabi.f64(ri)
        poly = rays._has_column('spectra')
        if poly:                                    # a spectrum per ray over its own wavelength grid, both (W, N)
            swl = _cabi.f64(N.asarray(rays.get_wavelengths())[:, selector])
            sp = _cabi.f64(N.asarray(rays.get_spectra())[:, selector])
        elif rays._has_column('wavelengths'):
            wl = _cabi.f64(rays.get_wavelengths(selector))
        mats = getattr(self, '_materials', None)
        if mats is not None:
            if wl is None or ri is None:
                raise ValueError("refraction between materials needs `wavelengths` and `ref_index` columns on the bundle")
            from .scene import material_rows
            mat = material_rows(mats, wl)
            if im is None:
                im = N.zeros(n)
        nrm = _cabi.f64(geometry.get_normals())
        hit = _cabi.f64(geometry.get_intersection_points_global())
        rid = N.arange(n, dtype=N.uint64)
        org = _cabi.f64(rays.get_vertices(selector))          # origins: path lengths of the attenuating optics
        rin = _cabi.make_rays(n, org[0], org[1], org[2], dx=d[0], dy=d[1], dz=d[2], e=e, ref_index=ri, wavelength=wl, rid=rid,
                              ref_index_im=im, spec_wl=swl, spectra=sp, mat=mat)
        m = 2 * n
        o = dict((k, N.empty(m)) for k in ('x', 'y', 'z', 'dx', 'dy', 'dz', 'e', 'ref', 'wl'))
        par = N.empty(m, dtype=N.int64)
        oim = N.empty(m) if im is not None else None
        osp = N.empty((sp.shape[0], m)) if poly else None
        oswl = N.empty((sp.shape[0], m)) if poly else None
        rout = _cabi.make_rays(m, o['x'], o['y'], o['z'], o['dx'], o['dy'], o['dz'], o['e'], parent=par,
                               ref_index=o['ref'], wavelength=o['wl'], ref_index_im=oim, spec_wl=oswl, spectra=osp)
        _cabi.check(ctx.lib.trc_optics_apply(
            ctx.handle, C.byref(desc), len(extra), _cabi.ptr(extra) if len(extra) else None, C.byref(rin),
            _cabi.ptr(hit[0]), _cabi.ptr(hit[1]), _cabi.ptr(hit[2]), _cabi.ptr(nrm[0]), _cabi.ptr(nrm[1]),
            _cabi.ptr(nrm[2]), rng.next_seed(), 1, C.byref(rout)))
        k = rout.n
        par = par[:k]
        src = selector[par]
        kw = {}
        if ri is not None:
            kw['ref_index'] = o['ref'][:k].copy() if oim is None else o['ref'][:k] + 1j * oim[:k]
        if poly:
            kw['spectra'] = osp[:, :k].copy()
        return rays.inherit(src, vertices=N.vstack((o['x'][:k], o['y'][:k], o['z'][:k])),
                            direction=N.vstack((o['dx'][:k], o['dy'][:k], o['dz'][:k])),
                            energy=o['e'][:k].copy(), parents=src, **kw)


class Transparent(NativeOptics):
    """Rays go through unchanged (optics_callables.py:93-113)."""
    def __init__(self):
        pass

    def _native(self):
        return _cabi.OPT_TRANSPARENT, [], []


class Reflective(NativeOptics):
    """Specular mirror absorbing a fixed fraction (optics_callables.py:116-140)."""
    def __init__(self, absorptivity):
        self._abs = absorptivity

    def _native(self):
        return _cabi.OPT_REFLECTIVE, [self._abs], []


perfect_mirror = Reflective(0)


class OneSidedReflective(Reflective):
    """Mirror whose back side (rays travelling along the surface's +z) absorbs everything (:195-212)."""
    def _native(self):
        return _cabi.OPT_ONE_SIDED_REFLECTIVE, [self._abs], []


class RealReflective(NativeOptics):
    """Mirror with Gaussian slope error, bi-variate or radial (optics_callables.py:214-269)."""
    def __init__(self, absorptivity, sigma, bi_var=False):
        self._abs = absorptivity
        self._sig = sigma
        self.bi_var = bi_var

    def _native(self):
        return _cabi.OPT_REAL_REFLECTIVE, [self._abs, self._sig, 1. if self.bi_var == True else 0.], []


class RealReflective_IAM(RealReflective):
    """RealReflective with the Incidence Angle Modifier on its reflected fraction, evaluated on the ideal normal
    (optics_callables.py:320-329; c = 1)."""
    def __init__(self, absorptivity, a_r, sigma, bi_var=False):
        RealReflective.__init__(self, absorptivity, sigma, bi_var)
        self.a_r, self.c = a_r, 1

    def _native(self):
        kind, params, extra = RealReflective._native(self)
        return kind, list(params) + [self.a_r, self.c], extra


class OneSidedRealReflective(RealReflective):
    """One-sided version of RealReflective (optics_callables.py:492-504)."""
    def _native(self):
        return _cabi.OPT_ONE_SIDED_REAL_REFLECTIVE, [self._abs, self._sig, 1. if self.bi_var == True else 0.], []


class Lambertian(NativeOptics):
    """Diffuse reflector, cosine-weighted within ang_range of the normal (optics_callables.py:143-176)."""
    def __init__(self, absorptivity=0., ang_range=N.pi / 2.):
        self._abs = absorptivity
        self._ang_range = ang_range

    def _native(self):
        return _cabi.OPT_LAMBERTIAN, [self._abs, self._ang_range], []


class Reflective_IAM(Reflective):
    """
    Specular mirror whose reflected fraction follows the Incidence Angle Modifier of Martin and Ruiz
    (optics_callables.py:271-300): E (1 - absorptivity) (1 - exp(-cos(theta)^c / a_r)) / (1 - exp(-1 / a_r)).  (The
    reference's __call__ constructs a Reflective instead of calling it, :295; its Lambertian and RealReflective siblings
    run, and this class does what they do.)
    """
    def __init__(self, absorptivity, a_r, c=1):
        Reflective.__init__(self, absorptivity)
        self.a_r, self.c = a_r, c

    def _native(self):
        return _cabi.OPT_REFLECTIVE, [self._abs, self.a_r, self.c], []


class Lambertian_IAM(Lambertian):
    """Lambertian reflector with the Incidence Angle Modifier on its reflected fraction (optics_callables.py:302-318)."""
    def __init__(self, absorptivity, a_r, c=1):
        Lambertian.__init__(self, absorptivity)
        self.a_r, self.c = a_r, c

    def _native(self):
        return _cabi.OPT_LAMBERTIAN, [self._abs, self._ang_range, 0., 0., self.a_r, self.c], []


class LambertianAbsorbant(Lambertian):
    """
    Opaque Lambertian wall at the boundary of an absorbing volume (optics_callables.py:891-906 on Absorbant.attenuate
    :874-889): the ray is attenuated by exp(-attenuation_coefficient * path * scaling) over the distance it travelled to the
    wall, then loses `absorptivity` of what is left.
    """
    def __init__(self, absorptivity=0., attenuation_coefficient=0., ang_range=N.pi / 2., scaling=1.):
        Lambertian.__init__(self, absorptivity, ang_range)
        self.a_c = float(N.ravel(attenuation_coefficient)[0])
        self._scaling = scaling

    def _native(self):
        kind, params, extra = Lambertian._native(self)
        return kind, list(params[:2]) + [self.a_c, self._scaling], extra


class LambertianSpecular(NativeOptics):
    """Each ray is specular with probability `specularity`, else Lambertian (optics_callables.py:553-585)."""
    def __init__(self, absorptivity=0., specularity=0.5):
        self._abs = absorptivity
        self.specularity = specularity

    def _native(self):
        return _cabi.OPT_LAMBERTIAN_SPECULAR, [self._abs, self.specularity], []


class SemiLambertian(NativeOptics):
    """
    Mirror for rays arriving at more than `angular_range` from the normal, Lambertian reflector (into the same cone) for the
    others (optics_callables.py:506-531, as its docstring and its two parent calls describe it; the reference's own
    __call__ indexes the direction array by rows, :525, and raises).
    """
    def __init__(self, absorptivity=0., angular_range=N.pi / 2.):
        self._abs = absorptivity
        self._ang_range = angular_range

    def _native(self):
        return _cabi.OPT_SEMI_LAMBERTIAN, [self._abs, self._ang_range], []


class LambertianSpecular_IAM(LambertianSpecular):
    """
    optics_callables.py:588-627.  The reference evaluates its incidence-angle factor on a direction array it has just filled
    with zeros (:607-609), so cos(theta) = 0 and the outgoing energy is the incident energy whatever the absorptivity: that
    is what its users get, and what this class gives (a LambertianSpecular that absorbs nothing).
    """
    def __init__(self, absorptivity=0., specularity=0.5, a_r=0.16):
        LambertianSpecular.__init__(self, 0., specularity)
        self._abs_declared = absorptivity
        self.a_r = a_r


class Reflective_spectral(NativeOptics):
    """Mirror whose absorptance is interpolated on the ray wavelength (optics_callables.py:178-193)."""
    def __init__(self, absorptances, wavelengths):
        self._wavelengths = wavelengths
        self._absorptances = absorptances

    def _native(self):
        lam = N.ravel(N.asarray(self._wavelengths, dtype=float))
        ab = N.ravel(N.asarray(self._absorptances, dtype=float))
        return _cabi.OPT_REFLECTIVE_SPECTRAL, [], N.concatenate((lam, ab)).tolist()


class Lambertian_directional_axisymmetric_piecewise(NativeOptics):
    """Lambertian reflector whose absorptance depends on the incidence angle, piecewise linear
    (optics_callables.py:331-361)."""
    def __init__(self, thetas, absorptance_th, specularity=0.):
        self.thetas = thetas
        self.abs_th = absorptance_th
        self.specularity = specularity

    def _native(self):
        th = N.ravel(N.asarray(self.thetas, dtype=float))
        ab = N.ravel(N.asarray(self.abs_th, dtype=float))
        return _cabi.OPT_LAMBERTIAN_DIRECTIONAL, [], N.concatenate((th, ab)).tolist()


class LambertianSpecular_directional_axisymmetric_piecewise(Lambertian_directional_axisymmetric_piecewise):
    """Angle-dependent absorptance as above; each ray is mirrored with probability `specularity`, else scattered
    (optics_callables.py:427-455)."""
    def _native(self):
        kind, params, extra = Lambertian_directional_axisymmetric_piecewise._native(self)
        return kind, [1., float(self.specularity)], extra


class Lambertian_piecewise_Specular_directional_axisymmetric_piecewise(NativeOptics):
    """Angle-dependent absorptance and angle-dependent probability of a specular reflection, both piecewise linear on the
    incidence angle (optics_callables.py:457-487)."""
    def __init__(self, thetas, absorptance_th, specularity_th):
        self.thetas = thetas
        self.abs_th = absorptance_th
        self.spec_th = specularity_th

    def _native(self):
        cols = [N.ravel(N.asarray(c, dtype=float)) for c in (self.thetas, self.abs_th, self.spec_th)]
        if not (len(cols[0]) == len(cols[1]) == len(cols[2])):
            raise ValueError('thetas, absorptance_th and specularity_th must have the same length')
        return _cabi.OPT_LAMBERTIAN_DIRECTIONAL, [2.], N.concatenate(cols).tolist()


class Lambertian_directional_axisymmetric_piecewise_spectral(NativeOptics):
    """Same with absorptance tabulated on (incidence angle, wavelength), bilinear (optics_callables.py:363-391).
    Arguments outside the table are clamped to its edge (the reference's RegularGridInterpolator raises)."""
    def __init__(self, thetas, absorptance, wavelengths):
        self._thetas, self._wavelengths = N.unique(thetas), N.unique(wavelengths)
        self._abs = N.reshape(absorptance, (len(self._thetas), len(self._wavelengths)))

    def _native(self):
        tab = N.concatenate(([len(self._thetas), len(self._wavelengths)], self._thetas, self._wavelengths, N.ravel(self._abs)))
        return _cabi.OPT_LAMBERTIAN_DIRECTIONAL_SPECTRAL, [], tab.tolist()


class TabulatedMaterial(object):
    """Complex refractive index m = n + i k tabulated over wavelength, linear in between (what the reference's
    ray_trace_utils.optical_constants materials do with interp1d)."""
    def __init__(self, wavelengths, n, k):
        self.wavelengths = N.asarray(wavelengths, dtype=float)
        self.n = N.asarray(n, dtype=float)
        self.k = N.asarray(k, dtype=float)

    def m(self, lambdas):
        return N.interp(lambdas, self.wavelengths, self.n) + 1j * N.interp(lambdas, self.wavelengths, self.k)

    def table(self):
        return self.wavelengths, self.n, self.k


class FresnelConductorHomogenous(NativeOptics):
    """Mirror-like interface to an absorbing medium (metal): unpolarised Fresnel reflectance from the complex index
    material.m(lambda) (optics_callables.py:1523-1558).  The material must expose its table: a TabulatedMaterial, or
    an object holding a scipy interp1d in `m_func` like the reference's OpticalMaterialFromFile."""
    def __init__(self, n1, material):
        self._n1 = n1
        self._material = material

    def _native(self):
        mat = self._material
        if hasattr(mat, 'table'):
            lam, n, k = mat.table()
        elif hasattr(mat, 'm_func') and hasattr(mat.m_func, 'x'):
            lam, n, k = mat.m_func.x, N.real(mat.m_func.y), N.imag(mat.m_func.y)
        else:
            raise NotImplementedError("FresnelConductorHomogenous needs a tabulated material")
        return _cabi.OPT_FRESNEL_CONDUCTOR, [self._n1], N.concatenate((N.ravel(lam), N.ravel(n), N.ravel(k))).tolist()


class RefractiveHomogenous(NativeOptics):
    """
    Interface between two homogeneous media of real indices n1, n2 (optics_callables.py:1186-1296):
    Snell + unpolarised Fresnel; single_ray=True picks reflection or refraction per ray, False emits
    both (reflected block first); sigma perturbs the normal.
    """
    def __init__(self, n1, n2, single_ray=True, sigma=None):
        self._ref_idxs = (n1, n2)
        self._single_ray = single_ray
        self._sigma = sigma

    @property
    def _splits(self):
        return not self._single_ray

    def toggle_ref_idx(self, current, wavelengths=None):
        return N.where(current == self._ref_idxs[0], self._ref_idxs[1], self._ref_idxs[0])

    def _native(self):
        return _cabi.OPT_REFRACTIVE_HOMOGENOUS, [self._ref_idxs[0], self._ref_idxs[1],
                                                 1. if self._single_ray else 0.,
                                                 -1. if self._sigma is None else self._sigma], []


class RefractiveTransmissiveHomogenous(RefractiveHomogenous):
    """
    RefractiveHomogenous with Beer-Lambert attenuation in the media (optics_callables.py:1326-1348 on Absorbant.attenuate
    :874-889): a ray arriving through the medium of index n1 is attenuated by exp(-attenuation_coefficients[0] * path * scaling),
    through n2 by attenuation_coefficients[1] (a single coefficient applies to both), before it is reflected / refracted.
    """
    def __init__(self, n1, n2, attenuation_coefficients, single_ray=True, sigma=None, scaling=1.):
        RefractiveHomogenous.__init__(self, n1, n2, single_ray, sigma)
        a_c = N.ravel(N.array(attenuation_coefficients, dtype=float))
        if len(a_c) not in (1, 2):
            raise ValueError('one attenuation coefficient, or one per medium')
        self.a_c = a_c
        self._scaling = scaling

    def _native(self):
        kind, params, extra = RefractiveHomogenous._native(self)
        return kind, list(params) + [self.a_c[0], self.a_c[-1], self._scaling, 1.], extra


class RefractiveAbsorbantHomogenous(RefractiveTransmissiveHomogenous):
    """
    optics_callables.py:1298-1324.  The reference keeps the coefficients only when both are None (:1313-1316, and then
    attenuates with the imaginary part of a complex refractive index, which the device rays do not carry); given
    coefficients are what its docstring describes, and what is done here.
    """
    def __init__(self, m1, m2, single_ray=True, sigma=None, attenuation_coefficient_1=None, attenuation_coefficient_2=None, scaling=1.):
        if attenuation_coefficient_1 is None or attenuation_coefficient_2 is None:
            raise NotImplementedError('attenuation from complex refractive indices is outside the native path: give both coefficients')
        RefractiveTransmissiveHomogenous.__init__(self, m1, m2, [attenuation_coefficient_1, attenuation_coefficient_2], single_ray, sigma, scaling)


class RefractiveScatteringHomogenous(RefractiveHomogenous):
    """
    RefractiveHomogenous between two media that scatter (optics_callables.py:1350-1376 on Scattering :946-1036 and
    RefractiveScattering :1108-1172): on its way to the surface a ray draws a free path -ln(R) / s_c in the medium it travels
    through (optics.py:214-239); if that is shorter than the way, it is scattered there into a direction drawn from the medium's
    Henyey-Greenstein phase function about its own direction (sampling.py:150-168) and keeps its energy and medium; otherwise
    it meets the surface like a ray of RefractiveHomogenous (reflected or refracted, one ray).  A scattered ray never reached the
    surface: tallies, flux maps and accountants of the surface do not see it.

    s_c1, s_c2: scattering coefficients (1/m) of the media of index n1, n2; g_HG_1, g_HG_2: their asymmetry factors (-1..1).
    The medium is told by the refractive index the ray carries, so n1 != n2 is required (the reference tells it by a
    scattering-coefficient column toggled together with the index).  The reference's classes do not run (Scattering._scatter
    and RefractiveScattering.__init__ read names they never define): this follows their docstrings and the body kept in comments
    at :1385-1470; the two pure functions they are built on are pinned by tests/golden/scattering.npz.
    Device only (fast and ordered engines); single_ray=True.
    """
    def __init__(self, n1, n2, s_c1, s_c2, g_HG_1, g_HG_2, single_ray=True, sigma=None):
        if not single_ray:
            raise NotImplementedError('scattering optics emit one ray per interaction (single_ray=True)')
        if n1 == n2:
            raise ValueError('the media are told apart by their refractive indices: n1 != n2')
        RefractiveHomogenous.__init__(self, n1, n2, True, sigma)
        self._s_cs = [float(s_c1), float(s_c2)]
        self._g = [float(g_HG_1), float(g_HG_2)]

    def get_media(self, current_ref_idx):
        return N.array(N.asarray(current_ref_idx) != self._ref_idxs[0], dtype=int)

    def _native(self):
        kind, params, extra = RefractiveHomogenous._native(self)
        return _cabi.OPT_REFRACTIVE_SCATTERING, list(params) + [0., 0., 1., 0.], self._s_cs + self._g

    def __call__(self, geometry, rays, selector):
        raise NotImplementedError('scattering optics run on the device engines (ray_tracer with engine "auto", "fast" or "ordered")')


class Refractive(NativeOptics):
    """
    Interface between two media whose complex refractive indices depend on the wavelength (optics_callables.py:726-858):
    material_1, material_2 are objects with m(wavelengths) -> complex index, like those of the reference's
    ray_trace_utils.optical_constants (TabulatedMaterial here).  Rays need `wavelengths` and a `ref_index` column holding the index
    of the medium they start in (complex or real); a ray whose index equals material_1's at its wavelength enters material_2, any
    other enters material_1 (:750-751).  Snell's law on the real parts, the real part of the complex Fresnel reflectance as the
    reference computes it (:838-840); single_ray / sigma as RefractiveHomogenous.
    The materials are evaluated by their own m() once per ray (the wavelength does not change along a path) and travel with the
    rays (trc_rays.mat): ordered engine and per-surface protocol.
    """
    _attenuate = False
    _scaling = 1.

    def __init__(self, material_1, material_2, single_ray=True, sigma=None):
        self._materials = [material_1, material_2]
        self._single_ray = single_ray
        self._sigma = sigma

    @property
    def _splits(self):
        return not self._single_ray

    def toggle_ref_idx(self, m1, wavelengths):
        mat_0 = self._materials[0].m(wavelengths)
        return N.where(m1 == mat_0, self._materials[1].m(wavelengths), mat_0)

    def _native(self):
        # opt[4], opt[5]: rows of the two materials in trc_rays.mat -- 0, 1 for a single call, the scene's numbering in a scene
        return _cabi.OPT_REFRACTIVE_MATERIAL, [1. if self._single_ray else 0., -1. if self._sigma is None else self._sigma,
                                               1. if self._attenuate else 0., self._scaling, 0., 1.], []


class RefractiveAbsorbant(Refractive):
    """
    Refractive whose media attenuate (optics_callables.py:908-944 on Absorbant.attenuate :874-889): the outgoing rays lose
    exp(-4 pi k L / lambda) of their energy over the path L the incident ray travelled, with k the imaginary part of the index the
    OUTGOING ray carries (optics.attenuations, optics.py:205-212, is handed the new bundle's index, :882).
    The constructor's coefficients are kept as the reference reads them: when either is given the attenuation follows the complex
    indices (:929-933 stores None); when both are None the reference stores [None, None] and its attenuate() raises on the first
    hit -- here the constructor does.
    """
    _attenuate = True

    def __init__(self, material_1, material_2, single_ray=True, sigma=None, attenuation_coefficient_1=None,
                 attenuation_coefficient_2=None, scaling=1.):
        if attenuation_coefficient_1 is None and attenuation_coefficient_2 is None:
            raise NotImplementedError("RefractiveAbsorbant without coefficients cannot run in the reference either (optics_callables.py:884 "
                                      "reads an attribute the class never sets); pass any coefficient to attenuate by the complex indices")
        Refractive.__init__(self, material_1, material_2, single_ray, sigma)
        self._scaling = scaling


class Lambertian_directional_axisymmetric_piecewise_Polychromatic(NativeOptics):
    """
    Diffuse wall for polychromatic bundles (optics_callables.py:393-425): every ray carries a spectrum `spectra` (W,N) sampled at
    `wavelengths` (W,N); sample w is scaled by 1 - absorptance(theta_in, lambda_w) (bilinear on the (thetas, wavelengths) grid) and
    the ray's energy becomes the trapezoid integral of the scaled spectrum.  Ordered engine and per-surface protocol.
    """
    def __init__(self, thetas, absorptance, wavelengths):
        thetas, wavelengths = N.unique(thetas), N.unique(wavelengths)
        self.thetas, self.wavelengths = thetas, wavelengths
        self.absorptance = N.reshape(absorptance, (len(thetas), len(wavelengths)))

    def _native(self):
        tab = N.concatenate(([len(self.thetas), len(self.wavelengths)], self.thetas, self.wavelengths, self.absorptance.ravel()))
        return _cabi.OPT_LAMBERTIAN_POLYCHROMATIC, [], tab.tolist()


# --------------------------------------------------------------------------------------------------
# optics composed on the host.  They are ordinary optics callables of the four-step protocol (their scenes are traced by
# TracerEngine with engine='protocol', the native surfaces and the optics they wrap still running on the device per call).
# --------------------------------------------------------------------------------------------------
class BiFacial(object):
    """
    Different optics for rays arriving on the front (+z of the surface) and on the back (optics_callables.py:1877-1926):
    both callables are run on the selected hits and the outgoing rays of the side each ray arrived on are kept, back side
    first.
    """
    def __init__(self, OpticsCallable_front, OpticsCallable_back):
        self.OpticsCallable_front = OpticsCallable_front
        self.OpticsCallable_back = OpticsCallable_back

    def __call__(self, geometry, rays, selector):
        from .ray_bundle import concatenate_rays
        proj = N.around(N.sum(rays.get_directions(selector) * geometry.up()[:, None], axis=0), decimals=6)
        back = proj > 0.
        parts = []
        if back.any():
            parts.append(self.OpticsCallable_back(geometry, rays, selector).inherit(N.nonzero(back)[0]))
        if not back.all():
            parts.append(self.OpticsCallable_front(geometry, rays, selector).inherit(N.nonzero(~back)[0]))
        return concatenate_rays(parts) if len(parts) > 1 else parts[0]

    def get_all_hits(self):
        def hits_of(side):
            try:
                return side.get_all_hits()
            except Exception:
                return []
        return hits_of(self.OpticsCallable_front), hits_of(self.OpticsCallable_back)

    def reset(self):
        for side in (self.OpticsCallable_front, self.OpticsCallable_back):
            if hasattr(side, 'reset'):
                side.reset()


class PeriodicBoundary(NativeOptics):
    """
    Periodic boundary condition (optics_callables.py:690-723): a ray that lands on the surface stops there (a zero-energy
    stub keeps the tree connected) and continues, unchanged, from the hit point translated by `period` along the surface
    normal.  A native kind (TRC_OPT_PERIODIC_BOUNDARY): the fast engines follow the moved ray, the ordered engine records
    stub and moved ray as the reference's bundle holds them.
    """
    def __init__(self, period):
        self.period = period

    def _native(self):
        return _cabi.OPT_PERIODIC_BOUNDARY, [self.period], []


# --------------------------------------------------------------------------------------------------
# accountants (optics_callables.py:1577-1848)
# --------------------------------------------------------------------------------------------------
class Accountant(object):
    shorthand = None

    def __init__(self):
        self.reset()

    def reset(self):
        self._data = []

    def count(self, geometry, rays, selector, new_bundle):
        raise NotImplementedError

    def feed(self, hit):
        """Fused engines: `hit` is a dict with e_in, e_out, points(3,H), directions(3,H), normals()."""
        raise NotImplementedError

    def _empty(self):
        return N.array([])

    def __getstate__(self):
        settle_marks(self)          # (a copy or a pickle of an accountant holds arrays, not promises of the device)
        return self.__dict__

    def __deepcopy__(self, memo):
        settle_marks(self)
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        new.__dict__.update(deepcopy(self.__dict__, memo))
        return new

    def get_data(self):
        # chunks the device engines still owe this accountant are fetched now (deferred.py): the hits of a fast trace stay in the
        # device's hit buffer, the levels of an ordered trace on the device, until somebody reads them
        settle_marks(self)
        chunks = [c for c in self._data if c.shape[-1]]
        if not chunks:
            return self._empty()
        if len(chunks) == 1:
            return chunks[0]        # (no copy of the one chunk a trace left: hstack of the 6.5e6 receiver hits of an NSTTF step took 14 ms)
        return N.hstack(chunks)


class _Vector3Accountant(Accountant):
    def _empty(self):
        return N.array([]).reshape(3, 0)


class LocationAccountant(_Vector3Accountant):
    """Hit points in global coordinates."""
    shorthand = 'Location'

    def count(self, geometry, rays, selector, new_bundle):
        self._data.append(N.asarray(geometry.get_intersection_points_global()))

    def feed(self, hit):
        self._data.append(hit['points'])


class AbsorptionAccountant(Accountant):
    """Energy absorbed by each hit: E_in - E_out."""
    shorthand = 'Absorber'

    def count(self, geometry, rays, selector, new_bundle):
        self._data.append(rays.get_energy(selector) - new_bundle.get_energy())

    def feed(self, hit):
        self._data.append(hit['e_abs'] if 'e_abs' in hit else hit['e_in'] - hit['e_out'])


class AttenuationAccountant(Accountant):
    shorthand = 'Attenuator'

    def count(self, geometry, rays, selector, new_bundle, attenuations):
        self._data.append(attenuations)

    def feed(self, hit):
        pass


class ReceptionAccountant(Accountant):
    """Incident energy of each hit."""
    shorthand = 'Receptor'

    def count(self, geometry, rays, selector, new_bundle):
        self._data.append(rays.get_energy(selector))

    def feed(self, hit):
        self._data.append(hit['e_in'])


class ScatteringAccountant(Accountant):
    """Outgoing energy of each hit."""
    shorthand = 'Scatterer'

    def count(self, geometry, rays, selector, new_bundle):
        self._data.append(new_bundle.get_energy())

    def feed(self, hit):
        self._data.append(hit['e_out'])


class DirectionAccountant(_Vector3Accountant):
    """Incident direction of each hit."""
    shorthand = 'Directional'

    def count(self, geometry, rays, selector, new_bundle):
        self._data.append(rays.get_directions(selector))

    def feed(self, hit):
        self._data.append(hit['directions'])


class NormalAccountant(_Vector3Accountant):
    """Surface normal at each hit."""
    shorthand = 'Normal'

    def count(self, geometry, rays, selector, new_bundle):
        self._data.append(geometry.get_normals())

    def feed(self, hit):
        self._data.append(hit['normals']())


class SpectralAccountant(Accountant):
    shorthand = 'Spectral'

    def count(self, geometry, rays, selector, new_bundle):
        self._data.append(rays.get_wavelengths()[selector])

    def feed(self, hit):
        self._data.append(hit['wavelengths'])


class PolychromaticAccountant(Accountant):
    """Spectral power absorbed by each hit of a polychromatic bundle, and the wavelengths it is sampled at
    (optics_callables.py:1825-1848): get_data() -> (wavelengths (W,H), absorbed spectra (W,H))."""
    shorthand = 'Polychromatic'

    def reset(self):
        self._data = []
        self._wavelengths = []

    def count(self, geometry, rays, selector, new_bundle):
        self._wavelengths.append(N.asarray(new_bundle.get_wavelengths()))
        self._data.append(N.asarray(rays.get_spectra())[:, selector] - N.asarray(new_bundle.get_spectra()))

    def feed(self, hit):
        self._wavelengths.append(hit['wavelengths'])
        self._data.append(hit['spectra_in'] - hit['spectra_out'])

    def get_data(self):
        settle_marks(self)
        if not self._data:
            return N.array([]), N.array([]).reshape(2, 0)
        return N.concatenate(self._wavelengths, axis=-1), N.concatenate(self._data, axis=-1)


# canonical accountant order: energy -> spectral -> location -> directions (optics_callables.py:2060-2071)
_ACCOUNTANT_ORDER = [AbsorptionAccountant, AttenuationAccountant, ReceptionAccountant, ScatteringAccountant,
                     PolychromaticAccountant, SpectralAccountant, LocationAccountant, DirectionAccountant,
                     NormalAccountant]
aliases = {'Receiver': ['Location', 'Absorber'], 'Detector': ['Directional', 'Location', 'Absorber'],
           'Transmitter': ['Location', 'Scatterer']}


def _accountant_suffix_table():
    """
    suffix -> tuple of accountant classes, produced by running the reference's naming loop
    (optics_callables.py:2043-2054, :2075-2083) once on an empty class name: combination names in
    canonical order, then the alias names, where the working name is rewritten cumulatively from one
    alias to the next exactly as the reference does.
    """
    table = {}
    for size in range(1, len(_ACCOUNTANT_ORDER)):
        for accs in combinations(_ACCOUNTANT_ORDER, size):
            if SpectralAccountant in accs and PolychromaticAccountant in accs:
                continue
            name = ''.join(a.shorthand for a in accs)
            table[name] = accs
            work = name
            for alias, parts in aliases.items():
                if all(p in name for p in parts):
                    for p in parts:
                        work = work.replace(p, '')
                    table[work + alias] = accs
    return table


_SUFFIXES = None


class OpticsCallable(object):
    """An optics instance wrapped with accountants (optics_callables.py:1562-1575, :1942-1971)."""
    optics_class = None
    accountant_classes = ()

    def __init__(self, *args, **kwargs):
        self._opt = self.optics_class(*args, **kwargs)
        self.accountants = [a() for a in self.accountant_classes]

    def __call__(self, geometry, rays, selector):
        new_bundle = self._opt(geometry, rays, selector)
        for a in self.accountants:
            a.count(geometry=geometry, rays=rays, selector=selector, new_bundle=new_bundle)
        return new_bundle

    def reset(self):
        for a in self.accountants:
            a.reset()

    def get_all_hits(self):
        return [a.get_data() for a in self.accountants]

    def _native(self):
        return self._opt._native()

    @property
    def _splits(self):
        return getattr(self._opt, '_splits', False)


def _optics_classes():
    g = globals()
    return dict((k, v) for k, v in g.items()
                if isinstance(v, type) and issubclass(v, NativeOptics) and v is not NativeOptics)


def __getattr__(name):
    """Synthesise `<OpticsClass><AccountantSuffix>` classes on demand."""
    global _SUFFIXES
    if name.startswith('__'):
        raise AttributeError(name)
    if _SUFFIXES is None:
        _SUFFIXES = _accountant_suffix_table()
    classes = _optics_classes()
    for cname in sorted(classes, key=len, reverse=True):
        if name.startswith(cname) and name[len(cname):] in _SUFFIXES:
            accs = _SUFFIXES[name[len(cname):]]
            cls = type(name, (OpticsCallable,), {'optics_class': classes[cname], 'accountant_classes': accs})
            globals()[name] = cls
            return cls
    raise AttributeError("module %r has no attribute %r" % (__name__, name))


def native_optics_of(opt):
    """The NativeOptics instance behind an optics object (itself or inside an accountant wrapper)."""
    if isinstance(opt, OpticsCallable):
        opt = opt._opt
    return opt if isinstance(opt, NativeOptics) else None
