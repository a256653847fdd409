"""
Ray-traced view factors of axisymmetric cavities (reference: emissive_losses/view_factors_3D.py).

Every element of the cavity wall emits a Lambertian bundle (sources.vf_cylinder_bundle / vf_frustum_bundle / disk_bundle),
the bundle is traced for ONE interaction against black receivers (`itmax = 1`, :142, :491), and the absorbed energy is
allocated to the receiving elements; passes are repeated until the 3-sigma confidence interval of every view factor, the
reciprocity rule and the summation rule meet the requested precision (`RTVF.test_precision`, :44-112).

What is different here: the trace runs on the GPU (the fast engine; one call per emitting element instead of `procs`
worker processes), and the allocation (`alloc_VF`, :598-674: fetch every hit, loop over the elements with boolean masks) is
one device pass over the hit buffer (`trc_scene_bin_hits`) that returns the row of the matrix.  The statistics are the
reference's, formula for formula.  `FONaR_RTVF` (:114-357) needs the FONaR receiver model, which the reference repository
does not contain (`import FONaR`, view_factors_3D_test.py:1), and is not offered.
"""
import time

import numpy as N

from .. import _cabi
from ..assembly import Assembly
from ..object import AssembledObject
from ..surface import Surface
from ..flat_surface import RoundPlateGM
from ..cylinder import FiniteCylinder
from ..cone import FiniteCone, ConicalFrustum
from ..spatial_geometry import translate, rotx
from ..sources import disk_bundle, vf_cylinder_bundle, vf_frustum_bundle
from ..tracer_engine import TracerEngine
from .. import optics_callables as opt


class RTVF(object):
    """
    Monte-Carlo view-factor estimation with an online confidence interval.
    num_rays: rays per emitting element and pass; precision: threshold on the 3-sigma interval of each view factor (half of
    it, 'absolute') or on its relative value ('relative') and on the summation rule; precision_rec: threshold on the
    reciprocity rule A_i F_ij = A_j F_ji (defaults to precision).
    """
    def __init__(self, num_rays=10000, precision=0.01, precision_option='absolute', precision_rec=None):
        self.num_rays = num_rays
        self.precision = precision
        self.precision_rec = precision if precision_rec is None else precision_rec
        self.stdev = N.inf
        self.precision_option = precision_option

    def _init_statistics(self, n):
        self.VF = N.zeros((n, n))
        self.progress = N.ones((n, n), dtype=bool)
        self.VF_esperance = N.zeros((n, n))
        self.Qsum = N.zeros((n, n))
        self.stdev_VF = N.zeros((n, n))
        self.stdev_reciprocity = N.zeros((n, n))
        self.p = N.zeros(n)

    def reset_opt(self):
        """forget the energy stored on the surfaces before the next emitter is traced (:37-42)"""
        for s in self.A.get_surfaces():
            s.get_optics_manager().reset()
        if getattr(self, 'engine', None) is not None:
            self.engine.reset_tallies()

    def test_precision(self):
        """
        Update the running mean and the weighted running variance of every view factor with the pass just traced
        (self.VF, self.ray_counts rays per emitter, self.p rays per emitter so far including this pass), then mark in
        self.progress the entries that still miss one of the three criteria (:44-112).
        """
        rays_now = N.vstack(self.ray_counts)
        rays_all = N.vstack(self.p)
        rays_before = rays_all - rays_now
        A_i = N.ones(N.shape(self.VF_esperance)) * N.vstack(self.areas)

        # weighted online variance (West): Q += w (W - w) / W (x - mean)^2, interval = 3 sqrt(Q / (W - 1)) / sqrt(W)
        self.Qsum = self.Qsum + rays_now * rays_before / rays_all * (self.VF - self.VF_esperance) ** 2.
        self.stdev_VF = 3. * N.sqrt(self.Qsum / (rays_all - 1.)) / N.sqrt(rays_all)
        self.VF_esperance = (self.VF_esperance * rays_before + self.VF * rays_now) / rays_all

        AiFij = self.VF_esperance * A_i
        self.VF_reciprocity = N.abs(AiFij - AiFij.T)

        if self.precision_option == 'absolute':
            interval_ok = self.stdev_VF <= self.precision / 2.
            weighted = self.stdev_VF * A_i
            reciprocity_ok = (weighted + weighted.T) <= self.precision_rec
        elif self.precision_option == 'relative':
            with N.errstate(divide='ignore', invalid='ignore'):
                rel = self.stdev_VF / self.VF_esperance
                rel[N.isnan(rel)] = 0.
                interval_ok = rel <= self.precision
                weighted = A_i * self.stdev_VF
                rel_rec = (weighted + weighted.T) / AiFij
            rel_rec[N.isnan(rel_rec)] = 0.
            rel_rec[N.isinf(rel_rec)] = 0.
            negligible = AiFij < N.vstack(self.precision_rec * N.amax(AiFij, axis=1))
            reciprocity_ok = N.logical_or(rel_rec <= self.precision_rec, negligible)
        else:
            raise ValueError("precision_option is 'absolute' or 'relative'")

        summation_ok = N.abs(N.sum(self.VF_esperance, axis=1) - 1.) < self.precision
        # NB the (n,) summation test broadcasts over the columns of the (n, n) tests, as it does in the reference (:98)
        self.progress = N.logical_not(N.logical_and(summation_ok, N.logical_and(interval_ok, reciprocity_ok)))


class Two_N_parameters_cavity_RTVF(RTVF):
    """
    Axisymmetric cavity made of an aperture disc, N frusta (or cylinders, or annuli) and a closing cone (or disc).

    apertureRadius; frustaRadii[k], frustaDepths[k]: radius at the far end and depth of the k-th section, following the
    profile from the aperture inwards (a negative depth folds the profile back towards the aperture); coneDepth: depth of
    the closing cone (> 0 outgoing, 0 flat disc, < 0 re-entrant); el_FRUs[k], el_CON: number of elements of equal depth each
    section / the cone is divided into.  After construction: VF_esperance (the matrix, aperture first, then the elements
    along the profile), areas, stdev_VF, p (rays fired per element).

    Extra keywords (not in the reference): seed (reproducible bundles), max_passes (stop even if not converged),
    verbose.
    """
    def __init__(self, apertureRadius, frustaRadii, frustaDepths, coneDepth, el_FRUs, el_CON, num_rays=10000, precision=0.01,
                 seed=None, max_passes=None, verbose=False):
        RTVF.__init__(self, num_rays, precision)
        self.apertureRadius = apertureRadius
        self.frustaRadii = frustaRadii
        self.frustaDepths = frustaDepths
        self.coneDepth = coneDepth
        if type(el_FRUs) == int:
            el_FRUs = N.asarray([el_FRUs])
        if type(el_CON) == int:
            el_CON = N.asarray([el_CON])
        self.el_FRUs = el_FRUs
        self.el_CON = el_CON
        self.t0 = time.time()

        n_sections = len(frustaRadii)
        n_el_sec = [int(k) for k in N.ravel(el_FRUs)]
        n_el_con = int(N.ravel(el_CON)[0])
        n = 1 + sum(n_el_sec) + n_el_con
        self._init_statistics(n)

        # the profile: (radius, height) at the section ends
        r_end = N.hstack([apertureRadius, N.asarray(frustaRadii, dtype=float)])
        z_end = N.add.accumulate(N.hstack([0., N.asarray(frustaDepths, dtype=float)]))
        max_depth = z_end[-1]
        R_back = r_end[-1]

        # -- elements: emitter description, receiving bin, area ------------------------------------------------------
        emitters = [dict(kind='aperture')]
        bins = [dict(surf=0, mode=0, rng=[0.] * 6)]
        areas = [N.pi * apertureRadius ** 2.]
        for k in range(n_sections):
            m = n_el_sec[k]
            dr, dz = (r_end[k + 1] - r_end[k]) / m, (z_end[k + 1] - z_end[k]) / m
            slant = N.sqrt(dz ** 2 + dr ** 2)
            for e in range(m):
                ra, rb = r_end[k] + e * dr, r_end[k] + (e + 1) * dr
                areas.append(N.pi * (ra + rb) * slant)
                emitters.append(dict(kind='wall', center=z_end[k] + e * dz, r0=ra, r1=rb, depth=dz,
                                     rays_in=not (k > 0 and frustaDepths[k] < 0.)))
                # receiving bin (:620-647): heights / radii at the element ends from the section ends, as the reference
                # computes them, both ranges closed, hits rounded to 9 decimals
                h = sorted([z_end[k] + e * (z_end[k + 1] - z_end[k]) / el_FRUs[k], z_end[k] + (e + 1) * (z_end[k + 1] - z_end[k]) / el_FRUs[k]])
                r = sorted([r_end[k] + e * (r_end[k + 1] - r_end[k]) / el_FRUs[k], r_end[k] + (e + 1) * (r_end[k + 1] - r_end[k]) / el_FRUs[k]])
                bins.append(dict(surf=1 + k, mode=_cabi.BIN_HEIGHT | _cabi.BIN_RADIUS | _cabi.BIN_ROUND9,
                                 rng=[0., 0., h[0], h[1], r[0], r[1]]))
        for e in range(n_el_con):
            ra, rb = R_back + e * (-R_back) / n_el_con, R_back + (e + 1) * (-R_back) / n_el_con
            areas.append(N.pi * (ra + rb) * N.sqrt(coneDepth ** 2 + R_back ** 2) / n_el_con)
            emitters.append(dict(kind='wall', center=max_depth + coneDepth * e / n_el_con, r0=ra, r1=rb,
                                 depth=coneDepth / n_el_con, rays_in=not coneDepth < 0.))
            # cone elements (:662-672): radius only, r2 <= r < r1, not rounded
            bins.append(dict(surf=1 + n_sections, mode=_cabi.BIN_RADIUS | _cabi.BIN_RADIUS_HALF_OPEN,
                             rng=[0., 0., 0., 0., R_back - (e + 1) * R_back / float(n_el_con), R_back - e * R_back / float(n_el_con)]))
        self.areas = N.array(areas)
        self.ray_counts = N.ones(n) * int(self.num_rays)

        # -- the scene: black receivers everywhere (:430-482) ---------------------------------------------------------
        def black():
            return opt.LambertianReceiver(absorptivity=1.)
        objects = [AssembledObject(surfs=[Surface(RoundPlateGM(Re=apertureRadius), black())], transform=None)]
        for k in range(n_sections):
            ra, rb, depth, za = r_end[k], r_end[k + 1], frustaDepths[k], z_end[k]
            if ra == rb:
                geom, frame = FiniteCylinder(diameter=2. * rb, height=depth), translate(z=za + depth / 2.)
            elif depth < 0.:
                geom, frame = ConicalFrustum(z1=0., r1=ra, z2=-depth, r2=rb), N.dot(translate(z=za), rotx(N.pi))
            elif depth > 0.:
                geom, frame = ConicalFrustum(z1=0., r1=ra, z2=depth, r2=rb), translate(z=za)
            else:
                geom, frame = RoundPlateGM(Re=ra, Ri=rb), translate(z=za)
            objects.append(AssembledObject(surfs=[Surface(geom, black())], transform=frame))
        if coneDepth > 0.:
            geom, frame = FiniteCone(r=R_back, h=coneDepth), N.dot(rotx(N.pi), translate(z=-(max_depth + coneDepth)))
        elif coneDepth == 0.:
            geom, frame = RoundPlateGM(Re=R_back), translate(z=max_depth)
        else:
            geom, frame = FiniteCone(r=R_back, h=-coneDepth), translate(z=max_depth + coneDepth)
        objects.append(AssembledObject(surfs=[Surface(geom, black())], transform=frame))
        self.A = Assembly(objects=objects)
        self.AP, self.FRU, self.CON = objects[0], objects[1:1 + n_sections], objects[1 + n_sections:]

        self._bins = (N.array([b['surf'] for b in bins], dtype=N.int32), N.array([b['mode'] for b in bins], dtype=N.int32),
                      N.array([b['rng'] for b in bins], dtype=float))
        self._emitters = emitters
        self._seed = seed
        self.engine = TracerEngine(self.A)
        self.itmax = 1          # one interaction: emitted rays are absorbed where they land
        self.minener = 1e-10

        # -- passes until every entry has met the criteria in two passes (:494-566) ------------------------------------
        stable_passes, passes = 0, 0
        while (self.progress.any() or stable_passes < 2) and (max_passes is None or passes < max_passes):
            t_pass = time.time()
            for i in range(n):
                if self.ray_counts[i] != 0.:
                    self._trace_element(i, passes)
                    self.alloc_VF(i)
            self.p += self.ray_counts
            self.test_precision()
            passes += 1
            if verbose:
                print('		Progress:', N.sum(self.progress), '/', self.progress.size, '; Pass duration:', time.time() - t_pass, 's')
            if N.sum(self.progress) == 0:
                stable_passes += 1
        self.passes = passes
        if verbose:
            print('	VF calculation time:', time.time() - self.t0, 's')

    def _trace_element(self, i, pass_no):
        em = self._emitters[i]
        num_rays = int(self.ray_counts[i])
        seed = None if self._seed is None else (self._seed + 7919 * pass_no + i)
        if em['kind'] == 'aperture':
            src = disk_bundle(num_rays, center=N.vstack([0, 0, 0]), direction=N.array([0, 0, 1]), radius=self.apertureRadius,
                              ang_range=N.pi / 2., flux=1. / (N.pi * self.apertureRadius ** 2.), seed=seed)
        else:
            src = self.gen_source(num_rays, em['r0'], em['r1'], em['depth'], N.vstack([0, 0, em['center']]), em['rays_in'], seed=seed)
        self.engine.ray_tracer(src, reps=self.itmax, min_energy=self.minener, tree=False, seed=seed, feed=False)

    def gen_source(self, num_rays, r0, r1, depth, center, rays_in, procs=1, seed=None):
        """the Lambertian emitter of one wall element (:585-596)"""
        if r0 == r1:
            # `center` is the base of the element, as for the frusta; sources.vf_cylinder_bundle wants the mid-height
            # (sources.py:722 -- the reference still passes the base, :587, and so emits half an element too low)
            mid = N.asarray(center, dtype=float) + N.vstack([0., 0., depth / 2.])
            return vf_cylinder_bundle(num_rays=num_rays, rc=r0, lc=abs(depth), center=mid, direction=N.array([0, 0, 1]), rays_in=rays_in, seed=seed)
        if depth == 0.:
            return disk_bundle(num_rays=num_rays, center=center, direction=N.array([0, 0, N.sign(r1 - r0)]), radius=r0,
                               ang_range=N.pi / 2., radius_in=r1, seed=seed)
        return vf_frustum_bundle(num_rays=num_rays, r0=r0, r1=r1, depth=depth, center=center, direction=N.array([0, 0, 1]),
                                 rays_in=rays_in, seed=seed)

    def alloc_VF(self, n):
        """row n of the pass matrix: absorbed energy per receiving element, binned on the device (:598-674)"""
        surf, mode, rng = self._bins
        self.VF[n, :] = self.engine.bin_hits(surf, surf, rng, mode)
        self.reset_opt()


class Four_parameters_cavity_RTVF(Two_N_parameters_cavity_RTVF):
    """
    One frustum and a cone ("Open cavity receiver geometry influence on radiative losses", DOI:10.13140/2.1.3845.5048;
    reference :677-684).
    """
    def __init__(self, apertureRadius, apertureDepth, coneRadius, coneDepth, el_FRU, el_CON, num_rays, precision, **kw):
        Two_N_parameters_cavity_RTVF.__init__(self, apertureRadius, [coneRadius], [apertureDepth], coneDepth, el_FRU, el_CON,
                                              num_rays, precision, **kw)
