"""
Pins the oracle (oracle/, the CPU restatement) against fixtures produced by the REAL reference
(tests/golden/make_golden.py).  CPU only.
"""
import json
import os

import numpy as N
import pytest

from oracle import geometry, optics, sources, engine, kinds
from helpers import load, case_names, oracle_scene, source_dict, GOLDEN

TOL = dict(rtol=1e-9, atol=1e-9)


def test_geometry_all_kinds():
    g = load('geometry.npz')
    names = case_names(g)
    seen = set()
    for ci in range(int(g['n_cases'])):
        pre = 'g%d_' % ci
        kind = int(g[pre + 'kind'])
        seen.add(kind)
        t = geometry.intersect(kind, g[pre + 'frame'], list(g[pre + 'gm']), g[pre + 'extra'], g[pre + 'v'], g[pre + 'd'])
        t_ref = g[pre + 't']
        assert N.array_equal(N.isfinite(t), N.isfinite(t_ref)), names[ci]
        idx = g[pre + 'hit_idx']
        assert N.allclose(t[idx], t_ref[idx], **TOL), names[ci]
        if len(idx):
            pts = g[pre + 'v'][:, idx] + t[idx] * g[pre + 'd'][:, idx]
            assert N.allclose(pts, g[pre + 'hits'], **TOL), names[ci]
            nrm = geometry.normals(kind, g[pre + 'frame'], list(g[pre + 'gm']), g[pre + 'hits'], g[pre + 'd'][:, idx])
            ok = N.all(N.isclose(nrm, g[pre + 'normals'], **TOL) | (N.isnan(nrm) & N.isnan(g[pre + 'normals'])), axis=0)
            assert ok.all(), (names[ci], N.nonzero(~ok)[0][:5])
    assert seen == set(range(30)), "every native geometry kind has a fixture"
    assert sum(int(N.isfinite(g['g%d_t' % ci]).sum()) for ci in range(int(g['n_cases']))) > 5000


def _optics_case(o, name):
    names = case_names(o)
    i = names.index(name)
    return 'o%d_' % i


def test_optics_variate_replay():
    o = load('optics.npz')
    frame, nrm, d, pts, e, wl = o['frame'], o['normals'], o['dirs'], o['points'], o['energy'], o['wavelengths']
    H = d.shape[1]
    up = frame[:3, 2]

    def check(pre, dirs, energy, parents=None, ref=None):
        assert N.allclose(dirs, o[pre + 'out_dirs'], **TOL), pre
        assert N.allclose(energy, o[pre + 'out_energy'], **TOL), pre
        if parents is not None:
            assert N.array_equal(parents, o[pre + 'out_parents']), pre
        if ref is not None:
            assert N.allclose(ref, o[pre + 'out_ref'], **TOL), pre

    # deterministic kinds go through shade() itself
    rid = N.arange(H, dtype=N.uint64)
    for name in ('transparent', 'reflective', 'one_sided_reflective', 'real_reflective_sigma0', 'reflective_spectral',
                 'refractive_split', 'fresnel_conductor', 'periodic_boundary'):
        pre = _optics_case(o, name)
        blocks = optics.shade(int(o[pre + 'kind']), list(o[pre + 'opt']), o[pre + 'extra'], up, d, e, o[pre + 'ref_in'], wl, nrm, 1, rid, 1)
        dirs = N.hstack([b['directions'] for b in blocks])
        en = N.hstack([b['energy'] for b in blocks])
        par = N.hstack([b['sel'] for b in blocks])
        ref = N.hstack([b['ref'] for b in blocks]) if (pre + 'out_ref') in o.files else None
        check(pre, dirs, en, par, ref)
        # outgoing rays start at the hit points -- but for a periodic boundary, whose second block starts one period along the normal
        start = N.hstack([pts[:, b['sel']] + (b['shift'][None, :] * nrm[:, b['sel']] if 'shift' in b else 0.) for b in blocks])
        assert N.allclose(o[pre + 'out_vertices'], start, **TOL)
        if name == 'periodic_boundary':         # (optics_callables.py:703-723: stubs of energy 0, then the rays themselves, moved)
            assert len(blocks) == 2 and (en[:H] == 0).all() and N.array_equal(en[H:], e) and N.array_equal(dirs[:, H:], d)
            assert N.allclose(o[pre + 'out_vertices'][:, H:], pts + 0.7 * nrm, **TOL)
    # slope error: replay numpy's normal / uniform draws
    for name, bi, onesided in (('real_reflective_bivar', True, False), ('real_reflective_radial', False, False),
                               ('one_sided_real_reflective', True, True), ('real_reflective_iam', True, False)):
        pre = _optics_case(o, name)
        absorb, sigma = o[pre + 'opt'][0], o[pre + 'opt'][1]
        g0 = o[pre + 'draw_g0'] / sigma
        g1 = o[pre + 'draw_g1'] / sigma if bi else N.zeros(H)
        u = o[pre + 'draw_phi'] / (2. * N.pi) if not bi else N.zeros(H)
        real = optics.slope_error_normals(nrm, sigma, bi, g0, g1, u)
        en = e * (1. - absorb) * optics.iam(list(o[pre + 'opt']), 3, 4, d, nrm)
        if onesided:
            en = en.copy()
            en[N.sum(d * up[:, None], axis=0) > 0] = 0
        check(pre, optics.reflections(d, real), en, N.arange(H))
    for name in ('lambertian', 'lambertian_narrow', 'lambertian_iam', 'lambertian_iam_c2'):
        pre = _optics_case(o, name)
        dirs = optics.lambertian_directions(nrm, o[pre + 'draw_xi1'], o[pre + 'draw_xi2'], o[pre + 'opt'][1])
        check(pre, dirs, e * (1. - o[pre + 'opt'][0]) * optics.iam(list(o[pre + 'opt']), 4, 5, d, nrm), N.arange(H))
        if 'iam' in name:
            f = optics.iam(list(o[pre + 'opt']), 4, 5, d, nrm)
            assert f.min() < 0.5 and f.max() > 0.95
    # attenuating media (Absorbant.attenuate): energies are deterministic, the Lambertian wall replays its direction draws
    for name in ('lambertian_absorbant', 'lambertian_absorbant_scaled'):
        pre = _optics_case(o, name)
        blocks = optics.shade(int(o[pre + 'kind']), list(o[pre + 'opt']), o[pre + 'extra'], up, d, e, o[pre + 'ref_in'], wl, nrm, 1, rid, 1,
                              path=o[pre + 'path'])
        dirs = optics.lambertian_directions(nrm, o[pre + 'draw_xi1'], o[pre + 'draw_xi2'], o[pre + 'opt'][1])
        check(pre, dirs, blocks[0]['energy'], N.arange(H))
        assert (blocks[0]['energy'] < e * (1. - o[pre + 'opt'][0]) * 0.95).all() and o[pre + 'path'].min() < 0.5 < 2.5 < o[pre + 'path'].max()
    for name in ('refractive_transmissive_split', 'refractive_transmissive_one_coefficient'):
        pre = _optics_case(o, name)
        blocks = optics.shade(int(o[pre + 'kind']), list(o[pre + 'opt']), o[pre + 'extra'], up, d, e, o[pre + 'ref_in'], wl, nrm, 1, rid, 1,
                              path=o[pre + 'path'])
        check(pre, N.hstack([b['directions'] for b in blocks]), N.hstack([b['energy'] for b in blocks]), N.hstack([b['sel'] for b in blocks]),
              N.hstack([b['ref'] for b in blocks]))
    # angle-dependent absorptance: energies are deterministic, directions replay the Lambertian draws
    for name in ('lambertian_directional', 'lambertian_directional_spectral'):
        pre = _optics_case(o, name)
        blocks = optics.shade(int(o[pre + 'kind']), list(o[pre + 'opt']), o[pre + 'extra'], up, d, e, o[pre + 'ref_in'], wl, nrm, 1, rid, 1)
        dirs = optics.lambertian_directions(nrm, o[pre + 'draw_xi1'], o[pre + 'draw_xi2'], N.pi / 2.)
        check(pre, dirs, blocks[0]['energy'], N.arange(H))
    # specular with a constant / an angle-dependent probability on top of the angle-dependent absorptance: replay the draws
    for name in ('lambertian_specular_directional', 'lambertian_piecewise_specular_directional'):
        pre = _optics_case(o, name)
        opt_p, ex = list(o[pre + 'opt']), o[pre + 'extra']
        th_in = N.arccos(N.sqrt(N.sum((N.sum(d * nrm, axis=0) * nrm) ** 2, axis=0)))
        k = len(ex) // (3 if int(opt_p[0]) == 2 else 2)
        prob = opt_p[1] if int(opt_p[0]) == 1 else N.interp(th_in, ex[:k], ex[2 * k:])
        spec = o[pre + 'draw_u'] < prob
        dirs = N.zeros((3, H))
        dirs[:, spec] = optics.reflections(d[:, spec], nrm[:, spec])
        dirs[:, ~spec] = optics.lambertian_directions(nrm[:, ~spec], o[pre + 'draw_xi1'], o[pre + 'draw_xi2'], N.pi / 2.)
        check(pre, dirs, e * (1. - N.interp(th_in, ex[:k], ex[k:2 * k])), N.arange(H))
        assert spec.any() and (~spec).any()
        # and shade() itself draws the same decision from its own uniforms: energies are deterministic
        blocks = optics.shade(int(o[pre + 'kind']), opt_p, ex, up, d, e, o[pre + 'ref_in'], wl, nrm, 1, rid, 1)
        assert N.allclose(blocks[0]['energy'], o[pre + 'out_energy'], **TOL)
    for name in ('lambertian_specular', 'lambertian_specular_iam'):      # (the _IAM class of the reference absorbs nothing, :607-609)
        pre = _optics_case(o, name)
        spec = o[pre + 'draw_u'] < o[pre + 'opt'][1]
        dirs = N.zeros((3, H))
        dirs[:, spec] = optics.reflections(d[:, spec], nrm[:, spec])
        dirs[:, ~spec] = optics.lambertian_directions(nrm[:, ~spec], o[pre + 'draw_xi1'], o[pre + 'draw_xi2'], N.pi / 2.)
        check(pre, dirs, e * (1. - o[pre + 'opt'][0]), N.arange(H))
    # single-ray refraction: replay the reflect-or-refract draw
    pre = _optics_case(o, 'refractive_single')
    n1 = o[pre + 'ref_in']
    n2 = N.where(n1 == o[pre + 'opt'][0], o[pre + 'opt'][1], o[pre + 'opt'][0])
    refr, out_dirs = optics.refractions(n1, n2, d, nrm)
    R = N.ones(H)
    R[refr] = optics.fresnel(d[:, refr], nrm[:, refr], n1[refr], n2[refr])
    refl = o[pre + 'draw_u'] <= R
    dr = N.zeros((3, H))
    dr[:, refr] = out_dirs
    dirs = N.hstack((optics.reflections(d, nrm)[:, refl], dr[:, ~refl]))
    par = N.hstack((N.nonzero(refl)[0], N.nonzero(~refl)[0]))
    check(pre, dirs, e[par], par)
    assert refl.any() and (~refl).any() and (~refr).any(), "fixture exercises reflection, refraction and TIR"
    # refraction with perturbed normals: replay theta, phi
    pre = _optics_case(o, 'refractive_split_sigma')
    sigma = o[pre + 'opt'][3]
    th, phi = o[pre + 'draw_g0'], o[pre + 'draw_phi']
    err = N.vstack((N.sin(th) * N.cos(phi), N.sin(th) * N.sin(phi), N.cos(th)))
    rots = optics.rotation_to_z(nrm.T)
    pn = N.array([N.dot(rots[i], err[:, i]) for i in range(H)]).T
    n1 = o[pre + 'ref_in']
    n2 = N.where(n1 == o[pre + 'opt'][0], o[pre + 'opt'][1], o[pre + 'opt'][0])
    refr, out_dirs = optics.refractions(n1, n2, d, pn)
    R = N.ones(H)
    R[refr] = optics.fresnel(d[:, refr], pn[:, refr], n1[refr], n2[refr])
    dirs = N.hstack((optics.reflections(d, pn), out_dirs))
    en = N.hstack((e * R, e[refr] * (1. - R[refr])))
    check(pre, dirs, en, N.hstack((N.arange(H), N.nonzero(refr)[0])))
    assert sigma > 0


def test_optics_carried_columns_replay():
    """SURVEY 8(f)2, rest: refraction between materials of complex, wavelength-dependent index with attenuation from Im m
    (Refractive, RefractiveAbsorbant), and polychromatic bundles (the wall that integrates spectra, the classes that scale them) --
    the oracle against the reference's own outputs."""
    o = load('optics.npz')
    frame, nrm, d, pts, e = o['frame'], o['normals'], o['dirs'], o['points'], o['energy']
    H = d.shape[1]
    up = frame[:3, 2]
    rid = N.arange(H, dtype=N.uint64)
    wl1 = o['wavelengths']

    def blocks_of(pre, ext, wl=None, path=None):
        return optics.shade(int(o[pre + 'kind']), list(o[pre + 'opt']), o[pre + 'extra'], up, d, e, o[pre + 'ref_in'],
                            wl1 if wl is None else wl, nrm, 1, rid, 1, path=o[pre + 'path'] if path is None else path, ext=ext)

    def cat(blocks, key):
        return N.hstack([b[key] for b in blocks])

    # deterministic material cases: everything, the complex indices of the outgoing rays included
    for name in ('material_split', 'material_absorbant_split', 'material_absorbant_scaled'):
        pre = _optics_case(o, name)
        b = blocks_of(pre, dict(mat=o[pre + 'mat']))
        assert N.array_equal(cat(b, 'sel'), o[pre + 'out_parents']), name
        assert N.allclose(cat(b, 'directions'), o[pre + 'out_dirs'], **TOL), name
        assert N.allclose(cat(b, 'energy'), o[pre + 'out_energy'], rtol=1e-9, atol=1e-12), name
        assert N.iscomplexobj(o[pre + 'out_ref']) and N.allclose(cat(b, 'ref'), o[pre + 'out_ref'], rtol=1e-12, atol=0), name
    pre, pre_a = _optics_case(o, 'material_split'), _optics_case(o, 'material_absorbant_split')
    ratio = o[pre_a + 'out_energy'] / o[pre + 'out_energy']
    assert ratio.max() <= 1. and ratio.min() < 0.5 and (ratio == 1.).any(), "the fixture attenuates in glass, not in air"
    # a ray in neither medium enters material_1
    assert o[pre + 'ref_in'][5] == 1.2 and N.allclose(o[pre + 'out_ref'][H:][N.nonzero(o[pre + 'out_parents'][H:] == 5)[0]], o[pre + 'mat'][0][5])
    # one ray per hit: replay the reflect-or-refract draw
    pre = _optics_case(o, 'material_single')
    m1, mat = o[pre + 'ref_in'], o[pre + 'mat']
    m2 = N.where(m1 == mat[0], mat[1], mat[0])
    refr, out_dirs = optics.refractions(m1.real, m2.real, d, nrm)
    R = N.ones(H)
    with N.errstate(all='ignore'):
        R[refr] = N.real(optics.fresnel(d[:, refr], nrm[:, refr], m1[refr], m2[refr]))
    refl = o[pre + 'draw_u'] <= R
    dr = N.zeros((3, H))
    dr[:, refr] = out_dirs
    assert N.allclose(N.hstack((optics.reflections(d, nrm)[:, refl], dr[:, ~refl])), o[pre + 'out_dirs'], **TOL)
    assert N.array_equal(N.hstack((N.nonzero(refl)[0], N.nonzero(~refl)[0])), o[pre + 'out_parents'])
    assert N.allclose(N.hstack((m1[refl], m2[~refl])), o[pre + 'out_ref'], rtol=1e-12, atol=0)
    assert refl.any() and (~refl).any() and (~refr).any()
    # perturbed normals: replay theta, phi
    pre = _optics_case(o, 'material_split_sigma')
    th, phi = o[pre + 'draw_g0'], o[pre + 'draw_phi']
    err = N.vstack((N.sin(th) * N.cos(phi), N.sin(th) * N.sin(phi), N.cos(th)))
    rots = optics.rotation_to_z(nrm.T)
    pn = N.array([N.dot(rots[i], err[:, i]) for i in range(H)]).T
    m1, mat = o[pre + 'ref_in'], o[pre + 'mat']
    m2 = N.where(m1 == mat[0], mat[1], mat[0])
    refr, out_dirs = optics.refractions(m1.real, m2.real, d, pn)
    R = N.ones(H)
    with N.errstate(all='ignore'):
        R[refr] = N.real(optics.fresnel(d[:, refr], pn[:, refr], m1[refr], m2[refr]))
    assert N.allclose(N.hstack((optics.reflections(d, pn), out_dirs)), o[pre + 'out_dirs'], **TOL)
    assert N.allclose(N.hstack((e * R, e[refr] * (1. - R[refr]))), o[pre + 'out_energy'], **TOL)

    # polychromatic bundles
    pre = _optics_case(o, 'polychromatic_wall')
    b = blocks_of(pre, dict(spec=o[pre + 'spec_in'], swl=o[pre + 'spec_wl']), wl=N.zeros(H))
    assert N.allclose(b[0]['spectra'], o[pre + 'out_spectra'], rtol=1e-12, atol=0)
    assert N.allclose(b[0]['energy'], o[pre + 'out_energy'], rtol=1e-12, atol=0)
    assert N.allclose(optics.lambertian_directions(nrm, o[pre + 'draw_xi1'], o[pre + 'draw_xi2'], N.pi / 2.), o[pre + 'out_dirs'], **TOL)
    assert (o[pre + 'out_spectra'] < o[pre + 'spec_in']).all()
    scaled = 0
    for name in case_names(o):
        if not name.startswith('poly_'):
            continue
        pre = _optics_case(o, name)
        b = blocks_of(pre, dict(spec=o[pre + 'spec_in'], swl=o[pre + 'spec_wl']), wl=N.zeros(H))
        if name == 'poly_lambertian_specular':      # which rays are mirrored is drawn; the spectrum is handed on unchanged either way
            assert N.array_equal(o[pre + 'out_spectra'], o[pre + 'spec_in'])
            assert N.allclose(cat(b, 'spectra'), o[pre + 'out_spectra'], rtol=1e-12, atol=0)
            continue
        assert N.array_equal(cat(b, 'sel'), o[pre + 'out_parents']), name
        assert N.allclose(cat(b, 'spectra'), o[pre + 'out_spectra'], rtol=1e-12, atol=0), name
        assert N.allclose(cat(b, 'energy'), o[pre + 'out_energy'], rtol=1e-9, atol=1e-12), name
        scaled += int(not N.array_equal(o[pre + 'out_spectra'], o[pre + 'spec_in'][:, o[pre + 'out_parents']]))
    assert scaled == 6, "Reflective, OneSidedReflective, RealReflective, Lambertian and the directional wall scale the spectrum; " \
                        "PeriodicBoundary cancels its stub's"
    pre = _optics_case(o, 'poly_periodic_boundary')
    assert (o[pre + 'out_spectra'][:, :H] == 0).all() and N.array_equal(o[pre + 'out_spectra'][:, H:], o[pre + 'spec_in'])


def test_fresnel_to_attenuating_grid():
    o = load('optics.npz')
    with N.errstate(all='ignore'):
        rp, rs, t2 = optics.fresnel_to_attenuating(float(o['fta_n1']), o['fta_m_re'] + 1j * o['fta_m_im'], o['fta_theta1'])
    assert N.allclose(rp, o['fta_rp'], rtol=1e-12) and N.allclose(rs, o['fta_rs'], rtol=1e-12)
    assert N.allclose(t2, o['fta_theta2'], rtol=1e-12)


def test_sources_variate_replay():
    s = load('sources.npz')
    names = case_names(s)
    for i, name in enumerate(names):
        pre = 's%d_' % i
        src = source_dict(s, pre)
        u = [s[pre + 'u%d' % k] for k in range(4)]
        n = len(u[0])
        v, d, e, rid = sources.generate(src, n, 0, 0, uniforms=u)
        assert N.allclose(v, s[pre + 'vertices'], rtol=1e-10, atol=1e-9), name
        assert N.allclose(d, s[pre + 'directions'], rtol=1e-9, atol=1e-11), name
        assert N.allclose(e, s[pre + 'energy'], rtol=1e-12), name


def test_buie_table_restatement_matches_packed_table():
    """oracle.sources.buie_tables (straight restatement) == the packed table the host ships to the device"""
    s = load('sources.npz')
    names = case_names(s)
    for name, csr, pre_csr in (('buie_csr0.01_raw', 0.01, False), ('buie_csr0.05', 0.05, True), ('buie_csr0.3', 0.3, True),
                               ('buie_csr0', 0., True)):
        pre = 's%d_' % names.index(name)
        packed = sources.table_from_desc_buie(s[pre + 'desc_buie'])
        tab = sources.buie_tables(csr, pre_csr)
        R = N.linspace(0., 1., 20001)[:-1]
        assert N.allclose(sources.buie_thetas(R, tab), sources.buie_thetas_packed(R, packed), rtol=1e-12, atol=1e-15), name


def _check_tree(res, g, pre, name):
    nlev = int(g[pre + 'n_levels'])
    assert len(res['levels']) == nlev, (name, [l['vertices'].shape[1] for l in res['levels']])
    for k in range(1, nlev):
        L = res['levels'][k]
        assert L['vertices'].shape == g[pre + 'L%d_vertices' % k].shape, (name, k)
        assert N.array_equal(L['parents'], g[pre + 'L%d_parents' % k]), (name, k)
        assert N.allclose(L['vertices'], g[pre + 'L%d_vertices' % k], rtol=1e-9, atol=1e-8), (name, k)
        assert N.allclose(L['directions'], g[pre + 'L%d_directions' % k], rtol=1e-9, atol=1e-9), (name, k)
        assert N.allclose(L['energy'], g[pre + 'L%d_energy' % k], rtol=1e-9, atol=1e-12), (name, k)


def test_engine_deterministic_scenes():
    g = load('engine.npz')
    for i, name in enumerate(case_names(g)):
        pre = 'e%d_' % i
        scene = oracle_scene(g, pre)
        v, d, e = g[pre + 'v'], g[pre + 'd'], g[pre + 'e']
        n = v.shape[1]
        ref = g[pre + 'ref_index'] if (pre + 'ref_index') in g.files else N.ones(n)
        res = engine.trace(scene, v, d, e, ref, N.zeros(n), N.arange(n, dtype=N.uint64), int(g[pre + 'reps']),
                           float(g[pre + 'min_energy']), 1)
        _check_tree(res, g, pre, name)
        assert res['last_vertices'].shape == g[pre + 'last_vertices'].shape, name
        assert N.allclose(res['last_vertices'], g[pre + 'last_vertices'], rtol=1e-9, atol=1e-8), name
        assert N.allclose(res['last_directions'], g[pre + 'last_directions'], rtol=1e-9, atol=1e-9), name


def test_kdtree_build_matches_reference():
    from tracer_amd import scenes
    from tracer_amd.accel_tree import KdTree
    g = load('kdtree_nsttf.npz')
    plant, field, rec, src = scenes.nsttf_field(sigma=0.)
    S = len(plant.get_surfaces())
    for tag, fast in (('', False), ('fast_', True)):
        kd = KdTree(plant, 8 + 1.3 * N.log(S), fast=fast, min_leaf=1)
        f = kd.flat()
        assert N.array_equal(f['flag'], g[tag + 'flag'])
        assert N.array_equal(f['child'], g[tag + 'child'])
        assert N.array_equal(f['split'], g[tag + 'split'])          # bit-exact: same planes
        assert N.array_equal(f['leaf_cnt'], g[tag + 'leaf_cnt'])
        assert N.array_equal(f['leaf_surfs'], g[tag + 'leaf_surfs'])
        assert N.array_equal(f['always_relevant'], g[tag + 'always_relevant'])
        assert N.array_equal(f['bounds'], N.concatenate((g[tag + 'minpoint'], g[tag + 'maxpoint'])))


def _kd_fixture_tree(g):
    tree = dict((k, g[k]) for k in ('flag', 'split', 'child', 'leaf_off', 'leaf_cnt', 'leaf_surfs', 'always_relevant'))
    tree['bounds'] = N.concatenate((g['minpoint'], g['maxpoint']))
    return tree


def test_kdtree_traversal_matches_reference():
    """KdTree.traversal restated (oracle/accel.py) == the relevancy matrix the reference returns: 700 rays on the NSTTF tree --
    from the sun's side, from the receiver, from inside the root box, along the axes (infinite inverse directions), missing the box"""
    from oracle import accel
    g = load('kdtree_nsttf.npz')
    S = int(g['trav_n_surf'])
    v, d = g['trav_vertices'], g['trav_directions']
    expected = N.unpackbits(g['trav_relevancy_bits'], axis=1)[:, :v.shape[1]].astype(bool)
    any_inter, rel = accel.traversal(_kd_fixture_tree(g), S, v, d)
    assert bool(any_inter) == bool(g['trav_any'])
    assert N.array_equal(rel, expected)
    crossed = expected[:-1].sum(axis=0)
    assert (crossed == 0).sum() > 200 and crossed.max() > 20 and expected[-1].all()     # misses, long walks, the receiver always
    assert (d == 0.).any()


def test_accountant_name_table():
    import tracer_amd.optics_callables as oc
    with open(os.path.join(GOLDEN, 'accountant_names.json')) as f:
        ref = json.load(f)
    oc.__getattr__('ReflectiveReceiver')     # builds the table
    mine = dict((k, [a.__name__ for a in v]) for k, v in oc._SUFFIXES.items())
    assert set(mine) == set(ref)
    for k in ref:
        assert mine[k] == ref[k], k
    assert [type(a).__name__ for a in oc.OneSidedReflectiveReceiver(1.).accountants] == ['AbsorptionAccountant', 'LocationAccountant']


def test_scattering_maps_vs_reference():
    """
    Participating media: the two pure functions under the reference's scattering optics, with numpy's draws
    replayed -- Henyey_Greenstein.sample (sampling.py:150-168) and optics.scattering (optics.py:214-239).  The oracle's
    restatements must give the reference's angles, free paths and scattered / not scattered decisions.
    """
    from oracle import optics
    g = load('scattering.npz')
    for k, gv in enumerate(g['hg_g']):
        th = optics.hg_theta(gv, g['hg%d_R' % k])
        assert N.allclose(th, g['hg%d_theta' % k], rtol=0., atol=1e-12), gv
        assert N.allclose(2. * N.pi * g['hg%d_U' % k], g['hg%d_phi' % k], rtol=0., atol=1e-14)
    scat, lengths = optics.scattering(g['sc_sigma'], g['sc_paths'], g['sc_R'])
    assert N.array_equal(scat, g['sc_scattered']) and 500 < scat.sum() < 3000
    assert N.allclose(lengths, g['sc_lengths'], rtol=1e-14, atol=0.)


def test_minidish_example_scene_vs_reference_runs():
    """
    The oracle on the scene of examples/test_case.py (tilted dish of models/tau_minidish.py, homogenizer duct, one-sided receiver),
    2e5 rays, against ten runs of the reference itself (mc_minidish.npz, make_golden.py --mc-minidish): power on the plate and on
    each duct wall, the size of every level of the ray tree.
    """
    import math
    from tracer_amd.models.tau_minidish import MiniDish
    from tracer_amd.sources import solar_disk_bundle
    from tracer_amd.spatial_geometry import rotx
    from tracer_amd.scene import compile_scene
    from oracle import engine
    mc = load('mc_minidish.npz')
    n = 200000
    x = -1 / math.sqrt(2)
    dish = MiniDish(5., 6.25, 0.9, 6.95, 0.4, 0.7, 0.9)
    dish.set_transform(rotx(-N.pi / 4))
    cs = compile_scene(dish)
    sun = solar_disk_bundle(n, N.c_[[0, 7., 7.]], N.array([0, x, x]), 3., 0.005, flux=1000., seed=43)
    with N.errstate(all='ignore'):
        out = engine.trace_from_compiled(cs, sun.source_args(), 100, 1e-6)
    surfs = dish.get_surfaces()
    plate = surfs.index(dish.get_receiver_surf().get_surfaces()[0])
    walls = [surfs.index(w) for w in dish.get_homogenizer().get_surfaces()]
    grow = math.sqrt(1. + 10 * float(mc['rays_per_run']) / n)            # this run's own Monte-Carlo error on top of the reference's
    got = out['absorbed']
    assert abs(got[plate] - float(mc['receiver_mean'])) <= 4. * float(mc['receiver_se']) * grow, (got[plate], float(mc['receiver_mean']))
    assert N.all(N.abs(got[walls] - mc['walls_mean']) <= 4. * mc['walls_se'] * grow), (got[walls], mc['walls_mean'])
    sizes = N.array([lv['energy'].shape[0] for lv in out['levels']][:5], dtype=float)          # levels[0] is the source bundle
    frac, frac_ref = sizes / n, mc['levels_mean'] / float(mc['rays_per_run'])
    frac_se = N.sqrt((mc['levels_se'] / float(mc['rays_per_run'])) ** 2 + frac * (1 - frac) / n)
    assert N.all(N.abs(frac - frac_ref) <= 4. * frac_se + 1e-12), (frac, frac_ref)


def test_plates_example_scene_vs_reference_runs():
    """
    The oracle on the scene of examples/accel_tree_example.py (1002 Lambertian surfaces, a dozen diffuse bounces), 3e4 rays, against
    eight runs of the reference itself (mc_plates.npz, make_golden.py --mc-plates): power absorbed in total, by the slab, by each layer.
    """
    import math
    from helpers import plates_scene, plates_source
    from tracer_amd.scene import compile_scene
    from oracle import engine
    mc = load('mc_plates.npz')
    asm, layers, side = plates_scene()
    n = 30000
    with N.errstate(all='ignore'):
        # (min_energy is an energy per ray: the example's 0.05 W on 2e4 rays, kept in proportion)
        out = engine.trace_from_compiled(compile_scene(asm), plates_source(n, layers, side, 53).source_args(), 1000,
                                         0.05 * float(mc['rays_per_run']) / n)
    per = out['absorbed']
    got = N.r_[per.sum(), per[1], per[2:].reshape(10, 100).sum(axis=1)]
    z = (got - mc['mean']) / (mc['se'] * math.sqrt(1. + 8 * float(mc['rays_per_run']) / n))
    assert N.abs(z).max() < 4.5, z

