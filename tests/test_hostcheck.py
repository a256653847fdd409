"""
CPU tier check of the per-ray core that the HIP kernels are made of: tracer_amd/csrc/trc_core.h compiled by g++
(tests/hostcheck, test-only, never loaded by the product) against the reference fixtures and the oracle.  The
authoritative parity tests are the GPU ones (test_gpu_parity.py); this tier catches core regressions in the build
container, where there is no GPU.
"""
import ctypes as C
import os
import subprocess

import numpy as N
import pytest

from helpers import load, case_names, source_dict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, 'tests', 'hostcheck', 'libtrc_hostcheck.so')


@pytest.fixture(scope='module')
def hc():
    subprocess.check_call(['make', '-C', ROOT, 'hostcheck'])
    return C.CDLL(SO)


def _p(a, typ=C.c_double):
    return a.ctypes.data_as(C.POINTER(typ))


def _desc(kind, frame, gm, extra, opt_kind=0, opt=()):
    from tracer_amd import _cabi
    from tracer_amd.geometry_manager import fill_desc
    d = _cabi.SurfaceDesc()
    fill_desc(d, frame, kind, list(gm), opt_kind, list(opt), extra_off=0 if len(extra) else -1, extra_len=len(extra))
    return d


def test_core_geometry_vs_reference(hc):
    g = load('geometry.npz')
    names = case_names(g)
    for ci in range(int(g['n_cases'])):
        pre = 'g%d_' % ci
        kind = int(g[pre + 'kind'])
        desc = _desc(kind, g[pre + 'frame'], g[pre + 'gm'], g[pre + 'extra'])
        v = N.ascontiguousarray(g[pre + 'v']); d = N.ascontiguousarray(g[pre + 'd']); extra = N.ascontiguousarray(g[pre + 'extra'])
        n = v.shape[1]
        t = N.empty(n)
        hc.hc_intersect(C.byref(desc), _p(extra), C.c_long(n), _p(v[0]), _p(v[1]), _p(v[2]), _p(d[0]), _p(d[1]), _p(d[2]), _p(t))
        assert N.array_equal(N.isfinite(t), N.isfinite(g[pre + 't'])), names[ci]
        idx = g[pre + 'hit_idx']
        assert N.allclose(t[idx], g[pre + 't'][idx], rtol=1e-9, atol=1e-8), names[ci]
        if len(idx):
            h = N.ascontiguousarray(g[pre + 'hits']); dd = N.ascontiguousarray(d[:, idx])
            nrm = N.empty_like(h)
            hc.hc_normals(C.byref(desc), C.c_long(len(idx)), _p(h[0]), _p(h[1]), _p(h[2]), _p(dd[0]), _p(dd[1]), _p(dd[2]),
                          _p(nrm[0]), _p(nrm[1]), _p(nrm[2]))
            ok = N.all(N.isclose(nrm, g[pre + 'normals'], rtol=1e-9, atol=1e-9) | (N.isnan(nrm) & N.isnan(g[pre + 'normals'])), axis=0)
            assert ok.all(), (names[ci], N.nonzero(~ok)[0][:5])


def test_core_optics_vs_oracle_same_streams(hc):
    """every optics fixture case through the device code compiled for the host (trc_shade_x: complex indices, materials and spectra
    included) against the oracle on the same Philox streams"""
    from oracle import optics
    o = load('optics.npz')
    frame = o['frame']
    nrm, d, e, wl = [N.ascontiguousarray(o[k]) for k in ('normals', 'dirs', 'energy', 'wavelengths')]
    H = d.shape[1]
    rid = N.arange(H, dtype=N.uint64) + N.uint64(2 ** 33 + 5)
    carried = 0
    for i, name in enumerate(case_names(o)):
        pre = 'o%d_' % i
        kind, opt, extra = int(o[pre + 'kind']), list(o[pre + 'opt']), N.ascontiguousarray(o[pre + 'extra'])
        ref_c = o[pre + 'ref_in']
        ref_in = N.ascontiguousarray(N.real(ref_c))
        mat = o[pre + 'mat'] if (pre + 'mat') in o.files else None
        spec = N.ascontiguousarray(o[pre + 'spec_in']) if (pre + 'spec_in') in o.files else None
        swl = N.ascontiguousarray(o[pre + 'spec_wl']) if spec is not None else None
        W = 0 if spec is None else spec.shape[0]
        wl_i = N.zeros(H) if spec is not None else wl
        ref_im = N.ascontiguousarray(N.imag(ref_c)) if N.iscomplexobj(ref_c) else None
        matr = None
        if mat is not None:
            matr = N.empty((2 * len(mat), H))
            matr[0::2], matr[1::2] = N.real(mat), N.imag(mat)
        desc = _desc(0, frame, [], extra, kind, opt)
        out = [N.empty(2 * H) for _ in range(5)]
        blk = N.empty(2 * H, dtype=N.int32)
        o_im = N.zeros(2 * H)
        o_spec = N.zeros((max(W, 1), 2 * H))
        path = N.ascontiguousarray(o[pre + 'path'])
        hc.hc_shade_x(C.byref(desc), _p(extra), C.c_long(H), _p(d[0]), _p(d[1]), _p(d[2]), _p(e), _p(ref_in), _p(wl_i), _p(nrm[0]), _p(nrm[1]),
                      _p(nrm[2]), _p(rid, C.c_uint64), C.c_uint64(987654321012), 2, *([_p(a) for a in out] + [_p(blk, C.c_int32), _p(path)]),
                      None if ref_im is None else _p(ref_im), 0 if matr is None else len(mat), None if matr is None else _p(matr),
                      W, None if spec is None else _p(swl), None if spec is None else _p(spec), _p(o_im), _p(o_spec))
        ext = {}
        if mat is not None:
            ext['mat'] = mat
        if spec is not None:
            ext.update(spec=spec, swl=swl)
        with N.errstate(all='ignore'):
            blocks = optics.shade(kind, opt, extra, frame[:3, 2], d, e, ref_c, wl_i, nrm, 987654321012, rid, 2, path=path, ext=ext)
        slots = N.concatenate([N.nonzero(blk == b)[0] for b in (0, 1)])
        par = N.where(slots < H, slots, slots - H)
        assert N.array_equal(par, N.hstack([b['sel'] for b in blocks])), name
        assert N.allclose(N.vstack([a[slots] for a in out[:3]]), N.hstack([b['directions'] for b in blocks]), rtol=1e-9, atol=1e-9), name
        assert N.allclose(out[3][slots], N.hstack([b['energy'] for b in blocks]), rtol=1e-9, atol=1e-12), name
        assert N.allclose(out[4][slots] + 1j * o_im[slots], N.hstack([b['ref'] for b in blocks]), rtol=1e-12, atol=0), name
        if spec is not None:
            assert N.allclose(o_spec[:, slots], N.hstack([b['spectra'] for b in blocks]), rtol=1e-12, atol=0), name
        carried += int(mat is not None or spec is not None)
    assert carried == 16


def test_core_sources_vs_oracle(hc):
    from oracle import sources
    from tracer_amd import _cabi
    s = load('sources.npz')
    for i, name in enumerate(case_names(s)):
        src = source_dict(s, 's%d_' % i)
        desc = _cabi.SourceDesc()
        desc.kind = src['kind']
        for k in range(3):
            desc.center[k] = src['center'][k]
        for k in range(9):
            desc.rot_pos[k] = src['rot_pos'].ravel()[k]
            desc.rot_dir[k] = src['rot_dir'].ravel()[k]
        for k in range(8):
            desc.p[k] = src['p'][k]
        desc.energy = src['energy']
        for k in range(_cabi.BUIE_LEN):
            desc.buie[k] = src['buie'][k]
        n = 20000
        v = N.empty((3, n)); d = N.empty((3, n))
        hc.hc_source(C.byref(desc), C.c_long(n), C.c_uint64(31337), C.c_uint64(10 ** 12), _p(v[0]), _p(v[1]), _p(v[2]), _p(d[0]), _p(d[1]), _p(d[2]))
        vo, do, eo, rid = sources.generate(src, n, 31337, 10 ** 12)
        assert N.allclose(v, vo, rtol=1e-10, atol=1e-8), name
        assert N.allclose(d, do, rtol=1e-9, atol=1e-11), name


def test_core_pow_pos_vs_libm(hc):
    """trc_pow_pos (the aureole inversion's power function) against numpy's pow over and beyond the aureole's range"""
    rng = N.random.RandomState(11)
    x = N.ascontiguousarray(N.hstack((rng.uniform(0.05, 0.6, 200000), 10. ** rng.uniform(-6, 6, 200000),
                                      [1., 2., 0.5, N.sqrt(2.), N.sqrt(0.5), 1. - 2 ** -53, 1. + 2 ** -52])))
    y = N.ascontiguousarray(N.hstack((rng.uniform(2.5, 4., 200000), rng.uniform(-0.8, 0.8, 200000), rng.uniform(-3, 3, 7))))
    out = N.empty_like(x)
    hc.hc_pow_pos(C.c_long(len(x)), _p(x), _p(y), _p(out))
    rel = N.abs(out / x ** y - 1.)
    assert rel.max() < 1.5e-15, rel.max()


def test_core_buie_staged_inversion_equals_plain(hc):
    """the per-bin folded form of sources.py:364-377 used by the streaming generation kernel against the plain form, on the
    Buie tables of the source fixtures (CSR 0.01 .. 0.3 and the CSR = 0 table) and on uniforms that cover every bin edge"""
    s = load('sources.npz')
    seen = 0
    for i, name in enumerate(case_names(s)):
        src = source_dict(s, 's%d_' % i)
        tab = N.ascontiguousarray(src['buie'], dtype=float)
        if not tab.any():
            continue
        seen += 1
        ne = (len(tab) - 6) // 3 - 1
        cdf = tab[2 * (ne + 1):3 * (ne + 1)]
        rng = N.random.RandomState(i)
        edges = N.hstack((cdf, N.nextafter(cdf, 0.), N.nextafter(cdf, 1.), N.arange(1024) / 1024., N.nextafter(N.arange(1, 1025) / 1024., 0.)))
        Rv = N.ascontiguousarray(N.hstack((rng.uniform(size=300000), edges[(edges >= 0.) & (edges < 1.)])))
        a, b = N.empty_like(Rv), N.empty_like(Rv)
        hc.hc_buie_theta(_p(tab), C.c_long(len(Rv)), _p(Rv), _p(a), _p(b))
        ok = N.isfinite(a)
        assert N.array_equal(ok, N.isfinite(b)), name
        assert N.allclose(a[ok], b[ok], rtol=1e-10, atol=1e-15), (name, N.abs(a[ok] - b[ok]).max())
    assert seen >= 2


def test_core_semi_lambertian_vs_oracle(hc):
    """SemiLambertian (optics_callables.py:506-531 as documented): the core against the oracle on the same streams, the
    mirror law for the glancing rays, the cone and the hemisphere for the others, specular block first"""
    from oracle import optics
    from tracer_amd import _cabi
    o = load('optics.npz')
    frame = o['frame']
    nrm, d, e, wl = [N.ascontiguousarray(o[k]) for k in ('normals', 'dirs', 'energy', 'wavelengths')]
    H = d.shape[1]
    rid = N.arange(H, dtype=N.uint64) + N.uint64(77)
    ref_in = N.ones(H)
    extra = N.zeros(1)
    ang_range, absorb = 0.8, 0.25
    desc = _desc(0, frame, [], [], _cabi.OPT_SEMI_LAMBERTIAN, [absorb, ang_range])
    out = [N.empty(2 * H) for _ in range(5)]
    blk = N.empty(2 * H, dtype=N.int32)
    hc.hc_shade(C.byref(desc), _p(extra), C.c_long(H), _p(d[0]), _p(d[1]), _p(d[2]), _p(e), _p(ref_in), _p(wl), _p(nrm[0]), _p(nrm[1]),
                _p(nrm[2]), _p(rid, C.c_uint64), C.c_uint64(4242), 3, *[_p(a) for a in out], _p(blk, C.c_int32))
    blocks = optics.shade(_cabi.OPT_SEMI_LAMBERTIAN, [absorb, ang_range], extra, frame[:3, 2], d, e, ref_in, wl, nrm, 4242, rid, 3)
    slots = N.concatenate([N.nonzero(blk == b)[0] for b in (0, 1)])
    assert N.array_equal(slots, N.hstack([b['sel'] for b in blocks]))
    got = N.vstack([a[slots] for a in out[:3]])
    assert N.allclose(got, N.hstack([b['directions'] for b in blocks]), rtol=1e-9, atol=1e-9)
    assert N.allclose(out[3][slots], N.hstack([b['energy'] for b in blocks]), rtol=1e-12)
    inc = N.arccos(-N.sum(d * nrm, axis=0))
    gl = inc > ang_range
    assert 0.1 * H < gl.sum() < 0.9 * H and N.array_equal(N.sort(blocks[0]['sel']), N.nonzero(gl)[0])
    mirrored = d[:, gl] - 2. * N.sum(d[:, gl] * nrm[:, gl], axis=0) * nrm[:, gl]
    assert N.allclose(blocks[0]['directions'], mirrored, atol=1e-12)
    cosn = N.sum(blocks[1]['directions'] * nrm[:, ~gl], axis=0)
    assert (cosn >= N.cos(ang_range) - 1e-9).all() and N.allclose(out[3][:H], e * (1. - absorb))


def test_core_kd_traversal_equals_brute_force(hc):
    """the device traversal (front-to-back with early exit) returns the brute-force (t, surface) for every ray"""
    from tracer_amd import scenes, _cabi
    from tracer_amd.accel_tree import KdTree
    from tracer_amd.scene import compile_scene
    from oracle import sources, engine
    plant, field, rec, src = scenes.nsttf_field()
    cs = compile_scene(plant)
    kd = KdTree(plant, 8 + 1.3 * N.log(cs.n_surf), min_leaf=1)
    f = kd.flat()
    d = _cabi.KdTreeDesc()
    d.n_nodes, d.n_leaf_surfs, d.n_always = len(f['flag']), len(f['leaf_surfs']), len(f['always_relevant'])
    i32 = C.POINTER(C.c_int32)
    d.flag, d.child, d.leaf_off, d.leaf_cnt = [f[k].ctypes.data_as(i32) for k in ('flag', 'child', 'leaf_off', 'leaf_cnt')]
    d.leaf_surfs, d.always_relevant = f['leaf_surfs'].ctypes.data_as(i32), f['always_relevant'].ctypes.data_as(i32)
    d.split = _p(f['split'])
    for k in range(6):
        d.bounds[k] = f['bounds'][k]
    n = 60000
    b = scenes.nsttf_source(n, src, seed=5)
    v, dr, e, rid = sources.generate(engine.source_from_desc(b.source_args()[0]), n, 5, 0)
    # second-bounce-like rays too: from mirror height towards the tower, and horizontal rays through the field
    rng = N.random.RandomState(2)
    v2 = N.vstack((rng.uniform(-130, 130, n // 3), rng.uniform(50, 190, n // 3), rng.uniform(0., 9., n // 3)))
    t2 = N.vstack((rng.uniform(-8, 8, n // 3), rng.uniform(-20, 20, n // 3), rng.uniform(0, 70., n // 3)))
    d2 = t2 - v2
    d2 /= N.sqrt(N.sum(d2 ** 2, axis=0))
    v = N.ascontiguousarray(N.hstack((v, v2))); dr = N.ascontiguousarray(N.hstack((dr, d2)))
    m = v.shape[1]
    tb, tk = N.empty(m), N.empty(m)
    sb, sk = N.empty(m, dtype=N.int32), N.empty(m, dtype=N.int32)
    extra = N.zeros(1)
    hc.hc_nearest(cs.n_surf, cs.descs, _p(extra), C.byref(d), C.c_long(m), _p(v[0]), _p(v[1]), _p(v[2]), _p(dr[0]), _p(dr[1]), _p(dr[2]),
                  _p(tb), _p(sb, C.c_int32), _p(tk), _p(sk, C.c_int32))
    assert (sb >= 0).sum() > 5000
    assert N.array_equal(sb, sk)
    assert N.array_equal(tb, tk)
    # the single-precision conservative search (fast engine): identical (t, surface), with and without the tree
    for use_kd in (True, False):
        t32, s32 = N.empty(m), N.empty(m, dtype=N.int32)
        rc = hc.hc_nearest32(cs.n_surf, cs.descs, _p(extra), C.byref(d) if use_kd else None, C.c_long(m), _p(v[0]), _p(v[1]), _p(v[2]),
                             _p(dr[0]), _p(dr[1]), _p(dr[2]), _p(t32), _p(s32, C.c_int32))
        assert rc == 0
        assert N.array_equal(s32, sb), use_kd
        assert N.array_equal(t32, tb), use_kd
    # the streaming engine's uniform grid + DDA: identical (t, surface) again
    t32, s32, st = N.empty(m), N.empty(m, dtype=N.int32), N.zeros(8)
    rc = hc.hc_nearest_grid(cs.n_surf, cs.descs, _p(extra), C.c_long(m), _p(v[0]), _p(v[1]), _p(v[2]), _p(dr[0]), _p(dr[1]), _p(dr[2]),
                            _p(t32), _p(s32, C.c_int32), _p(st))
    assert rc == 0
    assert N.array_equal(s32, sb) and N.array_equal(t32, tb)
    print('grid: %d cells, %d list entries; per ray %.1f cells, %.1f box tests, %.2f exact tests' % (st[3], st[4], st[0] / m, st[1] / m, st[2] / m))
    # and the oracle agrees with the brute-force core on which surface is hit first
    scene = engine.scene_from_compiled(cs)
    with N.errstate(all='ignore'):
        front, tmin = engine.intersect_ray(scene, v[:, :20000], dr[:, :20000])
    assert N.array_equal(front, sb[:20000])


def test_kd32_walk_with_origins_on_split_planes(hc):
    """
    The single-precision Kd walk when a ray starts within delta of a split plane (both children are then walked with the full
    interval): rays leaving a stack of plates from points on the lines where plates end -- the planes the tree splits at --
    find the brute-force (t, surface).  The scene is the one of examples/accel_tree_example.py:20-53; an early version of the walk
    stopped at the first pending child behind the best hit and lost the other child of such a node (3 rays in 4e5 on this scene).
    """
    from tracer_amd import _cabi
    from tracer_amd.accel_tree import KdTree
    from tracer_amd.scene import compile_scene
    from tracer_amd.assembly import Assembly
    from tracer_amd.object import AssembledObject
    from tracer_amd.surface import Surface
    from tracer_amd.flat_surface import RectPlateGM
    from tracer_amd.boundary_shape import BoundaryBox
    from tracer_amd.optics_callables import LambertianReceiver
    n = 10
    side = n + 1.
    objects = []
    for z in (-1., 0.):
        slab = AssembledObject(Surface(geometry=RectPlateGM(side, side), optics=LambertianReceiver(0.6)),
                               bounds=BoundaryBox([[-side / 2., -side / 2., 0.], [side / 2., side / 2., 0.]]))
        slab.set_location(N.array([0., 0., z]))
        objects.append(slab)
    for k in range(n):
        for i in range(n):
            for j in range(n):
                plate = AssembledObject(Surface(geometry=RectPlateGM(.8, .8), optics=LambertianReceiver(0.9)),
                                        bounds=BoundaryBox([[-.4, -.4, 0.], [.4, .4, 0.]]))
                plate.set_location(N.array([i + 0.5 - n / 2., j + 0.5 - n / 2., k + 1.]))
                objects.append(plate)
    asm = Assembly(objects=objects)
    cs = compile_scene(asm)
    f = KdTree(asm, 8 + 1.3 * N.log(cs.n_surf), min_leaf=1).flat()
    d = _cabi.KdTreeDesc()
    d.n_nodes, d.n_leaf_surfs, d.n_always = len(f['flag']), len(f['leaf_surfs']), len(f['always_relevant'])
    i32 = C.POINTER(C.c_int32)
    d.flag, d.child, d.leaf_off, d.leaf_cnt = [f[k].ctypes.data_as(i32) for k in ('flag', 'child', 'leaf_off', 'leaf_cnt')]
    d.leaf_surfs, d.always_relevant = f['leaf_surfs'].ctypes.data_as(i32), f['always_relevant'].ctypes.data_as(i32)
    d.split = _p(f['split'])
    for k in range(6):
        d.bounds[k] = f['bounds'][k]
    rng = N.random.RandomState(4)
    m = 300000
    # one coordinate within 1.5e-3 of a plate edge (delta is 1e-3 here), the start on a plate layer, cosine-law directions up or down
    edge = rng.randint(-5, 5, m) + rng.choice([0.1, 0.9], m) + rng.uniform(-1.5e-3, 1.5e-3, m)
    other = rng.uniform(-5.5, 5.5, m)
    on_x = rng.uniform(size=m) < 0.5
    v = N.ascontiguousarray(N.vstack((N.where(on_x, edge, other), N.where(on_x, other, edge), rng.randint(0, 11, m).astype(float))))
    th, ph = N.arcsin(N.sqrt(rng.uniform(0, 1, m))), rng.uniform(0, 2 * N.pi, m)
    dr = N.ascontiguousarray(N.vstack((N.sin(th) * N.cos(ph), N.sin(th) * N.sin(ph), rng.choice([-1., 1.], m) * N.cos(th))))
    tb, tk = N.empty(m), N.empty(m)
    sb, sk = N.empty(m, dtype=N.int32), N.empty(m, dtype=N.int32)
    extra = N.zeros(1)
    hc.hc_nearest(cs.n_surf, cs.descs, _p(extra), C.byref(d), C.c_long(m), _p(v[0]), _p(v[1]), _p(v[2]), _p(dr[0]), _p(dr[1]), _p(dr[2]),
                  _p(tb), _p(sb, C.c_int32), _p(tk), _p(sk, C.c_int32))
    assert (sb >= 0).sum() > m // 2 and N.array_equal(sb, sk) and N.array_equal(tb, tk)
    for use_kd in (True, False):
        t32, s32 = N.empty(m), N.empty(m, dtype=N.int32)
        rc = hc.hc_nearest32(cs.n_surf, cs.descs, _p(extra), C.byref(d) if use_kd else None, C.c_long(m), _p(v[0]), _p(v[1]), _p(v[2]),
                             _p(dr[0]), _p(dr[1]), _p(dr[2]), _p(t32), _p(s32, C.c_int32))
        assert rc == 0
        assert N.array_equal(s32, sb) and N.array_equal(t32, tb), use_kd
    t32, s32, st = N.empty(m), N.empty(m, dtype=N.int32), N.zeros(8)
    rc = hc.hc_nearest_grid(cs.n_surf, cs.descs, _p(extra), C.c_long(m), _p(v[0]), _p(v[1]), _p(v[2]), _p(dr[0]), _p(dr[1]), _p(dr[2]),
                            _p(t32), _p(s32, C.c_int32), _p(st))
    assert rc == 0 and N.array_equal(s32, sb) and N.array_equal(t32, tb)


def _kd_desc(asm, n_surf):
    from tracer_amd import _cabi
    from tracer_amd.accel_tree import KdTree
    f = KdTree(asm, 8 + 1.3 * N.log(n_surf), min_leaf=1).flat()
    d = _cabi.KdTreeDesc()
    d.n_nodes, d.n_leaf_surfs, d.n_always = len(f['flag']), len(f['leaf_surfs']), len(f['always_relevant'])
    i32 = C.POINTER(C.c_int32)
    d.flag, d.child, d.leaf_off, d.leaf_cnt = [f[k].ctypes.data_as(i32) for k in ('flag', 'child', 'leaf_off', 'leaf_cnt')]
    d.leaf_surfs, d.always_relevant = f['leaf_surfs'].ctypes.data_as(i32), f['always_relevant'].ctypes.data_as(i32)
    d.split = _p(f['split'])
    for k in range(6):
        d.bounds[k] = f['bounds'][k]
    return d, f                   # f keeps the arrays alive


def test_nearest_hit_searches_on_random_scenes(hc):
    """
    Every candidate search of the engines -- the float64 Kd walk, the single-precision walk with and without the tree, the uniform
    grid with its DDA -- against brute force on scenes of 120 plates, discs, spheres, hemispheres, cylinders and dishes thrown
    into a cube: once turned at random, once on integer positions with quarter turns (coincident planes, faces in cell and split
    planes), with axis-parallel rays and rays that start on those planes; then again for rays that leave the points just hit.
    """
    from tracer_amd.scene import compile_scene
    from tracer_amd.assembly import Assembly
    from tracer_amd.object import AssembledObject
    from tracer_amd.surface import Surface
    from tracer_amd.flat_surface import RectPlateGM, RoundPlateGM
    from tracer_amd.sphere_surface import SphericalGM, HemisphereGM
    from tracer_amd.cylinder import FiniteCylinder
    from tracer_amd.paraboloid import ParabolicDishGM
    from tracer_amd.boundary_shape import BoundaryBox
    from tracer_amd.optics_callables import Reflective
    from tracer_amd.spatial_geometry import generate_transform

    def scene(rng, aligned):
        objs = []
        for _ in range(120):
            kind, s = rng.randint(0, 6), rng.uniform(0.2, 1.5)
            if kind == 0:
                gm, lo, hi = RectPlateGM(2 * s, s), [-s, -s / 2, 0], [s, s / 2, 0]
            elif kind == 1:
                gm, lo, hi = RoundPlateGM(s), [-s, -s, 0], [s, s, 0]
            elif kind == 2:
                gm, lo, hi = SphericalGM(s), [-s, -s, -s], [s, s, s]
            elif kind == 3:
                gm, lo, hi = HemisphereGM(s), [-s, -s, -s], [s, s, 0]
            elif kind == 4:
                gm, lo, hi = FiniteCylinder(2 * s, 3 * s), [-s, -s, -1.5 * s], [s, s, 1.5 * s]
            else:
                f = rng.uniform(0.5, 2.)
                gm, lo, hi = ParabolicDishGM(2 * s, f), [-s, -s, 0], [s, s, s * s / (4 * f)]
            o = AssembledObject(Surface(gm, Reflective(0.1)), bounds=BoundaryBox([lo, hi]))
            loc = rng.uniform(-6., 6., 3)
            if aligned:
                o.set_transform(generate_transform(N.r_[1., 0, 0], rng.choice([0., N.pi / 2, N.pi]), N.round(loc)[:, None]))
            else:
                ax = rng.normal(size=3)
                o.set_transform(generate_transform(ax / N.linalg.norm(ax), rng.uniform(0, 2 * N.pi), loc[:, None]))
            objs.append(o)
        return Assembly(objects=objs)

    def searches(cs, d, v, dr):
        m = v.shape[1]
        extra = N.ascontiguousarray(cs.extra if len(cs.extra) else N.zeros(1))
        rays = (C.c_long(m), _p(v[0]), _p(v[1]), _p(v[2]), _p(dr[0]), _p(dr[1]), _p(dr[2]))
        tb, tk, sb, sk = N.empty(m), N.empty(m), N.empty(m, dtype=N.int32), N.empty(m, dtype=N.int32)
        assert hc.hc_nearest(cs.n_surf, cs.descs, _p(extra), C.byref(d), *rays, _p(tb), _p(sb, C.c_int32), _p(tk), _p(sk, C.c_int32)) == 0
        got = {'float64 Kd walk': (sk, tk)}
        for use_kd in (True, False):
            t32, s32 = N.empty(m), N.empty(m, dtype=N.int32)
            assert hc.hc_nearest32(cs.n_surf, cs.descs, _p(extra), C.byref(d) if use_kd else None, *rays, _p(t32), _p(s32, C.c_int32)) == 0
            got['float32 walk, tree %s' % use_kd] = (s32, t32)
        t32, s32, st = N.empty(m), N.empty(m, dtype=N.int32), N.zeros(8)
        assert hc.hc_nearest_grid(cs.n_surf, cs.descs, _p(extra), *rays, _p(t32), _p(s32, C.c_int32), _p(st)) == 0
        got['grid'] = (s32, t32)
        for name, (s_, t_) in got.items():
            assert N.array_equal(s_, sb), name
            assert N.array_equal(t_[sb >= 0], tb[sb >= 0]), name
        return sb, tb

    for aligned in (False, True):
        rng = N.random.RandomState(31 + aligned)
        asm = scene(rng, aligned)
        cs = compile_scene(asm)
        d, keep = _kd_desc(asm, cs.n_surf)
        m = 40000
        v = N.ascontiguousarray(rng.uniform(-8, 8, (3, m)))
        dr = rng.normal(size=(3, m))
        dr /= N.sqrt((dr ** 2).sum(axis=0))
        if aligned:
            dr[:, :m // 10] = N.eye(3)[:, rng.randint(0, 3, m // 10)] * rng.choice([-1., 1.], m // 10)
            v[:, m // 10: m // 5] = N.round(v[:, m // 10: m // 5])
        dr = N.ascontiguousarray(dr)
        sb, tb = searches(cs, d, v, dr)
        hit = sb >= 0
        assert hit.sum() > m // 5
        v2 = N.ascontiguousarray(v[:, hit] + tb[hit] * dr[:, hit])
        d2 = rng.normal(size=v2.shape)
        sb2, _ = searches(cs, d, v2, N.ascontiguousarray(d2 / N.sqrt((d2 ** 2).sum(axis=0))))
        assert (sb2 >= 0).sum() > hit.sum() // 3


def _fp(hc, cs, desc, n, M=512, seed=77, offset=0):
    out = N.zeros(10)
    why = C.create_string_buffer(128)
    extra = N.ascontiguousarray(cs.extra if len(cs.extra) else N.zeros(1))
    rc = hc.hc_footprint(cs.n_surf, cs.descs, _p(extra), C.byref(desc), C.c_long(n), C.c_uint64(seed), C.c_uint64(offset), M, _p(out),
                         why, 128)
    return rc, out, why.value.decode()


def test_footprint_map_is_conservative(hc):
    """
    trc_footprint.h (the streaming engine's fresh-ray kernel): rays of the scene's source generated in float64; every ray whose
    brute-force nearest hit exists (and that does not belong to the Buie aureole, which takes the general path) starts in a
    set cell of the mask, finds the surface in the cell's list and passes its oriented-box test; the float32 start point of
    stage A stays inside the margin the map was built with.
    """
    from tracer_amd import scenes, sources
    from tracer_amd.scene import compile_scene
    from tracer_amd.assembly import Assembly
    from tracer_amd.object import AssembledObject
    from tracer_amd.surface import Surface
    from tracer_amd.flat_surface import RectPlateGM, RoundPlateGM
    from tracer_amd.paraboloid import ParabolicDishGM
    from tracer_amd.sphere_surface import HemisphereGM
    from tracer_amd.cylinder import FiniteCylinder
    from tracer_amd import optics_callables as opt
    from tracer_amd.spatial_geometry import translate, rotx, roty, rotz
    hc.hc_footprint.restype = C.c_int
    # NSTTF, Buie disc: the bench workload
    plant, field, rec, src = scenes.nsttf_field()
    cs = compile_scene(plant)
    for M in (512, 256):
        rc, o, why = _fp(hc, cs, scenes.nsttf_source(10, src, seed=1).source_args()[0], 150000, M=M)
        assert rc == 0, why
        print('NSTTF M=%d: coverage %.3f, rays with the bit set %.3f, hits %.4f, generic %.4f, candidates per listed ray %.2f, box-passing %.2f, |f32-f64| %.2e (eps %.2e)'
              % (M, o[7], o[2] / o[0], o[3] / o[0], o[1] / o[0], o[5] / max(o[2], 1), o[6] / max(o[2], 1), o[8], o[9]))
        assert o[4] == 0 and o[3] > 5000 and o[8] < 0.1 * o[9]
        assert o[2] / o[0] < 0.35          # the map culls most of the disc
    # the dish under a Buie disc (everything is footprint), pillbox disc and rectangle sources over a small mixed scene, turned
    # and shifted, and a source that is oblique to its own plane
    asm, dish_surf, rec_surf, dsrc = scenes.dish()
    rc, o, why = _fp(hc, compile_scene(asm), scenes.dish_source(10, dsrc, seed=1).source_args()[0], 60000)
    assert rc == 0 and o[4] == 0 and o[3] > 50000, (why, o)
    rng = N.random.RandomState(5)
    objs = []
    for k in range(14):
        gm = [RectPlateGM(1.2, 0.7), RoundPlateGM(0.6), ParabolicDishGM(1.4, 1.1), HemisphereGM(0.5), FiniteCylinder(0.8, 1.1)][k % 5]
        tr = N.dot(translate(*rng.uniform(-4, 4, 3)), N.dot(rotx(rng.uniform(0, 6.3)), N.dot(roty(rng.uniform(0, 6.3)), rotz(rng.uniform(0, 6.3)))))
        objs.append(AssembledObject(surfs=[Surface(gm, opt.Reflective(0.1))], transform=tr))
    mixed = compile_scene(Assembly(objects=objs))
    direction = N.r_[0.3, -0.2, -1.] / N.linalg.norm([0.3, -0.2, -1.])
    center = N.c_[-25. * direction]
    cases = [sources.disk_bundle(10, center, direction, 9., 0.02, flux=1., seed=1),
             sources.disk_bundle(10, center, direction, 9., 0.004, flux=1., radius_in=2., angular_span=[0.3, 5.1], seed=1),
             sources.rect_bundle(10, center, direction, 16., 13., 0.01, flux=1., seed=1),
             sources.rect_bundle(10, N.c_[[0., 0., 30.]], N.r_[0., 0., -1.], 16., 13., 0.01, flux=1., seed=1),
             sources.buie_sunshape(10, center, direction, 9., 0.05, flux=1., seed=1),
             sources.rect_buie_sunshape(10, center, direction, 17., 15., 0.1, flux=1., seed=1),
             sources.oblique_solar_rect_bundle(10, N.c_[[-9., 6., 30.]], N.r_[0., 0., -1.], direction, 22., 20., 0.008, flux=1., seed=1)]
    for k, b in enumerate(cases):
        rc, o, why = _fp(hc, mixed, b.source_args()[0], 80000, seed=100 + k)
        assert rc == 0, (k, why)
        assert o[4] == 0 and o[3] > 1000 and o[8] < 0.1 * o[9], (k, list(o))
    # not applicable: a wide cone, a disc with x_cut, a scene with an unbounded plane
    rc, o, why = _fp(hc, mixed, sources.disk_bundle(10, center, direction, 9., 1.2, flux=1., seed=1).source_args()[0], 10)
    assert rc == -3 and 'cone' in why
    rc, o, why = _fp(hc, mixed, sources.disk_bundle(10, center, direction, 9., 0.01, flux=1., x_cut=1., seed=1).source_args()[0], 10)
    assert rc == -3 and 'x_cut' in why
    from tracer_amd.flat_surface import FlatGeometryManager
    unb = compile_scene(Assembly(objects=objs + [AssembledObject(surfs=[Surface(FlatGeometryManager(), opt.Reflective(0.))], transform=translate(0, 0, -9))]))
    rc, o, why = _fp(hc, unb, cases[0].source_args()[0], 10)
    assert rc == -3 and 'unbounded' in why


def test_oriented_box_never_rejects_a_hit(hc):
    """trc_obb_hit32 (float32, the candidate test in front of every exact test) on the geometry fixtures: a ray that the
    exact float64 test of the kind accepts passes the kind's oriented box -- all bounded kinds, three frames, 240 rays each"""
    hc.hc_obb.restype = C.c_long
    g = load('geometry.npz')
    names = case_names(g)
    tested = 0
    for ci in range(int(g['n_cases'])):
        pre = 'g%d_' % ci
        kind = int(g[pre + 'kind'])
        desc = _desc(kind, g[pre + 'frame'], g[pre + 'gm'], g[pre + 'extra'])
        v = N.ascontiguousarray(g[pre + 'v']); d = N.ascontiguousarray(g[pre + 'd']); extra = N.ascontiguousarray(g[pre + 'extra'])
        if not len(extra):
            extra = N.zeros(1)
        passed = C.c_long(0)
        bad = hc.hc_obb(C.byref(desc), _p(extra), C.c_long(v.shape[1]), _p(v[0]), _p(v[1]), _p(v[2]), _p(d[0]), _p(d[1]), _p(d[2]), C.byref(passed))
        assert bad == 0, (names[ci], bad)
        tested += int(N.isfinite(g[pre + 't']).sum())
    assert tested > 2000


def test_large_grid_search_on_triangle_soups(hc):
    """
    The search of scenes beyond LDS (trc_accel_build_grid32 + what trc_nearest_grid32 / k_s_bounce<2> / k_s_bounce_coop do per ray:
    a face listed only in the cells it touches, trc_tri_hit32 in front of the exact test, the walk that ends behind the best hit)
    against brute force, on what a smooth relief does not have: 4000 triangles thrown into a box -- needles (edges 1000 : 1),
    specks, faces that cross dozens of cells, faces that cut each other -- with a few plates among them (listed by their boxes)
    and one plate far away (set apart from the grid).  Rays from outside, from inside, along the faces' planes, parallel to the
    axes, and leaving from the points just hit (two generations): the same (t, surface) for every ray.
    """
    from tracer_amd import _cabi as K
    from tracer_amd.models.triangulated_surface import TriangulatedSurface
    from tracer_amd.assembly import Assembly
    from tracer_amd.object import AssembledObject
    from tracer_amd.surface import Surface
    from tracer_amd.flat_surface import RectPlateGM
    from tracer_amd.spatial_geometry import translate, general_axis_rotation
    from tracer_amd import optics_callables as opt
    from tracer_amd.scene import compile_scene
    for seed, box in ((3, 10.), (4, 0.02)):
        rng = N.random.default_rng(seed)
        nt = 4000
        c = rng.uniform(-box, box, size=(nt, 3))
        size = box * 10. ** rng.uniform(-3.5, -0.3, size=nt)                # specks to faces a third of the box long
        e1, e2 = rng.normal(size=(nt, 3)), rng.normal(size=(nt, 3))
        e1 *= (size / N.linalg.norm(e1, axis=1))[:, None]
        needle = rng.random(nt) < 0.3
        e2 *= (size * N.where(needle, 1e-3, rng.uniform(0.2, 1., nt)) / N.linalg.norm(e2, axis=1))[:, None]
        V = N.vstack((c, c + e1, c + e2))
        F = N.c_[N.arange(nt), N.arange(nt) + nt, N.arange(nt) + 2 * nt]
        soup = TriangulatedSurface(V, F, opt.Reflective(0.1))
        nf = len(soup.get_surfaces())
        plates = []
        for k in range(40):
            ax = rng.normal(size=3)
            tr = N.eye(4)
            tr[:3, :3] = general_axis_rotation(ax / N.linalg.norm(ax), rng.uniform(0., N.pi))
            tr[:3, 3] = rng.uniform(-box, box, size=3)
            plates.append(AssembledObject(surfs=[Surface(RectPlateGM(0.3 * box, 0.1 * box), opt.Reflective(0.1))], transform=tr))
        far = AssembledObject(surfs=[Surface(RectPlateGM(8. * box, 8. * box), opt.Reflective(0.1))], transform=translate(0., 0., 12. * box))
        cs = compile_scene(Assembly(objects=[soup] + plates + [far]))
        assert cs.n_surf == nf + 41 and nf > 3600          # (needles whose cross product vanishes are dropped, as in the reference)
        m = 60000
        v = rng.uniform(-1.5 * box, 1.5 * box, size=(3, m))
        d = rng.normal(size=(3, m))
        k = m // 4
        tgt = rng.uniform(-box, box, size=(3, k))
        v[:, :k] = 6. * box * (lambda u: u / N.linalg.norm(u, axis=0))(rng.normal(size=(3, k)))       # from outside, aimed inside
        d[:, :k] = tgt - v[:, :k]
        # along the planes of faces: from a point of the face's plane towards its inside
        f = rng.integers(0, nt, k)
        a, b = rng.uniform(-2., 3., k), rng.uniform(-2., 3., k)
        v[:, k:2 * k] = (c[f] + a[:, None] * e1[f] + b[:, None] * e2[f]).T
        d[:, k:2 * k] = (c[f] + (e1[f] + e2[f]) / 3.).T - v[:, k:2 * k] + 1e-9 * box * rng.normal(size=(3, k))
        # parallel to an axis
        ax = rng.integers(0, 3, k)
        d[:, 2 * k:3 * k] = 0.
        d[ax, N.arange(2 * k, 3 * k)] = rng.choice([-1., 1.], k)
        d /= N.linalg.norm(d, axis=0)
        extra = N.zeros(1)
        total = 0
        for gen in range(3):
            v, d = N.ascontiguousarray(v), N.ascontiguousarray(d)
            mm = v.shape[1]
            tb, tk, sb, sk = N.empty(mm), N.empty(mm), N.empty(mm, dtype=N.int32), N.empty(mm, dtype=N.int32)
            hc.hc_nearest(cs.n_surf, cs.descs, _p(extra), None, C.c_long(mm), _p(v[0]), _p(v[1]), _p(v[2]), _p(d[0]), _p(d[1]), _p(d[2]),
                          _p(tb), _p(sb, C.c_int32), _p(tk), _p(sk, C.c_int32))
            t32, s32, st = N.empty(mm), N.empty(mm, dtype=N.int32), N.zeros(8)
            rc = hc.hc_nearest_grid32(cs.n_surf, cs.descs, _p(extra), C.c_long(mm), _p(v[0]), _p(v[1]), _p(v[2]), _p(d[0]), _p(d[1]), _p(d[2]),
                                      _p(t32), _p(s32, C.c_int32), _p(st))
            assert rc == 0
            bad = N.nonzero((s32 != sb) | (t32 != tb))[0]
            assert len(bad) == 0, (seed, gen, len(bad), bad[:5], sb[bad[:5]], s32[bad[:5]], tb[bad[:5]], t32[bad[:5]])
            hit = sb >= 0
            total += int(hit.sum())
            if gen == 0:
                print('seed %d: %d cells, %d list entries for %d faces; per ray %.1f cells, %.1f faces looked at, %.2f exact tests; %d of %d rays hit'
                      % (seed, st[3], st[4], nf, st[0] / mm, st[1] / mm, st[2] / mm, hit.sum(), mm))
            # the next generation leaves from the points hit, in random directions
            p = v[:, hit] + tb[hit] * d[:, hit]
            dn = rng.normal(size=p.shape)
            v, d = p, dn / N.linalg.norm(dn, axis=0)
        assert total > 0.5 * m


def test_core_henyey_greenstein_vs_reference(hc):
    """trc_hg_theta (the device's scattering angle) on the reference's recorded draws, sampling.py:160-168"""
    g = load('scattering.npz')
    for k, gv in enumerate(g['hg_g']):
        R = N.ascontiguousarray(g['hg%d_R' % k])
        th = N.empty_like(R)
        hc.hc_hg_theta(C.c_double(float(gv)), C.c_long(len(R)), _p(R), _p(th))
        assert N.allclose(th, g['hg%d_theta' % k], rtol=0., atol=1e-12), gv
