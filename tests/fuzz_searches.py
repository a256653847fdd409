"""
Long-running companion of test_hostcheck.py (not collected by pytest): the candidate searches of the engines, compiled for the
CPU by `make hostcheck`, against brute force on chains of rays -- a sun-like beam into a scene, then from every point hit in a
random direction, six bounces deep; a fifth of the restarted rays are moved onto a split plane of the Kd-tree (within 1.5 delta),
where the single-precision walk takes both children.  One process does ~2e5 rays per second.

    python tests/fuzz_searches.py <mixed|aligned|plates> <kd32|grid> <seed> <rays>

Round 2: 1.9e8 rays on the two scenes of mixed shapes and 8e7 on the plates with kd32, 2.4e8 on the mixed scenes with grid: no
mismatch (the walk as it was before test_kd32_walk_with_origins_on_split_planes lost 7 hits per million on the plates).
"""
import ctypes as C
import os
import sys
import time

import numpy as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def _p(a, t=C.c_double):
    return a.ctypes.data_as(C.POINTER(t))


def mixed_scene(aligned):
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
    rng = N.random.RandomState(5 + aligned)
    objs = []
    for _ in range(150):
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
        o = AssembledObject(Surface(gm, Reflective(0.2)), bounds=BoundaryBox([lo, hi]))
        loc = rng.uniform(-6., 6., 3)
        if aligned:
            o.set_transform(generate_transform(N.r_[1., 0, 0], rng.choice([0., N.pi / 2, N.pi]), N.round(loc)[:, None]))
        else:
            ax = rng.normal(size=3)
            o.set_transform(generate_transform(ax / N.linalg.norm(ax), rng.uniform(0, 2 * N.pi), loc[:, None]))
        objs.append(o)
    return Assembly(objects=objs), N.r_[0.2, -0.1, -1.]


def main():
    which, search, seed, total = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(float(sys.argv[4]))
    from tracer_amd import _cabi
    from tracer_amd.accel_tree import KdTree
    from tracer_amd.scene import compile_scene
    hc = C.CDLL(os.path.join(ROOT, 'tests', 'hostcheck', 'libtrc_hostcheck.so'))
    if which == 'plates':
        from helpers import plates_scene
        asm, beam = plates_scene()[0], N.r_[0.05, -0.02, -1.]
    else:
        asm, beam = mixed_scene(which == 'aligned')
    beam = beam / N.linalg.norm(beam)
    cs = compile_scene(asm)
    f = KdTree(asm, 8 + 1.3 * N.log(cs.n_surf), min_leaf=1).flat()
    kd = _cabi.KdTreeDesc()
    kd.n_nodes, kd.n_leaf_surfs, kd.n_always = len(f['flag']), len(f['leaf_surfs']), len(f['always_relevant'])
    i32 = C.POINTER(C.c_int32)
    kd.flag, kd.child, kd.leaf_off, kd.leaf_cnt = [f[k].ctypes.data_as(i32) for k in ('flag', 'child', 'leaf_off', 'leaf_cnt')]
    kd.leaf_surfs, kd.always_relevant = f['leaf_surfs'].ctypes.data_as(i32), f['always_relevant'].ctypes.data_as(i32)
    kd.split = _p(f['split'])
    for k in range(6):
        kd.bounds[k] = f['bounds'][k]
    planes = {a: N.unique(f['split'][f['flag'] == a]) for a in range(3)}
    extra = N.ascontiguousarray(cs.extra if len(cs.extra) else N.zeros(1))
    rng = N.random.RandomState(seed)

    def compare(v, dr):
        m = v.shape[1]
        rays = (C.c_long(m), _p(v[0]), _p(v[1]), _p(v[2]), _p(dr[0]), _p(dr[1]), _p(dr[2]))
        tb, tk, sb, sk = N.empty(m), N.empty(m), N.empty(m, dtype=N.int32), N.empty(m, dtype=N.int32)
        hc.hc_nearest(cs.n_surf, cs.descs, _p(extra), None, *rays, _p(tb), _p(sb, C.c_int32), _p(tk), _p(sk, C.c_int32))
        t32, s32, st = N.empty(m), N.empty(m, dtype=N.int32), N.zeros(8)
        if search == 'grid':
            hc.hc_nearest_grid(cs.n_surf, cs.descs, _p(extra), *rays, _p(t32), _p(s32, C.c_int32), _p(st))
        else:
            hc.hc_nearest32(cs.n_surf, cs.descs, _p(extra), C.byref(kd), *rays, _p(t32), _p(s32, C.c_int32))
        bad = N.nonzero((s32 != sb) | ((t32 != tb) & (sb >= 0)))[0]
        for b in bad[:5]:
            print('MISMATCH', which, search, 'ray', v[:, b].tolist(), dr[:, b].tolist(), 'brute force', sb[b], tb[b], 'search', s32[b], t32[b], flush=True)
        return sb, tb, bad.size

    e1 = N.cross(beam, [1., 0, 0])
    e1 /= N.linalg.norm(e1)
    e2 = N.cross(beam, e1)
    t0, done, wrong = time.time(), 0, 0
    while done < total:
        k = 500000
        r, ph = 9. * N.sqrt(rng.uniform(size=k)), rng.uniform(0, 2 * N.pi, k)
        v = N.ascontiguousarray((-12. * beam)[:, None] + e1[:, None] * r * N.cos(ph) + e2[:, None] * r * N.sin(ph))
        dr = beam[:, None] + 5e-3 * rng.normal(size=(3, k))
        dr = N.ascontiguousarray(dr / N.sqrt((dr ** 2).sum(axis=0)))
        for bounce in range(6):
            sb, tb, nb = compare(v, dr)
            wrong += nb
            done += v.shape[1]
            hit = sb >= 0
            if hit.sum() < 1000:
                break
            v = N.ascontiguousarray(v[:, hit] + tb[hit] * dr[:, hit])
            for a in range(3):
                if len(planes[a]):
                    pick = rng.uniform(size=v.shape[1]) < 0.2
                    j = N.clip(N.searchsorted(planes[a], v[a, pick]), 0, len(planes[a]) - 1)
                    v[a, pick] = planes[a][j] + rng.uniform(-1.5e-3, 1.5e-3, pick.sum())
            dr = rng.normal(size=v.shape)
            dr = N.ascontiguousarray(dr / N.sqrt((dr ** 2).sum(axis=0)))
    print(which, search, 'seed', seed, 'rays', done, 'mismatches', wrong, '%.0f s' % (time.time() - t0), flush=True)


if __name__ == '__main__':
    main()
