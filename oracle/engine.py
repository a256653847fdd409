"""
TracerEngine.ray_tracer restated (tracer/tracer_engine.py:124-295, :27-64) on the flattened scene table.

scene: list of dicts (kind, opt_kind, frame(4x4), gm, opt, extra) in Assembly.get_surfaces() order.
The loop is the reference's: per bounce, every surface computes its parametric distances for the
whole bundle, the nearest hit is chosen with `t == 0 -> inf` and strict `<` in surface order, then
surfaces are shaded in index order, outputs concatenated surface-major, rays with E <= min_energy
culled and, in the recorded level, moved behind the live ones.
"""
import numpy as N
from .kinds import *
from . import geometry, optics, sources


def scene_from_compiled(cs):
    """Plain-python view of a tracer_amd CompiledScene (ctypes table) -- keeps the oracle free of product imports."""
    out = []
    extra = N.asarray(cs.extra, dtype=float)
    for d in cs.descs:
        fr = N.eye(4)
        fr[:3] = N.array(list(d.frame)).reshape(3, 4)
        ex = extra[d.extra_off:d.extra_off + d.extra_len] if d.extra_off >= 0 else None
        out.append(dict(kind=d.gm_kind, opt_kind=d.optics_kind, frame=fr, gm=list(d.gm), opt=list(d.opt), extra=ex))
    return out


def source_from_desc(desc):
    return dict(kind=desc.kind, center=N.array(list(desc.center)), rot_pos=N.array(list(desc.rot_pos)).reshape(3, 3),
                rot_dir=N.array(list(desc.rot_dir)).reshape(3, 3), p=list(desc.p), energy=desc.energy,
                buie=N.array(list(desc.buie)))


def intersect_ray(scene, v, d):
    """tracer_engine.py:27-64 with all surfaces relevant"""
    n = v.shape[1]
    mins = N.ones(n) * N.inf
    earliest = -1 * N.ones(n, dtype=int)
    for si, s in enumerate(scene):
        t = geometry.intersect(s['kind'], s['frame'], s['gm'], s['extra'], v, d)
        t[t == 0.] = N.inf
        earlier = t < mins
        mins[earlier] = t[earlier]
        earliest[earlier] = si
    return earliest, mins


def trace(scene, v, d, e, ref, wl, rid, reps, min_energy, seed, mat=None, spec=None, swl=None):
    """
    Returns dict(levels=[dict(vertices, directions, energy, parents, surf, ref, n_live)...] (level 0 = input),
    absorbed(S), received(S), hits(S), segments, hit_records=[per level dict(surf, e_in, e_out, points, directions)])
    mat (K, n) complex: the scene's materials at each ray's wavelength; spec, swl (W, n): polychromatic bundle.  Children
    inherit all three (RayBundle.inherit); levels then carry `spectra` and `swl`.
    """
    S = len(scene)
    carried = dict((k, a) for k, a in (('mat', mat), ('spec', spec), ('swl', swl)) if a is not None)
    absorbed = N.zeros(S)
    received = N.zeros(S)
    hits = N.zeros(S, dtype=N.int64)
    levels = [dict(vertices=v, directions=d, energy=e, parents=N.zeros(v.shape[1], dtype=int), surf=-N.ones(v.shape[1], dtype=int),
                   ref=ref, n_live=v.shape[1])]
    segments = 0
    events = 0          # interactions: surface hits and volume events (the device's `hits` statistic)
    for it in range(reps):
        n = v.shape[1]
        if n == 0:
            break
        segments += n
        front, tmin = intersect_ray(scene, v, d)
        outs = []
        for si, s in enumerate(scene):
            sel = N.nonzero(front == si)[0]
            if len(sel) == 0:
                continue
            pts = v[:, sel] + tmin[sel][None, :] * d[:, sel]
            nrm = geometry.normals(s['kind'], s['frame'], s['gm'], pts, d[:, sel])
            blocks = optics.shade(s['opt_kind'], s['opt'], s['extra'], s['frame'][:3, 2], d[:, sel], e[sel], ref[sel], wl[sel],
                                  nrm, seed, rid[sel], it + 1, path=N.sqrt(N.sum((pts - v[:, sel]) ** 2, axis=0)),
                                  ext=dict((k, a[:, sel]) for k, a in carried.items()))
            e_out = N.zeros(len(sel))
            reached = N.ones(len(sel), dtype=bool)      # False: scattered in the medium on the way -- the surface records nothing
            for b in blocks:
                N.add.at(e_out, b['sel'], b['energy'])
                if 'back' in b:
                    reached[b['sel']] = False
            absorbed[si] += N.sum((e[sel] - e_out)[reached])
            received[si] += N.sum(e[sel][reached])
            hits[si] += int(reached.sum())
            events += len(sel)
            for b in blocks:
                k = b['sel']
                start = pts[:, k] if 'back' not in b else pts[:, k] - b['back'][None, :] * d[:, sel][:, k]
                if 'shift' in b:            # a periodic boundary: the ray goes on one period along the oriented normal
                    start = start + b['shift'][None, :] * nrm[:, k]
                o = dict(vertices=start, directions=b['directions'], energy=b['energy'], parents=sel[k],
                         surf=N.full(len(k), si), ref=b['ref'], wl=wl[sel][k], rid=b['rid'])
                if 'mat' in carried:
                    o['mat'] = carried['mat'][:, sel][:, k]
                if 'spec' in carried:
                    o['spectra'] = b['spectra']
                    o['swl'] = carried['swl'][:, sel][:, k]
                outs.append(o)
        if not outs:                      # every ray escaped: "Ray bundle depleted" (tracer_engine.py:277-280)
            v, d, e = N.zeros((3, 0)), N.zeros((3, 0)), N.zeros(0)
            break
        cat = dict((key, N.hstack([o[key] for o in outs])) for key in outs[0])
        weak = cat['energy'] <= min_energy                               # tracer_engine.py:242
        order = N.concatenate((N.nonzero(~weak)[0], N.nonzero(weak)[0]))   # live first, culled last (:270-274)
        n_live = int((~weak).sum())
        rec = dict((key, val[..., order]) for key, val in cat.items())
        rec['n_live'] = n_live
        levels.append(rec)
        v, d, e = rec['vertices'][:, :n_live], rec['directions'][:, :n_live], rec['energy'][:n_live]
        ref, wl, rid = rec['ref'][:n_live], rec['wl'][:n_live], rec['rid'][:n_live]
        for k, name in (('mat', 'mat'), ('spec', 'spectra'), ('swl', 'swl')):
            if k in carried:
                carried[k] = rec[name][:, :n_live]
    return dict(levels=levels, absorbed=absorbed, received=received, hits=hits, segments=segments, events=events,
                last_vertices=v, last_directions=d, last_energy=e)


def trace_from_compiled(cs, source_args, reps, min_energy):
    """Trace a tracer_amd CompiledScene from a (desc, n, seed, offset) source tuple."""
    desc, n, seed, offset = source_args
    scene = scene_from_compiled(cs)
    v, d, e, rid = sources.generate(source_from_desc(desc), n, seed, offset)
    return trace(scene, v, d, e, N.ones(n), N.zeros(n), rid, reps, min_energy, seed)


def trace_bundle(cs, vertices, directions, energy, reps, min_energy, seed, ref_index=None, wavelengths=None, offset=0,
                 spectra=None):
    """ref_index may be complex (media that attenuate); spectra (W, n) with wavelengths (W, n): a polychromatic bundle; scenes that
    refract between materials (cs.materials) get the materials' m() at the rays' wavelengths, as the product hands them over."""
    scene = scene_from_compiled(cs)
    n = vertices.shape[1]
    rid = N.arange(n, dtype=N.uint64) + N.uint64(offset)
    ref = N.ones(n) if ref_index is None else N.asarray(ref_index)
    ref = ref.astype(complex) if N.iscomplexobj(ref) or getattr(cs, 'materials', None) else ref.astype(float)
    swl = spec = mat = None
    if spectra is not None:
        spec, swl = N.asarray(spectra, float), N.asarray(wavelengths, float)
        wl = N.zeros(n)
    else:
        wl = N.zeros(n) if wavelengths is None else N.asarray(wavelengths, dtype=float)
    if getattr(cs, 'materials', None):
        with N.errstate(all='ignore'):
            mat = N.array([N.asarray(m.m(wl), dtype=complex) for m in cs.materials])
    return trace(scene, N.asarray(vertices, float), N.asarray(directions, float), N.asarray(energy, float), ref, wl, rid,
                 reps, min_energy, seed, mat=mat, spec=spec, swl=swl)
