"""
Geometry managers, restated per surface and vectorised over rays like the reference.
frame: 4x4 Surface._temp_frame; g: parameter list as in trc_surface_desc.gm; extra: table or None;
v, d: (3,N) ray vertices and directions.
"""
import numpy as N
from .kinds import *


def _local(frame, pts):
    """inv(frame) . [pts; 1] -- how every reference GM goes to local coordinates (e.g. flat_surface.py:160-164)"""
    return N.dot(N.linalg.inv(frame), N.vstack((pts, N.ones(pts.shape[1]))))[:3]


def _flat(kind, frame, g, extra, v, d):
    """flat_surface.py:33-60 plane; :203-211 rect; :266-274 extruded; :370-377 perforated; :482-492 round;
    :552-560 cut; triangular_face.py:54-74"""
    n = v.shape[1]
    vv = v - frame[:3, 3][:, None]
    dt = N.dot(d.T, frame[:3, 2])
    unparallel = N.abs(dt) > 1e-7
    t = N.full(n, N.inf)
    vt = N.dot(frame[:3, 2], vv[:, unparallel])
    t[unparallel] = -vt / dt[unparallel]
    with N.errstate(invalid='ignore'):
        t[t < 1e-7] = N.inf
    if kind == GM_FLAT_INF:
        return t
    with N.errstate(invalid='ignore', over='ignore'):
        glob = v + t[None, :] * d
        if kind == GM_TRIANGLE:
            verts = N.array([[g[0], g[3]], [g[1], g[4]], [g[2], g[5]]])
            glob_verts = N.dot(frame, N.vstack((verts, N.array([1, 1]))))
            rel_glob = glob_verts[:3].T - frame[:3, 3]
            w = glob.T - frame[:3, 3]
            uv = N.dot(verts[:, 0], verts[:, 1])
            rel_dots = N.dot(w, rel_glob.T)
            norms_sq = N.sum(verts ** 2, axis=0)
            bc = (uv * rel_dots[:, ::-1] - norms_sq[::-1] * rel_dots) / (uv ** 2 - norms_sq[0] * norms_sq[1])
            outside = N.any(bc < 0, axis=1) | (bc.sum(axis=1) > 1.)
            t[outside] = N.inf
            return t
        loc = _local(frame, glob)
        if kind in (GM_RECT, GM_RECT_EXTRUDED, GM_RECT_PERFORATED):
            half = N.c_[[g[0], g[1]]]
            t[N.any(N.abs(loc[:2]) > half, axis=0)] = N.inf
            if kind == GM_RECT_EXTRUDED:
                c = N.c_[[g[2], g[3]]]
                eh = N.c_[[g[4], g[5]]]
                t[N.all(N.abs(loc[:2] - c) < eh, axis=0)] = N.inf
            if kind == GM_RECT_PERFORATED:
                ex = N.asarray(extra).reshape(-1, 3)
                dist = N.sqrt(N.sum((loc[:2, :, None] - ex[:, :2].T[:, None, :]) ** 2, axis=0))
                t[N.any(dist < ex[:, 2], axis=1)] = N.inf
        elif kind == GM_POLYGON:
            nv, nh = int(g[0]), int(g[1])
            ex = N.asarray(extra, dtype=float)
            profile = N.vstack((ex[:nv], ex[nv:2 * nv]))
            profile = N.concatenate((profile, profile[:, 0, None]), axis=1)            # polygon.py:23
            t[~in_poly(loc[:2], profile)] = N.inf
            if nh:                                                                      # :192-195
                holes = ex[2 * nv:].reshape(-1, 3)
                dist = N.sqrt(N.sum((loc[:2, :, None] - holes[:, :2].T[:, None, :]) ** 2, axis=0))
                t[N.any(dist < holes[:, 2], axis=1)] = N.inf
        elif kind in (GM_ROUND, GM_ROUND_CUT):
            r2 = N.sum(loc[:2] ** 2., axis=0)
            t[r2 > g[0] ** 2.] = N.inf
            if g[1] >= 0:
                t[r2 < g[1] ** 2.] = N.inf
            if kind == GM_ROUND_CUT:
                t[loc[0] > g[2]] = N.inf
    return t


def _abc(kind, frame, g, v, d):
    if kind in SPHERE_KINDS:                      # sphere_surface.py:58-66 (global frame)
        c = frame[:3, 3]
        A = (d ** 2).sum(axis=0)
        B = 2. * (d * (v - c[:, None])).sum(axis=0)
        C = ((v - c[:, None]) ** 2).sum(axis=0) - g[0] ** 2
        return A, B, C
    dl = N.dot(frame[:3, :3].T, d)
    vl = _local(frame, v)
    if kind in (GM_PARABOLOID, GM_PARAB_DISH, GM_PARAB_HEX, GM_PARAB_RECT, GM_PARAB_RECT_OFFAXIS):   # paraboloid.py:39-41
        a, b = g[0], g[1]
        return (a * dl[0] ** 2 + b * dl[1] ** 2, 2 * a * dl[0] * vl[0] + 2 * b * dl[1] * vl[1] - dl[2],
                a * vl[0] ** 2 + b * vl[1] ** 2 - vl[2])
    if kind in (GM_PARAB_CYL, GM_PARAB_TROUGH):   # paraboloid.py:354-356
        a = g[0]
        return a * dl[0] ** 2, 2 * a * dl[0] * vl[0] - dl[2], a * vl[0] ** 2 - vl[2]
    if kind in (GM_CYL_INF, GM_CYL_FINITE, GM_CYL_RECTCUT):   # cylinder.py:53-55
        return (N.sum(dl[:2] ** 2, axis=0), 2. * N.sum(dl[:2] * vl[:2], axis=0), N.sum(vl[:2] ** 2, axis=0) - g[0] ** 2)
    if kind in (GM_CONE_INF, GM_CONE_FINITE, GM_FRUSTUM, GM_FRUSTUM_RECTCUT):   # cone.py:68-70
        c, a = g[0], g[1]
        return (dl[0] ** 2. + dl[1] ** 2. - (c * dl[2]) ** 2.,
                2. * (vl[0] * dl[0] + vl[1] * dl[1] - c ** 2. * (vl[2] - a) * dl[2]),
                vl[0] ** 2. + vl[1] ** 2. - (c * (vl[2] - a)) ** 2.)
    if kind in (GM_QUADRATIC, GM_QUADRATIC_RECT):   # quadratic_surface.py:57-59
        a, b, c, dd, e, f = g[:6]
        return (a * dl[0] ** 2. + b * dl[1] ** 2. + c * dl[0] * dl[1],
                2. * a * dl[0] * vl[0] + 2. * b * dl[1] * vl[1] + c * (vl[0] * dl[1] + vl[1] * dl[0]) + dd * dl[0] + e * dl[1] - dl[2],
                a * vl[0] ** 2 + b * vl[1] ** 2 + c * vl[0] * vl[1] + dd * vl[0] + e * vl[1] + f - vl[2])
    if kind in (GM_ELLIPSOID, GM_ELLIPSOID_CUT):   # ellipsoid.py:31-33
        a, b, c = g[:3]
        return (a * dl[0] ** 2 + b * dl[1] ** 2 + c * dl[2] ** 2, 2 * a * dl[0] * vl[0] + 2 * b * dl[1] * vl[1] + 2 * c * dl[2] * vl[2],
                a * vl[0] ** 2 + b * vl[1] ** 2 + c * vl[2] ** 2 - 1)
    raise ValueError(kind)


def _base_select(prm):
    """quadric.py:133-142"""
    is_positive = prm >= 1e-6
    select = N.full(prm.shape[1], N.nan)
    select[N.logical_and(*is_positive)] = 1
    one_pos = N.logical_xor(*is_positive)
    select[one_pos] = N.nonzero(is_positive.T[one_pos, :])[1]
    return select


def _own_select(hitting):
    """the `select[and]=1; select[xor]=index` idiom of e.g. paraboloid.py:114-117"""
    select = N.full(hitting.shape[1], N.nan)
    select[N.logical_and(*hitting)] = 1
    one = N.logical_xor(*hitting)
    select[one] = N.nonzero(hitting.T[one, :])[1]
    return select


def _restrict(select, inside):
    """the `select[~or]=nan; select[xor]=index` idiom of e.g. sphere_surface.py:135-137"""
    select[~N.logical_or(*inside)] = N.nan
    one = N.logical_xor(*inside)
    select[one] = N.nonzero(inside.T[one, :])[1]
    return select


def _select(kind, frame, g, coords, prm):
    """coords (2,3,n) global candidate points, prm (2,n)"""
    if kind in (GM_PARABOLOID, GM_PARAB_CYL, GM_SPHERE, GM_CYL_INF, GM_CONE_INF, GM_QUADRATIC, GM_ELLIPSOID):
        return _base_select(prm)
    n = prm.shape[1]
    loc = N.array([_local(frame, coords[0]), _local(frame, coords[1])])   # (2,3,n)
    x, y, z = loc[:, 0], loc[:, 1], loc[:, 2]
    with N.errstate(invalid='ignore'):
        if kind == GM_PARAB_DISH:                                # paraboloid.py:107-117
            return _own_select((z <= g[2]) & (z >= 0) & (prm > 1e-6))
        if kind == GM_PARAB_TROUGH:                              # :423-441
            return _own_select((N.abs(y) <= g[1]) & (z <= g[2]) & (z >= 0.) & (prm > 1e-6))
        if kind == GM_PARAB_HEX:                                 # :206-221
            outside = N.abs(x) > N.sqrt(3) * g[2] / 2.
            outside |= N.abs(y) > g[2] - N.tan(N.pi / 6.) * N.abs(x)
            return _restrict(_base_select(prm), (~outside) & (prm > 0))
        if kind in (GM_PARAB_RECT, GM_PARAB_RECT_OFFAXIS):        # :272-292
            if kind == GM_PARAB_RECT_OFFAXIS:
                rot = N.asarray(g[4:13]).reshape(3, 3)
                cen = N.asarray(g[13:16])
                loc = N.array([N.dot(rot, loc[0] + N.vstack(cen)), N.dot(rot, loc[1] + N.vstack(cen))])
                x, y = loc[:, 0], loc[:, 1]
            outside = (N.abs(x) > g[2]) | (N.abs(y) > g[3])
            return _restrict(_base_select(prm), (~outside) & (prm > 1e-6))
        if kind == GM_HEMISPHERE:                                # sphere_surface.py:128-137
            return _restrict(_base_select(prm), (z <= 0) & (prm > 1e-6))
        if kind == GM_SPHERE_CUT:                                # :188-202 (with range for xrange), bound shapes of
            rb, cb = N.asarray(g[2:11]).reshape(3, 3), N.asarray(g[11:14])   # boundary_shape.py:104-110, :139-149, :152-162
            bound = N.eye(4)
            bound[:3, :3], bound[:3, 3] = rb, cb
            temp = N.dot(frame, bound)                           # CutSphereGM.find_intersections: bound.transform_frame(frame)
            in_bd = []
            for k in range(2):
                pts = N.vstack((coords[k], N.ones(n)))
                if int(g[1]) == 1:
                    in_bd.append(N.dot(N.linalg.inv(temp)[2], pts) >= 0)
                elif int(g[1]) == 2:       # the reference keeps the sphere at its untransformed location (defect); intended form
                    in_bd.append(g[14] ** 2 >= ((coords[k] - temp[:3, 3:4]) ** 2).sum(axis=0))
                else:                      # the reference applies temp_frame[:2] instead of its inverse (defect); intended form
                    in_bd.append(N.sum(N.dot(N.linalg.inv(temp)[:2], pts) ** 2, axis=0) <= g[14] ** 2)
            return _restrict(_base_select(prm), N.array(in_bd) & (prm > 1e-6))
        if kind == GM_SPHERE_RECT:                               # :217-227
            good = (z <= 0) & (prm > 1e-6) & (N.abs(x) <= g[1]) & (N.abs(y) <= g[2])
            return _restrict(_base_select(prm), good)
        if kind == GM_CYL_FINITE:                                # cylinder.py:97-108
            angs = N.arctan2(y, x)
            angs[angs < 0] = 2 * N.pi + angs[angs < 0]
            inside = (N.abs(z) <= g[1]) & (angs >= g[2]) & (angs <= g[3])
            return _own_select(inside & (prm > 1e-6))
        if kind == GM_CYL_RECTCUT:                               # :185-197
            inside = (-g[1] <= z) & (z <= g[1]) & (N.abs(x) <= g[2]) & (N.abs(y) <= g[3])
            return _own_select(inside & (prm > 1e-6))
        if kind == GM_CONE_FINITE:                               # cone.py:111-118
            return _own_select((z >= 0) & (z <= g[2]) & (prm > 1e-9))
        if kind == GM_FRUSTUM:                                   # :311-318
            return _own_select((g[2] <= z) & (z <= g[3]) & (prm > 1e-6))
        if kind == GM_FRUSTUM_RECTCUT:                           # :381-393
            inside = (g[2] <= z) & (z <= g[3]) & (N.abs(x) <= g[4]) & (N.abs(y) <= g[5])
            return _own_select(inside & (prm > 1e-6))
        if kind == GM_QUADRATIC_RECT:                            # quadratic_surface.py:95-103
            outside = (N.abs(x) > g[6]) | (N.abs(y) > g[7])
            return _restrict(_base_select(prm), (~outside) & (prm > 1e-6))
        if kind == GM_ELLIPSOID_CUT:                             # ellipsoid.py:96-116
            ins = (x >= g[3]) & (x <= g[4]) & (y >= g[5]) & (y <= g[6]) & (z >= g[7]) & (z <= g[8])
            return _own_select(ins & (prm > 1e-7))
    raise ValueError(kind)


def _quadric(kind, frame, g, v, d):
    """quadric.py:32-113"""
    n = v.shape[1]
    A, B, C = _abc(kind, frame, g, v, d)
    delta = B ** 2. - 4. * A * C
    with N.errstate(invalid='ignore'):
        any_inters = delta >= 1e-6
    params = N.full(n, N.inf)
    num = any_inters.sum()
    if num == 0:
        return params
    A, B, C = A[any_inters], B[any_inters], C[any_inters]
    with N.errstate(invalid='ignore', divide='ignore'):
        delta = N.sqrt(B ** 2. - 4. * A * C)
        hits = N.full((2, num), N.nan)
        lin = A == 0
        bnull = B == 0
        hits[:, lin & ~bnull] = N.tile(-C[lin & ~bnull] / B[lin & ~bnull], (2, 1))
        hits[0, ~lin & bnull] = -N.sqrt(-C[~lin & bnull] / A[~lin & bnull])
        hits[1, ~lin & bnull] = N.sqrt(-C[~lin & bnull] / A[~lin & bnull])
        q = -0.5 * (B + N.sign(B) * delta)
        reg = ~lin & ~bnull
        hits[0, reg] = q[reg] / A[reg]
        hits[1, reg] = C[reg] / q[reg]
        coords = v[:, any_inters] + d[:, any_inters] * hits.reshape(2, 1, -1)
        select = _select(kind, frame, g, coords, hits)
    not_missed = ~N.isnan(select)
    any_inters[any_inters] = not_missed
    sel = N.array(select[not_missed], dtype=N.int_)
    params[any_inters] = N.choose(sel, hits[:, not_missed])
    return params


def in_poly(points, profile):
    """FlatSimplePolygonGM.in_poly (polygon.py:30-53) with `intersect` (:55-63): boundary-crossing parity; profile closed."""
    x_pos = (points[0] <= profile[0, :, None]).T
    y_pos = (points[1] <= profile[1, :, None]).T
    beyond_x = N.logical_and(x_pos[:, :-1], x_pos[:, 1:])
    across_y = N.logical_xor(y_pos[:, :-1], y_pos[:, 1:])
    inters = N.logical_and(beyond_x, across_y)
    across_x = N.logical_xor(x_pos[:, :-1], x_pos[:, 1:])
    rows, cols = N.nonzero(N.logical_and(across_x, across_y))
    with N.errstate(all='ignore'):
        x0, y0, x1, y1 = profile[0, cols], profile[1, cols], profile[0, cols + 1], profile[1, cols + 1]
        a = (y1 - y0) / (x1 - x0)
        inters[rows, cols] = (points[1, rows] - (y0 - a * x0)) / a >= points[0, rows]
    return N.array(N.sum(inters, axis=1) % 2, dtype=bool)


def intersect(kind, frame, g, extra, v, d):
    """GeometryManager.find_intersections: parametric distances, +inf = miss"""
    frame = N.asarray(frame, dtype=float)
    if kind in FLAT_KINDS:
        return _flat(kind, frame, g, extra, v, d)
    return _quadric(kind, frame, g, v, d)


def normals(kind, frame, g, hits, dirs):
    """GeometryManager.get_normals for hit points (3,H) and incident directions (3,H)"""
    frame = N.asarray(frame, dtype=float)
    H = hits.shape[1]
    if kind in FLAT_KINDS:                          # flat_surface.py:84-91 with backside = dt > 0 (:57-60)
        norms = N.tile(frame[:3, 2].copy()[:, None], (1, H))
        back = N.dot(dirs.T, frame[:3, 2]) > 0.
        norms[:, back] *= -1
        return norms
    if kind in SPHERE_KINDS:                        # sphere_surface.py:44-49
        c = frame[:3, 3]
        sides = N.sum((c - hits.T) * dirs.T, axis=1)
        normal = (hits.T - c).T.copy()
        normal[:, sides < 0.] *= -1
        return normal / N.sqrt(N.sum(normal ** 2, axis=0))
    hit = _local(frame, hits)
    dl = N.dot(frame[:3, :3].T, dirs)
    if kind in (GM_CYL_INF, GM_CYL_FINITE, GM_CYL_RECTCUT):      # cylinder.py:22-33
        ln = N.vstack((hit[:2], N.zeros(H))) / g[0]
        ln[:, N.sum(ln[:2] * dl[:2], axis=0) > 0.] *= -1.
        return N.dot(frame[:3, :3], ln)
    if kind in (GM_CONE_INF, GM_CONE_FINITE, GM_FRUSTUM, GM_FRUSTUM_RECTCUT):   # cone.py:39-57
        c, a = g[0], g[1]
        ln = N.vstack((2. * hit[0], 2. * hit[1], -2. * (hit[2] - a) * c ** 2.))
        with N.errstate(invalid='ignore', divide='ignore'):
            lu = ln / N.sqrt(N.sum(ln ** 2., axis=0))
            down = N.sum(dl * lu, axis=0) > 1e-9
        apex = hit[2] == a
        lu[:, down] *= -1.
        lu[:, apex] = N.vstack((0., 0., -1.))
        return N.dot(frame[:3, :3], lu)
    if kind in (GM_PARAB_CYL, GM_PARAB_TROUGH):                  # paraboloid.py:367-382
        ln = N.vstack((2 * hit[0] * g[0], N.zeros(H), -1 * N.ones(H)))
    elif kind in (GM_QUADRATIC, GM_QUADRATIC_RECT):              # quadratic_surface.py:32-42
        ln = N.vstack((2. * hit[0] * g[0] + g[2] * hit[1] + g[3], 2. * hit[1] * g[1] + g[2] * hit[0] + g[4], -1 * N.ones(H)))
    elif kind in (GM_ELLIPSOID, GM_ELLIPSOID_CUT):               # ellipsoid.py:49-59
        ln = N.vstack((2 * hit[0] * g[0], 2 * hit[1] * g[1], 2 * hit[2] * g[2]))
    else:                                                        # paraboloid.py:52-67
        ln = N.vstack((2 * hit[0] * g[0], 2 * hit[1] * g[1], -1 * N.ones(H)))
    lu = ln / N.sqrt(N.sum(ln ** 2, axis=0))
    down = N.sum(dl * lu, axis=0) > 0.
    lu[:, down] *= -1
    return N.dot(frame[:3, :3], lu)
