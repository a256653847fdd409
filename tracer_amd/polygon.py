"""
Flat surfaces bounded by a simple polygon, optionally with circular perforations (reference: tracer/polygon.py:8-63,
:173-198).  The boundary-crossing test runs on the GPU (trc_intersect_flat, TRC_GM_POLYGON) with the reference's
segment rules, so that points level with a vertex fall on the same side as they do there.
"""
import numpy as N

from . import _cabi
from .flat_surface import FiniteFlatGM


class FlatSimplePolygonGM(FiniteFlatGM):
    def __init__(self, profile):
        """profile: [[xs], [ys]] of the vertices of the simple polygon in sequence, CLOCKWISE, not closed."""
        self.profile = N.asarray(profile, dtype=float)
        if self.profile.ndim != 2 or self.profile.shape[0] != 2 or self.profile.shape[1] < 3:
            raise ValueError('profile is a (2, n) array of at least three vertices')
        FiniteFlatGM.__init__(self)

    def _holes(self):
        return N.zeros((0, 3))

    def _native(self):
        xs, ys = self.profile
        holes = self._holes()
        gm = [float(len(xs)), float(len(holes)), xs.min(), xs.max(), ys.min(), ys.max()]
        return _cabi.GM_POLYGON, gm, list(xs) + list(ys) + list(holes.ravel())

    def in_poly(self, points, profile):
        """boolean per column of points (2, n): inside the closed `profile` (2, m+1)?  Host copy of the device rule."""
        points, profile = N.asarray(points, dtype=float), N.asarray(profile, dtype=float)
        px, py = points[0][:, None], points[1][:, None]
        x0, y0, x1, y1 = profile[0, :-1], profile[1, :-1], profile[0, 1:], profile[1, 1:]
        xp0, xp1, yp0, yp1 = px <= x0, px <= x1, py <= y0, py <= y1
        across_y = yp0 != yp1
        with N.errstate(all='ignore'):
            slope = (y1 - y0) / (x1 - x0)
            x_cross = (py - (y0 - slope * x0)) / slope
        counted = across_y & ((xp0 & xp1) | ((xp0 != xp1) & (x_cross >= px)))
        return counted.sum(axis=1) % 2 == 1

    def mesh(self, resolution=None):
        """triangles of the polygon for rendering: ear clipping of the clockwise profile, each as a 2x2 patch (x, y, z)"""
        idx = list(range(self.profile.shape[1]))
        P = self.profile
        tris = []

        def is_ear(a, b, c, rest):
            cross = (P[0, b] - P[0, a]) * (P[1, c] - P[1, b]) - (P[1, b] - P[1, a]) * (P[0, c] - P[0, b])
            if cross > 0.:          # reflex corner of a clockwise polygon
                return False
            if not rest:
                return True
            tri = N.array([[P[0, a], P[0, b], P[0, c], P[0, a]], [P[1, a], P[1, b], P[1, c], P[1, a]]])
            return not self.in_poly(P[:, rest], tri).any()

        guard = 0
        while len(idx) > 3 and guard < 10 * P.shape[1]:
            guard += 1
            for k in range(len(idx)):
                a, b, c = idx[k - 1], idx[k], idx[(k + 1) % len(idx)]
                if is_ear(a, b, c, [i for i in idx if i not in (a, b, c)]):
                    tris.append((a, b, c))
                    idx.pop(k)
                    break
            else:
                break
        if len(idx) == 3:
            tris.append(tuple(idx))
        alpha, beta = N.meshgrid(N.linspace(0, 1, 2), N.linspace(0, 1, 2))
        out = []
        for a, b, c in tris:
            e0 = N.array([P[0, b] - P[0, a], P[1, b] - P[1, a], 0.])
            e1 = N.array([P[0, c] - P[0, a], P[1, c] - P[1, a], 0.])
            x, y, z = alpha * e1[:, None, None] * (1 - beta) + alpha * e0[:, None, None] * beta
            out += [x + P[0, a], y + P[1, a], N.zeros(x.shape)]
        return out


class PerforatedPolygonGM(FlatSimplePolygonGM):
    def __init__(self, profile, extr_centers, extr_radii):
        """extr_centers: (n, 2) centres of the circular perforations in local coordinates; extr_radii: their n radii."""
        self.extr_centers = N.asarray(extr_centers, dtype=float).reshape(-1, 2)
        self.extr_radii = N.asarray(extr_radii, dtype=float).ravel()
        if len(self.extr_radii) != len(self.extr_centers):
            raise ValueError('one radius per perforation centre')
        FlatSimplePolygonGM.__init__(self, profile)

    def _holes(self):
        return N.column_stack((self.extr_centers, self.extr_radii))
