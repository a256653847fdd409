// trc_bounds.h -- host-side construction of the single-precision acceleration data used by
// trc_nearest_accel32 (trc_core.h): a conservative axis-aligned box for every bounded surface, derived from
// the surface's own geometry (never from user-declared BoundaryBoxes), the scene box and centre, the
// inflation `delta`, and the packed Kd nodes.  Plain C++ so that the test-only host build can use it too.
#ifndef TRC_BOUNDS_H
#define TRC_BOUNDS_H

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>
#include "trc_core.h"

// Box of a surface in its own frame (l, h).  Returns false for unbounded kinds.  The sphere kinds work in the global
// frame (sphere_surface.py:58-66): their box is centred on the frame's origin with the GLOBAL axes (*global_axes = true).
// Every point the exact test of the kind can return lies inside the box (aperture rules of trc_core.h).
static inline bool trc_surface_local_box(const trc_surface_desc &s, double l[3], double h[3], bool *global_axes) {
    const double *g = s.gm;
    *global_axes = false;
    switch (s.gm_kind) {
    case TRC_GM_RECT: case TRC_GM_RECT_EXTRUDED: case TRC_GM_RECT_PERFORATED:
        l[0] = -g[0]; h[0] = g[0]; l[1] = -g[1]; h[1] = g[1]; l[2] = h[2] = 0.0; break;
    case TRC_GM_ROUND: case TRC_GM_ROUND_CUT:
        l[0] = l[1] = -g[0]; h[0] = h[1] = g[0]; l[2] = h[2] = 0.0; break;
    case TRC_GM_TRIANGLE:
        for (int i = 0; i < 3; ++i) { l[i] = std::fmin(0.0, std::fmin(g[i], g[3 + i])); h[i] = std::fmax(0.0, std::fmax(g[i], g[3 + i])); }
        break;
    case TRC_GM_POLYGON:
        l[0] = g[2]; h[0] = g[3]; l[1] = g[4]; h[1] = g[5]; l[2] = h[2] = 0.0; break;
    case TRC_GM_PARAB_DISH: {
        double rx = std::sqrt(g[2] / g[0]), ry = std::sqrt(g[2] / g[1]);
        l[0] = -rx; h[0] = rx; l[1] = -ry; h[1] = ry; l[2] = 0.0; h[2] = g[2]; break;
    }
    case TRC_GM_PARAB_HEX:
        l[0] = l[1] = -g[2]; h[0] = h[1] = g[2]; l[2] = 0.0; h[2] = (g[0] + g[1]) * g[2] * g[2]; break;
    case TRC_GM_PARAB_RECT:
        l[0] = -g[2]; h[0] = g[2]; l[1] = -g[3]; h[1] = g[3]; l[2] = 0.0; h[2] = g[0] * g[2] * g[2] + g[1] * g[3] * g[3]; break;
    case TRC_GM_PARAB_TROUGH: {
        double rx = std::sqrt(g[2] / g[0]);
        l[0] = -rx; h[0] = rx; l[1] = -g[1]; h[1] = g[1]; l[2] = 0.0; h[2] = g[2]; break;
    }
    case TRC_GM_SPHERE: case TRC_GM_HEMISPHERE: case TRC_GM_SPHERE_RECT: case TRC_GM_SPHERE_CUT:
        for (int i = 0; i < 3; ++i) { l[i] = -g[0]; h[i] = g[0]; }
        *global_axes = true;
        break;
    case TRC_GM_CYL_FINITE: case TRC_GM_CYL_RECTCUT:
        l[0] = l[1] = -g[0]; h[0] = h[1] = g[0]; l[2] = -g[1]; h[2] = g[1]; break;
    case TRC_GM_CONE_FINITE: {
        double r = std::fabs(g[0]) * std::fmax(std::fabs(0.0 - g[1]), std::fabs(g[2] - g[1]));
        l[0] = l[1] = -r; h[0] = h[1] = r; l[2] = 0.0; h[2] = g[2]; break;
    }
    case TRC_GM_FRUSTUM: case TRC_GM_FRUSTUM_RECTCUT: {
        double r = std::fabs(g[0]) * std::fmax(std::fabs(g[2] - g[1]), std::fabs(g[3] - g[1]));
        l[0] = l[1] = -r; h[0] = h[1] = r; l[2] = g[2]; h[2] = g[3];
        if (s.gm_kind == TRC_GM_FRUSTUM_RECTCUT) { l[0] = -std::fmin(r, g[4]); h[0] = -l[0]; l[1] = -std::fmin(r, g[5]); h[1] = -l[1]; }
        break;
    }
    case TRC_GM_QUADRATIC_RECT: {
        double w = g[6], hh = g[7];
        double m = std::fabs(g[0]) * w * w + std::fabs(g[1]) * hh * hh + std::fabs(g[2]) * w * hh + std::fabs(g[3]) * w +
                   std::fabs(g[4]) * hh + std::fabs(g[5]);
        l[0] = -w; h[0] = w; l[1] = -hh; h[1] = hh; l[2] = -m; h[2] = m; break;
    }
    case TRC_GM_ELLIPSOID: case TRC_GM_ELLIPSOID_CUT:
        for (int i = 0; i < 3; ++i) { double r = 1.0 / std::sqrt(g[i]); l[i] = -r; h[i] = r; }
        if (s.gm_kind == TRC_GM_ELLIPSOID_CUT)
            for (int i = 0; i < 3; ++i) { l[i] = std::fmax(l[i], g[3 + 2 * i]); h[i] = std::fmin(h[i], g[4 + 2 * i]); }
        break;
    default:
        return false;   // FLAT_INF, PARABOLOID, PARAB_RECT_OFFAXIS, PARAB_CYL, CYL_INF, CONE_INF, QUADRATIC
    }
    for (int i = 0; i < 3; ++i) if (!(l[i] <= h[i]) || !std::isfinite(l[i]) || !std::isfinite(h[i])) return false;
    return true;
}

// corner c (0..7) of a surface's local box in global coordinates
static inline void trc_surface_box_corner(const trc_surface_desc &s, const double l[3], const double h[3], bool global_axes, int c,
                                          double q[3]) {
    const double p[3] = {(c & 1) ? h[0] : l[0], (c & 2) ? h[1] : l[1], (c & 4) ? h[2] : l[2]};
    for (int i = 0; i < 3; ++i)
        q[i] = global_axes ? p[i] + s.frame[4 * i + 3]
                           : s.frame[4 * i] * p[0] + s.frame[4 * i + 1] * p[1] + s.frame[4 * i + 2] * p[2] + s.frame[4 * i + 3];
}

// Box of a surface in global coordinates.  Returns false for unbounded kinds.
static inline bool trc_surface_bounds(const trc_surface_desc &s, double lo[3], double hi[3]) {
    double l[3], h[3];
    bool global_axes;
    if (!trc_surface_local_box(s, l, h, &global_axes)) return false;
    for (int i = 0; i < 3; ++i) { lo[i] = std::numeric_limits<double>::infinity(); hi[i] = -lo[i]; }
    for (int c = 0; c < 8; ++c) {
        double q[3];
        trc_surface_box_corner(s, l, h, global_axes, c, q);
        for (int i = 0; i < 3; ++i) { lo[i] = std::fmin(lo[i], q[i]); hi[i] = std::fmax(hi[i], q[i]); }
    }
    return true;
}

static inline float trc_f32_down(double x) { float f = (float)x; return ((double)f > x) ? std::nextafterf(f, -INFINITY) : f; }
static inline float trc_f32_up(double x) { float f = (float)x; return ((double)f < x) ? std::nextafterf(f, INFINITY) : f; }

// Oriented box of a surface for the single-precision candidate test (trc_obb_hit32, trc_core.h): the surface's own box in
// its own frame, inflated by delta, with the rotation global -> local and the frame origin relative to the scene centre.
// For a flat surface the box is the plate itself +- delta: a ray that passes this test nearly always passes the exact one.
#define TRC_BG_ENT 12      /* floats per entry of the large grid's lists */
#define TRC_OBB_STRIDE 20   /* [A0 A1 A2 c0 | A3 A4 A5 c1 | A6 A7 A8 c2 | lo0 lo1 lo2 hi0 | hi1 hi2 - -]; A = R^T */

struct trc_accel_host {
    std::vector<float> sbox;          // 6 per surface
    std::vector<float> obb;           // TRC_OBB_STRIDE per surface (unbounded: a box nothing misses)
    std::vector<int32_t> unbounded;
    std::vector<uint32_t> nodes;      // 2 per Kd node
    std::vector<uint16_t> leaf_surfs;
    std::vector<uint16_t> brute_leaf;  // all bounded surfaces: the single leaf used when no Kd-tree is given
    uint32_t brute_nodes[2];
    float brute_root[6];
    float root[6];
    float delta;
    double cen[3], slo[3], shi[3];
    bool any_bounded;
    int kd_depth;
    // uniform grid over the scene box (trc_accel_build_grid)
    bool grid_ok;
    int32_t grid_dim[3];
    float grid_lo[3], grid_cs[3], grid_inv[3];
    std::vector<uint16_t> grid_off;    // cells + 1
    std::vector<uint16_t> grid_list;
    std::vector<int32_t> grid_apart;   // bounded surfaces kept out of the grid: their boxes are tested for every ray
    float grid_root[6];                // box of the surfaces in the grid, relative, rounded outwards
    // the same kind of grid without the limits of LDS, for scenes the one above cannot hold (trc_accel_build_grid32):
    // 32-bit offsets and lists in global memory; big_apart: the few surfaces kept out of it (as grid_apart)
    bool big_ok;
    std::vector<int32_t> big_apart;
    int32_t big_dim[3];
    float big_lo[3], big_cs[3], big_inv[3], big_root[6];
    std::vector<uint32_t> big_off, big_list;
    // what the walk reads per listed surface, in the order of big_list (TRC_BG_ENT floats each, see trc_tri_hit32 in trc_core.h):
    // a triangle with its corner and two edges, any other kind with its box -- the list entry, the box and the oriented box of a
    // candidate are three loads one behind the other otherwise; and one bit per cell that lists anything
    std::vector<float> big_ent;
    std::vector<uint32_t> big_occ;
};

// surfaces -> boxes, scene box, centre, delta
static inline void trc_accel_build_surfaces(const trc_surface_desc *surfs, int n, trc_accel_host &A) {
    const double INF = std::numeric_limits<double>::infinity();
    std::vector<double> lo(3 * (size_t)n), hi(3 * (size_t)n);
    std::vector<char> bounded(n);
    double slo[3] = {INF, INF, INF}, shi[3] = {-INF, -INF, -INF};
    A.unbounded.clear();
    A.any_bounded = false;
    for (int i = 0; i < n; ++i) {
        bounded[i] = trc_surface_bounds(surfs[i], &lo[3 * (size_t)i], &hi[3 * (size_t)i]);
        if (bounded[i]) {
            A.any_bounded = true;
            for (int k = 0; k < 3; ++k) { slo[k] = std::fmin(slo[k], lo[3 * (size_t)i + k]); shi[k] = std::fmax(shi[k], hi[3 * (size_t)i + k]); }
        } else A.unbounded.push_back(i);
    }
    if (!A.any_bounded) for (int k = 0; k < 3; ++k) { slo[k] = 0.0; shi[k] = 0.0; }
    double ext = 0.0;
    for (int k = 0; k < 3; ++k) { A.cen[k] = 0.5 * (slo[k] + shi[k]); ext = std::fmax(ext, shi[k] - slo[k]); }
    // float32 spacing at the largest relative coordinate is ~ext * 6e-8: delta is >= 400 such steps
    double delta = std::fmax(1e-3, 2.5e-5 * ext);
    A.delta = (float)delta;
    for (int k = 0; k < 3; ++k) { A.slo[k] = slo[k] - 2.0 * delta; A.shi[k] = shi[k] + 2.0 * delta; }
    A.brute_leaf.clear();
    for (int i = 0; i < n && i < 65536; ++i) if (bounded[i]) A.brute_leaf.push_back((uint16_t)i);
    A.brute_nodes[0] = 0u;
    A.brute_nodes[1] = ((uint32_t)A.brute_leaf.size() << 2) | 3u;
    for (int k = 0; k < 3; ++k) {
        A.brute_root[k] = trc_f32_down(A.slo[k] - A.cen[k]);
        A.brute_root[3 + k] = trc_f32_up(A.shi[k] - A.cen[k]);
    }
    A.grid_ok = false;
    A.big_ok = false;
    A.big_off.clear(); A.big_list.clear();
    A.sbox.assign(6 * (size_t)n, 0.0f);
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            if (bounded[i]) {
                A.sbox[6 * (size_t)i + k] = trc_f32_down(lo[3 * (size_t)i + k] - A.cen[k] - delta);
                A.sbox[6 * (size_t)i + 3 + k] = trc_f32_up(hi[3 * (size_t)i + k] - A.cen[k] + delta);
            } else {
                A.sbox[6 * (size_t)i + k] = -INFINITY;
                A.sbox[6 * (size_t)i + 3 + k] = INFINITY;
            }
        }
    A.obb.assign((size_t)TRC_OBB_STRIDE * (size_t)n, 0.0f);
    for (int i = 0; i < n; ++i) {
        float *B = &A.obb[(size_t)TRC_OBB_STRIDE * (size_t)i];
        double l[3], h[3];
        bool global_axes = false;
        const bool ok = bounded[i] && trc_surface_local_box(surfs[i], l, h, &global_axes);
        for (int r = 0; r < 3; ++r) {
            for (int k = 0; k < 3; ++k)         // row r of A = column r of the frame's rotation (local axis r in global coordinates)
                B[4 * r + k] = (ok && !global_axes) ? (float)surfs[i].frame[4 * k + r] : (r == k ? 1.0f : 0.0f);
            B[4 * r + 3] = ok ? (float)(surfs[i].frame[4 * r + 3] - A.cen[r]) : 0.0f;
        }
        for (int k = 0; k < 3; ++k) {
            // the rotation is rounded to float32 (6e-8 relative): over a box of half-size |l|, |h| that moves a point by at most
            // ~2e-7 * size, far inside delta
            const float blo = ok ? trc_f32_down(l[k] - delta) : -INFINITY, bhi = ok ? trc_f32_up(h[k] + delta) : INFINITY;
            B[12 + k] = blo;
            B[15 + k] = bhi;
        }
    }
}

// Uniform grid over the scene box for the DDA of trc_core.h (call after trc_accel_build_surfaces).  About two cells per
// bounded surface, cubic cells, at most 8192 cells and 65535 list entries (uint16 offsets, everything stays in LDS);
// the resolution is halved until that holds.  A surface is listed in every cell its box, inflated by 2*delta on top of
// the delta already in sbox, overlaps.
static inline void trc_accel_build_grid(trc_accel_host &A, int n_surf) {
    A.grid_ok = false;
    A.grid_off.clear();
    A.grid_list.clear();
    A.grid_apart.clear();
    if (A.brute_leaf.empty() || n_surf > 65535) return;
    // Surfaces that stand far from the rest (the receiver on its tower above a heliostat field) would stretch the grid over
    // mostly empty space that every ray then has to step through.  Up to 8 of them (and at most a quarter of the scene) are
    // set apart while leaving one out shrinks the box of the others by more than 30 %: their boxes are tested for every ray.
    std::vector<uint16_t> members(A.brute_leaf);
    auto box_of = [&](const std::vector<uint16_t> &m, int skip, double *blo, double *bhi) {
        for (int k = 0; k < 3; ++k) { blo[k] = INFINITY; bhi[k] = -INFINITY; }
        for (size_t j = 0; j < m.size(); ++j) {
            if ((int)j == skip) continue;
            const float *b = &A.sbox[6 * (size_t)m[j]];
            for (int k = 0; k < 3; ++k) { blo[k] = std::fmin(blo[k], (double)b[k]); bhi[k] = std::fmax(bhi[k], (double)b[3 + k]); }
        }
    };
    auto volume = [](const double *blo, const double *bhi) {
        double e[3], emax = 0.0;
        for (int k = 0; k < 3; ++k) { e[k] = bhi[k] - blo[k]; emax = std::fmax(emax, e[k]); }
        double v = 1.0;
        for (int k = 0; k < 3; ++k) v *= std::fmax(e[k], 1e-3 * emax);     // flat axes do not make the volume vanish
        return v;
    };
    // (the search is quadratic in the number of surfaces: scenes beyond a few thousand keep everything in the grid)
    while (A.grid_apart.size() < 8 && members.size() > 4 && members.size() <= 4096 && A.grid_apart.size() * 4 < A.brute_leaf.size()) {
        double blo[3], bhi[3];
        box_of(members, -1, blo, bhi);
        const double v_all = volume(blo, bhi);
        int best = -1;
        double v_best = v_all;
        for (size_t j = 0; j < members.size(); ++j) {
            const float *b = &A.sbox[6 * (size_t)members[j]];
            bool on_face = false;              // only a surface that touches a face of the box can shrink it
            for (int k = 0; k < 3; ++k) on_face = on_face || (double)b[k] <= blo[k] || (double)b[3 + k] >= bhi[k];
            if (!on_face) continue;
            double l2[3], h2[3];
            box_of(members, (int)j, l2, h2);
            const double v = volume(l2, h2);
            if (v < v_best) { v_best = v; best = (int)j; }
        }
        if (best < 0 || !(v_best < 0.7 * v_all)) break;
        A.grid_apart.push_back((int32_t)members[(size_t)best]);
        members.erase(members.begin() + best);
    }
    const size_t nb = members.size();
    double lo[3], hi[3], ext[3];
    box_of(members, -1, lo, hi);
    for (int k = 0; k < 3; ++k) {
        A.grid_root[k] = trc_f32_down(lo[k] - 2.0 * (double)A.delta);
        A.grid_root[3 + k] = trc_f32_up(hi[k] + 2.0 * (double)A.delta);
        lo[k] = (double)A.grid_root[k];
        ext[k] = (double)A.grid_root[3 + k] - lo[k];
    }
#ifndef TRC_GRID_DENSITY
#define TRC_GRID_DENSITY 2.0
#endif
    double target = std::fmin(8192.0, std::fmax(8.0, TRC_GRID_DENSITY * (double)nb));
    for (int attempt = 0; attempt < 12; ++attempt, target *= 0.5) {
        // cubic cells of volume V/target over the axes that have an extent; flat axes get one cell
        double emax = std::fmax(ext[0], std::fmax(ext[1], ext[2]));
        if (!(emax > 0.0)) return;
        double vol = 1.0;
        int nax = 0;
        for (int k = 0; k < 3; ++k) if (ext[k] > 1e-3 * emax) { vol *= ext[k]; ++nax; }
        double cell = std::pow(vol / target, 1.0 / (double)(nax > 0 ? nax : 1));
        int dim[3];
        size_t cells = 1;
        for (int k = 0; k < 3; ++k) {
            int n = (ext[k] > 1e-3 * emax) ? (int)std::ceil(ext[k] / cell) : 1;
            dim[k] = n < 1 ? 1 : (n > 128 ? 128 : n);
            cells *= (size_t)dim[k];
        }
        if (cells > 8192) continue;
        float cs[3], inv[3];
        for (int k = 0; k < 3; ++k) {
            double c = ext[k] / dim[k];
            if (!(c > 0.0)) c = 1.0;
            cs[k] = (float)c;
            inv[k] = (float)(1.0 / c);
        }
        const double pad = 2.0 * (double)A.delta;
        std::vector<uint32_t> count(cells + 1, 0u);
        std::vector<int> range(6 * nb);
        size_t total = 0;
        for (size_t j = 0; j < nb; ++j) {
            const float *b = &A.sbox[6 * (size_t)members[j]];
            size_t c = 1;
            for (int k = 0; k < 3; ++k) {
                int a = (int)std::floor(((double)b[k] - pad - lo[k]) / (double)cs[k]);
                int z = (int)std::floor(((double)b[3 + k] + pad - lo[k]) / (double)cs[k]);
                a = a < 0 ? 0 : (a >= dim[k] ? dim[k] - 1 : a);
                z = z < 0 ? 0 : (z >= dim[k] ? dim[k] - 1 : z);
                range[6 * j + k] = a; range[6 * j + 3 + k] = z;
                c *= (size_t)(z - a + 1);
            }
            total += c;
        }
        if (total > 65535) continue;
        for (size_t j = 0; j < nb; ++j)
            for (int z = range[6 * j + 2]; z <= range[6 * j + 5]; ++z)
                for (int y = range[6 * j + 1]; y <= range[6 * j + 4]; ++y)
                    for (int x = range[6 * j]; x <= range[6 * j + 3]; ++x) count[((size_t)z * dim[1] + y) * dim[0] + x + 1]++;
        for (size_t c = 0; c < cells; ++c) count[c + 1] += count[c];
        A.grid_off.assign(cells + 1, 0);
        for (size_t c = 0; c <= cells; ++c) A.grid_off[c] = (uint16_t)count[c];
        A.grid_list.assign(total > 0 ? total : 1, 0);
        std::vector<uint32_t> cur(count.begin(), count.end() - 1);
        for (size_t j = 0; j < nb; ++j)      // ascending surface index inside every cell
            for (int z = range[6 * j + 2]; z <= range[6 * j + 5]; ++z)
                for (int y = range[6 * j + 1]; y <= range[6 * j + 4]; ++y)
                    for (int x = range[6 * j]; x <= range[6 * j + 3]; ++x)
                        A.grid_list[cur[((size_t)z * dim[1] + y) * dim[0] + x]++] = members[j];
        for (int k = 0; k < 3; ++k) { A.grid_dim[k] = dim[k]; A.grid_lo[k] = (float)lo[k]; A.grid_cs[k] = cs[k]; A.grid_inv[k] = inv[k]; }
        A.grid_ok = true;
        return;
    }
}

// Does the triangle (v0, v1, v2) touch the axis-aligned box of centre c and half sides h?  Separating axes (the box's three, the
// triangle's normal, the nine cross products of edges and axes); "touch" includes contact, and 1e-12 of the sizes involved is
// allowed on every comparison so that rounding never leaves a cell out.  Used to list a triangular face only in the cells of the
// large grid it really crosses: its box is twice its size in the plane and as thick as the face is steep.
static inline bool trc_tri_box_overlap(const double c[3], const double h[3], const double v0[3], const double v1[3], const double v2[3]) {
    double a[3][3], e[3][3];
    for (int i = 0; i < 3; ++i) { a[0][i] = v0[i] - c[i]; a[1][i] = v1[i] - c[i]; a[2][i] = v2[i] - c[i]; }
    for (int i = 0; i < 3; ++i) { e[0][i] = a[1][i] - a[0][i]; e[1][i] = a[2][i] - a[1][i]; e[2][i] = a[0][i] - a[2][i]; }
    double scale = 0.0;
    for (int k = 0; k < 3; ++k) for (int i = 0; i < 3; ++i) scale = std::fmax(scale, std::fabs(a[k][i]));
    for (int i = 0; i < 3; ++i) scale = std::fmax(scale, h[i]);
    const double tol = 1e-12 * scale;
    for (int i = 0; i < 3; ++i) {       // the box's axes
        const double mn = std::fmin(a[0][i], std::fmin(a[1][i], a[2][i])), mx = std::fmax(a[0][i], std::fmax(a[1][i], a[2][i]));
        if (mn > h[i] + tol || mx < -h[i] - tol) return false;
    }
    {                                   // the triangle's normal
        const double n[3] = {e[0][1] * e[1][2] - e[0][2] * e[1][1], e[0][2] * e[1][0] - e[0][0] * e[1][2], e[0][0] * e[1][1] - e[0][1] * e[1][0]};
        const double d = n[0] * a[0][0] + n[1] * a[0][1] + n[2] * a[0][2];
        const double rr = h[0] * std::fabs(n[0]) + h[1] * std::fabs(n[1]) + h[2] * std::fabs(n[2]);
        if (std::fabs(d) > rr + tol * (std::fabs(n[0]) + std::fabs(n[1]) + std::fabs(n[2]))) return false;
    }
    for (int k = 0; k < 3; ++k)         // edge k x axis i
        for (int i = 0; i < 3; ++i) {
            const int j1 = (i + 1) % 3, j2 = (i + 2) % 3;
            // axis = unit(i) x e[k] = (.., -e[k][j2] at j1, e[k][j1] at j2)
            const double ax1 = -e[k][j2], ax2 = e[k][j1];
            double mn = INFINITY, mx = -INFINITY;
            for (int q = 0; q < 3; ++q) { const double pr = ax1 * a[q][j1] + ax2 * a[q][j2]; mn = std::fmin(mn, pr); mx = std::fmax(mx, pr); }
            const double rr = h[j1] * std::fabs(ax1) + h[j2] * std::fabs(ax2);
            if (mn > rr + tol * (std::fabs(ax1) + std::fabs(ax2)) || mx < -rr - tol * (std::fabs(ax1) + std::fabs(ax2))) return false;
        }
    return true;
}

// The grid for scenes that do not fit the one above: same construction (cells of about equal sides, three cells per
// surface, a surface listed in every cell its box inflated by 2*delta overlaps), at most 2^24 cells and 2^28 list entries, all
// bounded surfaces but the few set apart (big_apart), 32-bit offsets and lists.  Call after trc_accel_build_surfaces.
static inline void trc_accel_build_grid32(const trc_surface_desc *surfs, int n_surf, trc_accel_host &A) {
    A.big_ok = false;
    A.big_off.clear(); A.big_list.clear(); A.big_apart.clear(); A.big_ent.clear(); A.big_occ.clear();
    std::vector<uint32_t> members;
    for (int i = 0; i < n_surf; ++i) {
        const float *b = &A.sbox[6 * (size_t)i];
        if (!(b[3] == INFINITY && b[0] == -INFINITY)) members.push_back((uint32_t)i);
    }
    // Surfaces that stand far from the rest -- the lid 50 m above a mesh of 1e5 faces -- would stretch the grid over empty space
    // and leave hundreds of faces in every cell of the mesh (the 105 800-triangle relief: 280 -> ms per 2e7 rays).  As in
    // trc_accel_build_grid up to 8 of them are set apart while leaving one out shrinks the box of the others by more than 30 %;
    // here in linear time: only a surface that holds an extreme of the box on some axis can shrink it, and the box without it
    // follows from the two smallest lower and two largest upper bounds per axis.
    while (A.big_apart.size() < 8 && members.size() > 16) {
        double lo1[3], lo2[3], hi1[3], hi2[3];
        long ilo[3], ihi[3];
        for (int k = 0; k < 3; ++k) { lo1[k] = lo2[k] = INFINITY; hi1[k] = hi2[k] = -INFINITY; ilo[k] = ihi[k] = -1; }
        for (size_t j = 0; j < members.size(); ++j) {
            const float *b = &A.sbox[6 * (size_t)members[j]];
            for (int k = 0; k < 3; ++k) {
                const double l = (double)b[k], h = (double)b[3 + k];
                if (l < lo1[k]) { lo2[k] = lo1[k]; lo1[k] = l; ilo[k] = (long)j; } else if (l < lo2[k]) lo2[k] = l;
                if (h > hi1[k]) { hi2[k] = hi1[k]; hi1[k] = h; ihi[k] = (long)j; } else if (h > hi2[k]) hi2[k] = h;
            }
        }
        auto volume = [](const double *blo, const double *bhi) {
            double e[3], emax = 0.0;
            for (int k = 0; k < 3; ++k) { e[k] = bhi[k] - blo[k]; emax = std::fmax(emax, e[k]); }
            double v = 1.0;
            for (int k = 0; k < 3; ++k) v *= std::fmax(e[k], 1e-3 * emax);
            return v;
        };
        const double v_all = volume(lo1, hi1);
        long best = -1;
        double v_best = v_all;
        for (int c = 0; c < 6; ++c) {
            const long j = c < 3 ? ilo[c] : ihi[c - 3];
            if (j < 0) continue;
            double l2[3], h2[3];
            for (int k = 0; k < 3; ++k) { l2[k] = (ilo[k] == j) ? lo2[k] : lo1[k]; h2[k] = (ihi[k] == j) ? hi2[k] : hi1[k]; }
            const double v = volume(l2, h2);
            if (v < v_best) { v_best = v; best = j; }
        }
        if (best < 0 || !(v_best < 0.7 * v_all)) break;
        A.big_apart.push_back((int32_t)members[(size_t)best]);
        members.erase(members.begin() + best);
    }
    const size_t nb = members.size();
    if (nb == 0) return;
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, ext[3];
    for (size_t j = 0; j < nb; ++j) {
        const float *b = &A.sbox[6 * (size_t)members[j]];
        for (int k = 0; k < 3; ++k) { lo[k] = std::fmin(lo[k], (double)b[k]); hi[k] = std::fmax(hi[k], (double)b[3 + k]); }
    }
    for (int k = 0; k < 3; ++k) {
        A.big_root[k] = trc_f32_down(lo[k] - 2.0 * (double)A.delta);
        A.big_root[3 + k] = trc_f32_up(hi[k] + 2.0 * (double)A.delta);
        lo[k] = (double)A.big_root[k];
        ext[k] = (double)A.big_root[3 + k] - lo[k];
    }
    // cells per listed surface: 3 measured best on the relief of 105 800 triangles with k_s_bounce_coop (1: 9.4 ms per 1e7 rays,
    // 2: 9.0, 3: 8.7, 4: 8.9, 6: 9.8, 12: 10.5; with a lane per ray, k_s_bounce<2>, 6: 14.1); TRC_GRID32_DENSITY overrides it
    double density = 3.0;
    if (const char *e = std::getenv("TRC_GRID32_DENSITY")) { const double v = std::atof(e); if (v > 0.0) density = v; }
    double target = std::fmin(16777216.0, std::fmax(8.0, density * (double)nb));
    for (int attempt = 0; attempt < 16; ++attempt, target *= 0.5) {
        double emax = std::fmax(ext[0], std::fmax(ext[1], ext[2]));
        if (!(emax > 0.0)) return;
        double vol = 1.0;
        int nax = 0;
        for (int k = 0; k < 3; ++k) if (ext[k] > 1e-3 * emax) { vol *= ext[k]; ++nax; }
        double cell = std::pow(vol / target, 1.0 / (double)(nax > 0 ? nax : 1));
        int dim[3];
        size_t cells = 1;
        for (int k = 0; k < 3; ++k) {
            int n = (ext[k] > 1e-3 * emax) ? (int)std::ceil(ext[k] / cell) : 1;
            dim[k] = n < 1 ? 1 : (n > 1024 ? 1024 : n);
            cells *= (size_t)dim[k];
        }
        if (cells > 16777216) continue;
        float cs[3], inv[3];
        for (int k = 0; k < 3; ++k) {
            double c = ext[k] / dim[k];
            if (!(c > 0.0)) c = 1.0;
            cs[k] = (float)c;
            inv[k] = (float)(1.0 / c);
        }
        const double pad = 2.0 * (double)A.delta;
        std::vector<uint32_t> count(cells + 1, 0u);
        std::vector<int> range(6 * nb);
        size_t total = 0;
        for (size_t j = 0; j < nb; ++j) {
            const float *b = &A.sbox[6 * (size_t)members[j]];
            size_t c = 1;
            for (int k = 0; k < 3; ++k) {
                int a = (int)std::floor(((double)b[k] - pad - lo[k]) / (double)cs[k]);
                int z = (int)std::floor(((double)b[3 + k] + pad - lo[k]) / (double)cs[k]);
                a = a < 0 ? 0 : (a >= dim[k] ? dim[k] - 1 : a);
                z = z < 0 ? 0 : (z >= dim[k] ? dim[k] - 1 : z);
                range[6 * j + k] = a; range[6 * j + 3 + k] = z;
                c *= (size_t)(z - a + 1);
            }
            total += c;
        }
        if (total > 268435456) continue;
        // a triangular face is listed in the cells (grown by the pad) it touches, not in all those of its box
        auto listed = [&](size_t j, int x, int y, int z, const double (*tv)[3]) {
            if (!tv) return true;
            const int c3[3] = {x, y, z};
            double cc[3], hh[3];
            for (int k = 0; k < 3; ++k) { cc[k] = lo[k] + ((double)c3[k] + 0.5) * (double)cs[k]; hh[k] = 0.5 * (double)cs[k] + pad + 1e-6 * (double)cs[k]; }
            (void)j;
            return trc_tri_box_overlap(cc, hh, tv[0], tv[1], tv[2]);
        };
        std::vector<double> tri_v;           // 9 per member that is a triangle (corners relative to the scene centre), else unused
        std::vector<char> is_tri(nb, 0);
        tri_v.assign(9 * nb, 0.0);
        for (size_t j = 0; j < nb; ++j) {
            const trc_surface_desc &sd = surfs[members[j]];
            if (sd.gm_kind != TRC_GM_TRIANGLE) continue;
            is_tri[j] = 1;
            double *t = &tri_v[9 * j];
            for (int i = 0; i < 3; ++i) {
                t[i] = sd.frame[4 * i + 3] - A.cen[i];
                for (int q = 0; q < 2; ++q)
                    t[3 + 3 * q + i] = t[i] + sd.frame[4 * i] * sd.gm[3 * q] + sd.frame[4 * i + 1] * sd.gm[3 * q + 1] + sd.frame[4 * i + 2] * sd.gm[3 * q + 2];
            }
        }
        total = 0;
        for (size_t j = 0; j < nb; ++j) {
            const double (*tv)[3] = is_tri[j] ? (const double (*)[3])&tri_v[9 * j] : nullptr;
            for (int z = range[6 * j + 2]; z <= range[6 * j + 5]; ++z)
                for (int y = range[6 * j + 1]; y <= range[6 * j + 4]; ++y)
                    for (int x = range[6 * j]; x <= range[6 * j + 3]; ++x)
                        if (listed(j, x, y, z, tv)) { count[((size_t)z * dim[1] + y) * dim[0] + x + 1]++; ++total; }
        }
        for (size_t c = 0; c < cells; ++c) count[c + 1] += count[c];
        A.big_off.assign(count.begin(), count.end());
        A.big_list.assign(total > 0 ? total : 1, 0u);
        std::vector<uint32_t> cur(count.begin(), count.end() - 1);
        for (size_t j = 0; j < nb; ++j) {    // ascending surface index inside every cell
            const double (*tv)[3] = is_tri[j] ? (const double (*)[3])&tri_v[9 * j] : nullptr;
            for (int z = range[6 * j + 2]; z <= range[6 * j + 5]; ++z)
                for (int y = range[6 * j + 1]; y <= range[6 * j + 4]; ++y)
                    for (int x = range[6 * j]; x <= range[6 * j + 3]; ++x)
                        if (listed(j, x, y, z, tv)) A.big_list[cur[((size_t)z * dim[1] + y) * dim[0] + x]++] = members[j];
        }
        for (int k = 0; k < 3; ++k) { A.big_dim[k] = dim[k]; A.big_lo[k] = (float)lo[k]; A.big_cs[k] = cs[k]; A.big_inv[k] = inv[k]; }
        // the entries of the walk: per member once, then copied in list order
        std::vector<float> ent_of((size_t)TRC_BG_ENT * nb, 0.0f);
        std::vector<uint32_t> member_no((size_t)n_surf, 0u);
        for (size_t j = 0; j < nb; ++j) {
            const uint32_t si = members[j];
            member_no[si] = (uint32_t)j;
            float *e = &ent_of[(size_t)TRC_BG_ENT * j];
            const trc_surface_desc &sd = surfs[si];
            uint32_t w = si;
            if (sd.gm_kind == TRC_GM_TRIANGLE) {
                // corner (the frame's origin) relative to the scene centre, the two edges in global axes; [7] = 1.5 delta max|edge|,
                // [11] = max|edge|^2, both rounded up
                double ed[2][3], emax = 0.0;
                for (int q = 0; q < 2; ++q) {
                    double n2 = 0.0;
                    for (int i = 0; i < 3; ++i) {
                        ed[q][i] = sd.frame[4 * i] * sd.gm[3 * q] + sd.frame[4 * i + 1] * sd.gm[3 * q + 1] + sd.frame[4 * i + 2] * sd.gm[3 * q + 2];
                        n2 += ed[q][i] * ed[q][i];
                    }
                    emax = std::fmax(emax, std::sqrt(n2));
                }
                for (int i = 0; i < 3; ++i) {
                    e[1 + i] = (float)(sd.frame[4 * i + 3] - A.cen[i]);
                    e[4 + i] = (float)ed[0][i];
                    e[8 + i] = (float)ed[1][i];
                }
                e[7] = trc_f32_up(1.5 * (double)A.delta * emax * 1.000001);
                e[11] = trc_f32_up(emax * emax * 1.000001);
            } else {
                w |= 0x80000000u;        // its box here, its oriented box from the table
                const float *b = &A.sbox[6 * (size_t)si];
                for (int i = 0; i < 3; ++i) { e[1 + i] = b[i]; e[4 + i] = b[3 + i]; }
            }
            std::memcpy(&e[0], &w, 4);
        }
        A.big_ent.assign((size_t)TRC_BG_ENT * (total > 0 ? total : 1), 0.0f);
        for (size_t i = 0; i < total; ++i)
            std::memcpy(&A.big_ent[(size_t)TRC_BG_ENT * i], &ent_of[(size_t)TRC_BG_ENT * member_no[A.big_list[i]]], TRC_BG_ENT * sizeof(float));
        A.big_occ.assign((cells + 31) / 32 + 1, 0u);
        for (size_t c = 0; c < cells; ++c) if (A.big_off[c + 1] > A.big_off[c]) A.big_occ[c >> 5] |= 1u << (c & 31);
        A.big_ok = true;
        return;
    }
}

// Kd-tree -> packed nodes relative to the centre (call after trc_accel_build_surfaces). Returns false when the
// tree cannot use the packed form (surface index or node count too large).
static inline bool trc_accel_build_kd(const trc_kdtree_desc *kd, trc_accel_host &A) {
    if (kd->n_nodes >= (1 << 28)) return false;
    A.nodes.assign(2 * (size_t)kd->n_nodes, 0u);
    A.leaf_surfs.assign((size_t)kd->n_leaf_surfs, 0);
    for (int i = 0; i < kd->n_leaf_surfs; ++i) {
        if (kd->leaf_surfs[i] > 65535) return false;
        A.leaf_surfs[i] = (uint16_t)kd->leaf_surfs[i];
    }
    std::vector<int> depth(kd->n_nodes, 0);
    A.kd_depth = 0;
    for (int i = 0; i < kd->n_nodes; ++i) {
        int f = kd->flag[i];
        if (f == 3) {
            A.nodes[2 * (size_t)i] = (uint32_t)kd->leaf_off[i];
            A.nodes[2 * (size_t)i + 1] = ((uint32_t)kd->leaf_cnt[i] << 2) | 3u;
        } else {
            float sp = (float)(kd->split[i] - A.cen[f]);
            uint32_t w;
            std::memcpy(&w, &sp, 4);
            A.nodes[2 * (size_t)i] = w;
            A.nodes[2 * (size_t)i + 1] = ((uint32_t)kd->child[i] << 2) | (uint32_t)f;
            int c = kd->child[i];
            depth[c] = depth[c + 1] = depth[i] + 1;
            if (depth[c] > A.kd_depth) A.kd_depth = depth[c];
        }
    }
    for (int k = 0; k < 3; ++k) {
        A.root[k] = trc_f32_down(kd->bounds[k] - A.cen[k] - A.delta);
        A.root[3 + k] = trc_f32_up(kd->bounds[3 + k] - A.cen[k] + A.delta);
    }
    return true;
}

#endif  // TRC_BOUNDS_H
