// trc_footprint.h -- footprint map of a source: where on the source's plane a fresh ray has to start to be able to
// touch a surface at all.
//
// The sources of sources.py:175-515 start their rays on a disc or a rectangle and aim them into a narrow cone about
// one direction w (pillbox: ang_range; Buie: the solar disc, 4.65 mrad -- the rare rays of the aureole, up to 43.6
// mrad, take the general path).  A ray that starts at o and reaches the point x of a surface at depth s along w has
//     o = x - s (w + tan(theta) u),   |u| = 1, u perpendicular to w, theta <= theta_c,
// i.e. o lies within margin(x) = |h| (a / (|wn| (|wn| - a)) + tan(theta_c) / (|wn| - a)) of the projection of x along w onto
// the source plane, with h = (x - c).n the height of x over the plane, wn = w.n and a = tan(theta_c) sqrt(1 - wn^2)
// (a = 0 for a source that faces its direction: margin = depth * tan(theta_c)).  The footprint of a surface is therefore
// the convex hull of the projections of the eight corners of its box, grown by the largest margin of a corner; a ray
// that starts outside every footprint cannot hit anything and is finished after its Philox block and a float32 position
// -- before its direction is sampled.  Two tables are made from the footprints, in the source's local coordinates
// (lx, ly) over [-half, half]^2:
//   mask   M x M bits: some footprint (+ cell half-diagonal + eps) overlaps the cell          (k_s_fresh stage A, in LDS)
//   lists  (M/4) x (M/4) cells: the surfaces whose footprint overlaps the cell, ascending    (stage B, candidate tests)
// eps covers the float32 evaluation of the start point in stage A (trc_fp_position32; checked on the device against
// the float64 position by tests/test_gpu_parity.py).  Conservative by construction: a set bit or a listed surface
// only costs time.  tests/test_hostcheck.py checks on the CPU that every ray of the source that hits a surface
// (brute force, float64) has its bit set and that surface listed and passing the oriented-box test.
//
// Plain C++ for the host part; the lookups are TRC_HD and shared with the kernels.
#ifndef TRC_FOOTPRINT_H
#define TRC_FOOTPRINT_H

#include "trc_core.h"

#define TRC_FP_SHIFT 2            /* a list cell is 4 x 4 mask cells */
#define TRC_FP_EPS_REL 1e-4       /* eps = TRC_FP_EPS_REL * half: float32 start points are good to ~1e-6 * half */

struct trc_fp_params {
    int32_t kind;                 // source kind (trc_source_kind)
    int32_t M, Mc;                // mask cells / list cells per side; M = Mc << TRC_FP_SHIFT, a multiple of 32
    int32_t has_generic;          // Buie: rays with u2 >= cdf_end (aureole) take the general path
    float half, inv_cell;         // the map covers [-half, half]^2; inv_cell = M / (2 half)
    float p[6];                   // start-point mapping in float32, per kind (trc_fp_position32)
    double cdf_end;
    double t_adv;                 // every ray may be advanced by this much before its float32 copy is taken: no surface is nearer
};

// float32 start point of a source ray in the source's local coordinates from its Philox block (the float64 one is
// trc_source_ray_t's).  Device: v_sqrt_f32 / v_sin_f32 / v_cos_f32 (1 ulp / ~1e-6 absolute: see TRC_FP_EPS_REL).
// KIND >= 0 promises F.kind == KIND (the kernels compile one instance per source kind).
TRC_HD float trc_fp_sqrt32(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(x);
#else
    return sqrtf(x);
#endif
}
TRC_HD void trc_fp_sincos_rev32(float rev, float *sn, float *cs) {        // sine and cosine of 2 pi rev
#if defined(__HIP_DEVICE_COMPILE__)
    *sn = __builtin_amdgcn_sinf(rev); *cs = __builtin_amdgcn_cosf(rev);
#else
    *sn = sinf(6.2831853071795865f * rev); *cs = cosf(6.2831853071795865f * rev);
#endif
}
template <int KIND>
TRC_HD void trc_fp_position32_t(const trc_fp_params &F, const uint32_t o[4], float *lx, float *ly) {
    const float s = 1.0f / 4294967296.0f;
    const int kind = KIND >= 0 ? KIND : F.kind;
    float sn, cs;
    if (kind == TRC_SRC_BUIE_DISK) {            // r = R sqrt(u0), phi = 2 pi u1
        const float u0 = ((float)o[0] + 0.5f) * s, u1 = ((float)o[1] + 0.5f) * s;
        const float r = F.p[0] * trc_fp_sqrt32(u0);
        trc_fp_sincos_rev32(u1, &sn, &cs);
        *lx = r * cs; *ly = r * sn;
    } else if (kind == TRC_SRC_BUIE_RECT) {
        const float u0 = ((float)o[0] + 0.5f) * s, u1 = ((float)o[1] + 0.5f) * s;
        *lx = F.p[0] * (u0 - 0.5f); *ly = F.p[1] * (u1 - 0.5f);
    } else if (kind == TRC_SRC_PILLBOX_DISK) {  // r = sqrt(Ri^2 + u2 (Re^2 - Ri^2)), th = span0 + (span1 - span0) u3
        const float u2 = ((float)o[2] + 0.5f) * s, u3 = ((float)o[3] + 0.5f) * s;
        const float r = trc_fp_sqrt32(F.p[0] + u2 * F.p[1]);
        trc_fp_sincos_rev32(F.p[2] + F.p[3] * u3, &sn, &cs);
        *lx = r * cs; *ly = r * sn;
    } else {                                    // TRC_SRC_PILLBOX_RECT: (lx, ly) = (ys, xs), swapped when the source says so
        const float u2 = ((float)o[2] + 0.5f) * s, u3 = ((float)o[3] + 0.5f) * s;
        const float a = F.p[0] * (u2 - 0.5f), b = F.p[1] * (u3 - 0.5f);    // xs, ys
        const bool swap = F.p[2] != 0.0f;
        *lx = swap ? a : b; *ly = swap ? b : a;
    }
}
TRC_HD void trc_fp_position32(const trc_fp_params &F, const uint32_t o[4], float *lx, float *ly) { trc_fp_position32_t<-1>(F, o, lx, ly); }

// mask cell of a start point (clamped: a point that float32 puts just outside the map belongs to its border cell)
TRC_HD void trc_fp_cell(const trc_fp_params &F, float lx, float ly, int32_t *ix, int32_t *iy) {
    int32_t x = (int32_t)floorf((lx + F.half) * F.inv_cell), y = (int32_t)floorf((ly + F.half) * F.inv_cell);
    *ix = x < 0 ? 0 : (x >= F.M ? F.M - 1 : x);
    *iy = y < 0 ? 0 : (y >= F.M ? F.M - 1 : y);
}

// does this ray take the general path (Buie aureole)?  The same comparison as trc_buie_theta_fast makes on the same value.
TRC_HD bool trc_fp_generic(const trc_fp_params &F, const uint32_t o[4]) {
    return F.has_generic && !(((double)o[2] + 0.5) * (1.0 / 4294967296.0) < F.cdf_end);
}

// ---- host side: construction (plain C++, parsed by both passes of hipcc like trc_bounds.h) ----
#include <algorithm>
#include <cmath>
#include <vector>
#include "trc_bounds.h"

struct trc_fp_host {
    bool ok;
    trc_fp_params P;
    std::vector<uint32_t> mask;       // M * M / 32 words, bit (iy * M + ix)
    std::vector<uint32_t> coff;       // Mc * Mc + 1
    std::vector<uint32_t> clist;
    double coverage;                  // fraction of mask cells set
    const char *why;                  // when !ok
};

// distance of a point to a convex polygon given counter-clockwise (0 inside); 1 or 2 vertices: point / segment
static inline double trc_fp_dist_poly(const std::vector<double> &px, const std::vector<double> &py, double x, double y) {
    const size_t n = px.size();
    if (n == 0) return INFINITY;
    if (n == 1) return std::hypot(x - px[0], y - py[0]);
    bool inside = n >= 3;
    double best = INFINITY;
    for (size_t i = 0; i < n; ++i) {
        const size_t j = (i + 1) % n;
        if (n == 2 && i == 1) break;
        const double ex = px[j] - px[i], ey = py[j] - py[i], vx = x - px[i], vy = y - py[i];
        if (ex * vy - ey * vx < 0.0) inside = false;
        const double l2 = ex * ex + ey * ey;
        double t = l2 > 0.0 ? (vx * ex + vy * ey) / l2 : 0.0;
        t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
        best = std::fmin(best, std::hypot(vx - t * ex, vy - t * ey));
    }
    return inside ? 0.0 : best;
}

// convex hull (Andrew's monotone chain), counter-clockwise, collinear points dropped
static inline void trc_fp_hull(std::vector<std::pair<double, double>> pts, std::vector<double> &hx, std::vector<double> &hy) {
    std::sort(pts.begin(), pts.end());
    pts.erase(std::unique(pts.begin(), pts.end()), pts.end());
    const size_t n = pts.size();
    hx.clear(); hy.clear();
    if (n <= 2) { for (auto &p : pts) { hx.push_back(p.first); hy.push_back(p.second); } return; }
    std::vector<std::pair<double, double>> h(2 * n);
    size_t k = 0;
    auto cross = [](const std::pair<double, double> &o, const std::pair<double, double> &a, const std::pair<double, double> &b) {
        return (a.first - o.first) * (b.second - o.second) - (a.second - o.second) * (b.first - o.first);
    };
    for (size_t i = 0; i < n; ++i) { while (k >= 2 && cross(h[k - 2], h[k - 1], pts[i]) <= 0.0) --k; h[k++] = pts[i]; }
    for (size_t i = n - 1, t = k + 1; i > 0; --i) { while (k >= t && cross(h[k - 2], h[k - 1], pts[i - 1]) <= 0.0) --k; h[k++] = pts[i - 1]; }
    h.resize(k > 1 ? k - 1 : k);
    for (auto &p : h) { hx.push_back(p.first); hy.push_back(p.second); }
}

// The source's part of the map: kind, extent of the start shape (half), float32 mapping parameters, the cone's half angle.
// Returns false (reason in *why) for a source the map does not apply to.
static inline bool trc_fp_source(const trc_source_desc &src, trc_fp_params &P, double *half_out, double *theta_c_out, const char **why) {
    memset(&P, 0, sizeof(P));
    const double *p = src.p;
    double half = 0.0, theta_c = 0.0;
    P.kind = src.kind;
    switch (src.kind) {
    case TRC_SRC_BUIE_DISK: {
        const double *sc = src.buie + 3 * (TRC_BUIE_NELEM + 1);
        half = p[0]; theta_c = sc[3];                 // theta_dni: the largest polar angle of the tabulated part
        P.has_generic = 1; P.cdf_end = src.buie[2 * (TRC_BUIE_NELEM + 1) + TRC_BUIE_NELEM];
        P.p[0] = (float)p[0];
        break;
    }
    case TRC_SRC_BUIE_RECT: {
        const double *sc = src.buie + 3 * (TRC_BUIE_NELEM + 1);
        half = 0.5 * std::fmax(std::fabs(p[0]), std::fabs(p[1])); theta_c = sc[3];
        P.has_generic = 1; P.cdf_end = src.buie[2 * (TRC_BUIE_NELEM + 1) + TRC_BUIE_NELEM];
        P.p[0] = (float)p[0]; P.p[1] = (float)p[1];
        break;
    }
    case TRC_SRC_PILLBOX_DISK:
        if (p[5] != 0.0) { *why = "disc source with x_cut (positions are redrawn)"; return false; }
        half = std::fmax(std::fabs(p[0]), std::fabs(p[1])); theta_c = p[4];
        P.p[0] = (float)(p[1] * p[1]); P.p[1] = (float)(p[0] * p[0] - p[1] * p[1]);
        P.p[2] = (float)(p[2] / TRC_TWO_PI); P.p[3] = (float)((p[3] - p[2]) / TRC_TWO_PI);
        break;
    case TRC_SRC_PILLBOX_RECT:
        half = 0.5 * std::fmax(std::fabs(p[0]), std::fabs(p[1])); theta_c = p[2];
        P.p[0] = (float)p[0]; P.p[1] = (float)p[1]; P.p[2] = p[3] != 0.0 ? 1.0f : 0.0f;
        break;
    default:
        *why = "source kind without a plane start shape"; return false;
    }
    if (!(half > 0.0) || !std::isfinite(half) || !(theta_c >= 0.0) || !(theta_c < 0.5)) { *why = "cone too wide (or degenerate source)"; return false; }
    P.half = (float)half;
    *half_out = half; *theta_c_out = theta_c;
    return true;
}

// A: trc_accel_build_surfaces of the same surfaces.  M: mask cells per side (a multiple of 32 << TRC_FP_SHIFT is not needed,
// a multiple of 32 is).
static inline void trc_fp_build(const trc_surface_desc *surfs, int n_surf, const trc_accel_host &A, const trc_source_desc &src,
                                trc_fp_host &F, int M = 512) {
    F.ok = false;
    F.why = "";
    F.mask.clear(); F.coff.clear(); F.clist.clear();
    F.coverage = 1.0;
    memset(&F.P, 0, sizeof(F.P));
    if (!A.unbounded.empty()) { F.why = "the scene has unbounded surfaces"; return; }
    if (!A.any_bounded) { F.why = "no bounded surface"; return; }
    double half = 0.0, theta_c = 0.0;
    trc_fp_params &P = F.P;
    if (!trc_fp_source(src, P, &half, &theta_c, &F.why)) return;
    const double *rp = src.rot_pos, *rd = src.rot_dir;
    const double e1[3] = {rp[0], rp[3], rp[6]}, e2[3] = {rp[1], rp[4], rp[7]}, w[3] = {rd[2], rd[5], rd[8]};
    auto dot = [](const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
    if (std::fabs(dot(e1, e1) - 1.0) > 1e-9 || std::fabs(dot(e2, e2) - 1.0) > 1e-9 || std::fabs(dot(e1, e2)) > 1e-9 ||
        std::fabs(dot(w, w) - 1.0) > 1e-9) { F.why = "start-point axes are not orthonormal"; return; }
    const double nrm[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    const double wn = dot(w, nrm);
    const double tan_c = std::tan(theta_c);
    const double a = tan_c * std::sqrt(std::fmax(0.0, 1.0 - wn * wn));
    if (!(std::fabs(wn) > 0.1) || !(a < 0.5 * std::fabs(wn))) { F.why = "source direction too oblique to its plane"; return; }
    const double K = a / (std::fabs(wn) * (std::fabs(wn) - a)) + tan_c / (std::fabs(wn) - a);
    const double eps = TRC_FP_EPS_REL * half;
    if (M < 32) M = 32;
    M = (M + 31) & ~31;
    const int Mc = M >> TRC_FP_SHIFT;
    const double cell = 2.0 * half / M, ccell = cell * (1 << TRC_FP_SHIFT);
    const double hd = 0.5 * cell * std::sqrt(2.0), chd = 0.5 * ccell * std::sqrt(2.0);
    P.M = M; P.Mc = Mc; P.half = (float)half; P.inv_cell = (float)(M / (2.0 * half));
    F.mask.assign((size_t)M * M / 32, 0u);
    std::vector<std::vector<uint32_t>> lists((size_t)Mc * Mc);
    double depth_min = INFINITY;
    std::vector<double> hx, hy;
    for (int s = 0; s < n_surf; ++s) {
        double l[3], h[3];
        bool global_axes;
        if (!trc_surface_local_box(surfs[s], l, h, &global_axes)) { F.ok = false; F.why = "unbounded surface"; return; }
        std::vector<std::pair<double, double>> pts;
        double depth_max = -INFINITY, d_lo = INFINITY;
        for (int c = 0; c < 8; ++c) {
            double q[3];
            trc_surface_box_corner(surfs[s], l, h, global_axes, c, q);
            const double v[3] = {q[0] - src.center[0], q[1] - src.center[1], q[2] - src.center[2]};
            const double depth = dot(v, nrm) / wn;                  // distance along w from the source plane to the corner
            const double f[3] = {v[0] - depth * w[0], v[1] - depth * w[1], v[2] - depth * w[2]};
            pts.emplace_back(dot(f, e1), dot(f, e2));
            depth_max = std::fmax(depth_max, depth); d_lo = std::fmin(d_lo, depth);
        }
        if (!(depth_max > 0.0)) continue;                           // wholly behind the source plane: no fresh ray reaches it
        depth_min = std::fmin(depth_min, d_lo);
        // |h| = |depth * wn| <= depth_max |wn| for the corners in front; corners behind (depth < 0) are not reached
        const double margin = depth_max * std::fabs(wn) * K + eps + 1e-9 * (half + depth_max);
        trc_fp_hull(pts, hx, hy);
        double bx0 = INFINITY, bx1 = -INFINITY, by0 = INFINITY, by1 = -INFINITY;
        for (size_t k = 0; k < hx.size(); ++k) { bx0 = std::fmin(bx0, hx[k]); bx1 = std::fmax(bx1, hx[k]); by0 = std::fmin(by0, hy[k]); by1 = std::fmax(by1, hy[k]); }
        auto range = [&](double lo, double hi, double c, int n, int *i0, int *i1) {
            double a0 = std::floor((lo + half) / c), a1 = std::floor((hi + half) / c);
            *i0 = a0 < 0.0 ? 0 : (a0 > n - 1 ? n - 1 : (int)a0);
            *i1 = a1 < 0.0 ? 0 : (a1 > n - 1 ? n - 1 : (int)a1);
            return !(hi + half < 0.0) && !(lo + half > c * n);
        };
        int x0, x1, y0, y1;
        if (range(bx0 - margin - cell, bx1 + margin + cell, cell, M, &x0, &x1) && range(by0 - margin - cell, by1 + margin + cell, cell, M, &y0, &y1))
            for (int iy = y0; iy <= y1; ++iy)
                for (int ix = x0; ix <= x1; ++ix) {
                    const double cx = -half + (ix + 0.5) * cell, cy = -half + (iy + 0.5) * cell;
                    if (trc_fp_dist_poly(hx, hy, cx, cy) <= margin + hd) F.mask[((size_t)iy * M + ix) >> 5] |= 1u << (ix & 31);
                }
        if (range(bx0 - margin - ccell, bx1 + margin + ccell, ccell, Mc, &x0, &x1) && range(by0 - margin - ccell, by1 + margin + ccell, ccell, Mc, &y0, &y1))
            for (int iy = y0; iy <= y1; ++iy)
                for (int ix = x0; ix <= x1; ++ix) {
                    const double cx = -half + (ix + 0.5) * ccell, cy = -half + (iy + 0.5) * ccell;
                    if (trc_fp_dist_poly(hx, hy, cx, cy) <= margin + chd) lists[(size_t)iy * Mc + ix].push_back((uint32_t)s);   // ascending s
                }
    }
    size_t total = 0, bits = 0;
    for (auto &v : lists) total += v.size();
    for (uint32_t wd : F.mask) bits += (size_t)__builtin_popcount(wd);
    F.coverage = (double)bits / ((double)M * M);
    F.coff.assign((size_t)Mc * Mc + 1, 0u);
    F.clist.assign(total > 0 ? total : 1, 0);
    size_t k = 0;
    for (size_t c = 0; c < lists.size(); ++c) {
        F.coff[c] = (uint32_t)k;
        for (uint32_t s : lists[c]) F.clist[k++] = s;
    }
    F.coff[lists.size()] = (uint32_t)k;
    // a ray can be advanced to just before the nearest depth of any box: a point at parameter t has depth
    // t (cos(theta) + sin(theta) (u.n) / wn) <= t (1 + a / |wn|)
    const double t_min = depth_min / (1.0 + a / std::fabs(wn));
    const double slack = 4.0 * (double)A.delta + 1e-6 * std::fabs(t_min);
    P.t_adv = (std::isfinite(t_min) && t_min - slack > 0.0) ? t_min - slack : 0.0;
    F.ok = true;
}
#endif  // TRC_FOOTPRINT_H
