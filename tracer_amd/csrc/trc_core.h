// trc_core.h -- per-ray math of the tracing core (one lane = one ray).
//
// Everything here is a pure function of its arguments, marked TRC_HD so that the SAME source
// is compiled by hipcc for gfx950 (the product, trc_kernels.hip) and by g++ for the host-side
// debug harness under tests/hostcheck (never shipped, never loaded by the package).
//
// The formulation is per ray, not per surface as in the reference: a lane walks the candidate
// surfaces with the ray in registers.  Each function cites the reference lines whose results it
// must reproduce (paths relative to the reference tree).
#ifndef TRC_CORE_H
#define TRC_CORE_H

#include <stdint.h>
#include <math.h>
#include "../../include/tracer_amd.h"

#if defined(__HIPCC__)
#define TRC_HD __host__ __device__ __forceinline__
#define TRC_HD_NOINLINE __host__ __device__
#else
#define TRC_HD static inline
#define TRC_HD_NOINLINE static
#endif

#define TRC_INF (__builtin_inf())
#define TRC_INF32 (__builtin_inff())
#define TRC_NAN (__builtin_nan(""))
#define TRC_TWO_PI 6.283185307179586476925286766559
#define TRC_PI 3.14159265358979323846264338327950288
#define TRC_BUIE_TABLE (3 * (TRC_BUIE_NELEM + 1) + 6)     /* doubles in trc_source_desc.buie */
#define TRC_BUIE_STAGED (TRC_BUIE_TABLE + 3)              /* + trc_buie_aureole_consts, in the kernels' LDS copies */

// ---------------------------------------------------------------------------------------------
// Compact per-surface record used by the kernels (LDS-staged).  Layout in doubles:
//   [0..8]  R row-major (local->global rotation)     [9..11] c (origin)
//   [12]    int32 gm_kind | int32 optics_kind        [13]    int32 extra_off | int32 extra_len
//   [14..]  geometry parameters (trc_surface_desc.gm)
// The record stride is chosen per scene (14 + max params used, made odd to spread LDS banks).
// ---------------------------------------------------------------------------------------------
#define TRC_REC_HDR 14

TRC_HD int trc_rec_gm_kind(const double *rec) { return ((const int32_t *)(rec + 12))[0]; }
TRC_HD int trc_rec_opt_kind(const double *rec) { return ((const int32_t *)(rec + 12))[1]; }
TRC_HD int trc_rec_extra_off(const double *rec) { return ((const int32_t *)(rec + 13))[0]; }
TRC_HD int trc_rec_extra_len(const double *rec) { return ((const int32_t *)(rec + 13))[1]; }

// number of geometry parameters each kind reads (host side uses it to size the record)
TRC_HD int trc_gm_nparams(int kind) {
    switch (kind) {
    case TRC_GM_FLAT_INF: return 0;
    case TRC_GM_RECT: return 2;
    case TRC_GM_RECT_EXTRUDED: return 6;
    case TRC_GM_RECT_PERFORATED: return 2;
    case TRC_GM_ROUND: return 2;
    case TRC_GM_ROUND_CUT: return 3;
    case TRC_GM_TRIANGLE: return 6;
    case TRC_GM_PARABOLOID: return 2;
    case TRC_GM_PARAB_DISH: return 3;
    case TRC_GM_PARAB_HEX: return 3;
    case TRC_GM_PARAB_RECT: return 4;
    case TRC_GM_PARAB_RECT_OFFAXIS: return 16;
    case TRC_GM_PARAB_CYL: return 1;
    case TRC_GM_PARAB_TROUGH: return 3;
    case TRC_GM_SPHERE: return 1;
    case TRC_GM_HEMISPHERE: return 1;
    case TRC_GM_SPHERE_RECT: return 3;
    case TRC_GM_CYL_INF: return 1;
    case TRC_GM_CYL_FINITE: return 4;
    case TRC_GM_CYL_RECTCUT: return 4;
    case TRC_GM_CONE_INF: return 2;
    case TRC_GM_CONE_FINITE: return 3;
    case TRC_GM_FRUSTUM: return 4;
    case TRC_GM_FRUSTUM_RECTCUT: return 6;
    case TRC_GM_QUADRATIC: return 6;
    case TRC_GM_QUADRATIC_RECT: return 8;
    case TRC_GM_POLYGON: return 6;
    case TRC_GM_ELLIPSOID: return 3;
    case TRC_GM_ELLIPSOID_CUT: return 9;
    case TRC_GM_SPHERE_CUT: return 15;
    default: return -1;
    }
}

// ---------------------------------------------------------------------------------------------
// Random numbers: Philox4x32-10 (Salmon et al., SC'11), counter-based so that every draw is a
// pure function of (seed, ray stream id, event index, block) -- results do not depend on how
// rays are scheduled on waves, on the engine used, or on the number of GPUs.
//   counter = (rid_lo, rid_hi, event, block)   key = (seed_lo, seed_hi)
//   event 0 = source generation; event k>=1 = the k-th surface interaction of the ray.
// ---------------------------------------------------------------------------------------------
TRC_HD void trc_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                              uint32_t k1, uint32_t out[4]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)M0 * c0;
        uint64_t p1 = (uint64_t)M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// two 32-bit words -> double in [0,1) with 53 random bits
TRC_HD double trc_u01_from_bits(uint32_t a, uint32_t b) {
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

// uniforms (2k, 2k+1) of the event's stream live in Philox block k
TRC_HD void trc_uniform_pair(uint64_t seed, uint64_t rid, uint32_t event, uint32_t block, double *u0,
                             double *u1) {
    uint32_t o[4];
    trc_philox4x32_10((uint32_t)rid, (uint32_t)(rid >> 32), event, block, (uint32_t)seed,
                      (uint32_t)(seed >> 32), o);
    *u0 = trc_u01_from_bits(o[0], o[1]);
    *u1 = trc_u01_from_bits(o[2], o[3]);
}

// sine and cosine of the same angle with one argument reduction
TRC_HD void trc_sincos(double x, double *s, double *c) { sincos(x, s, c); }

// sin and cos of 2*pi*u for u in [0,1): on the device through sincospi (exact argument reduction, cheaper than reducing
// the rounded product 2*pi*u); the two forms agree to a few units of the last place
TRC_HD void trc_sincos_2pi(double u, double *s, double *c) {
#if defined(__HIP_DEVICE_COMPILE__)
    sincospi(2.0 * u, s, c);
#else
    sincos(TRC_TWO_PI * u, s, c);
#endif
}

// sin and cos of a small angle (|x| <= 0.05: the Buie polar angle is below 43.6 mrad) by their Taylor series: the
// first neglected terms are x^11/11! and x^12/12!, below 1e-21 relative -- full double precision without range reduction
TRC_HD void trc_sincos_small(double x, double *s, double *c) {
    double x2 = x * x;
    *s = x * (1.0 + x2 * (-1.0 / 6.0 + x2 * (1.0 / 120.0 + x2 * (-1.0 / 5040.0 + x2 * (1.0 / 362880.0)))));
    *c = 1.0 + x2 * (-0.5 + x2 * (1.0 / 24.0 + x2 * (-1.0 / 720.0 + x2 * (1.0 / 40320.0 + x2 * (-1.0 / 3628800.0)))));
}

// four uniforms in (0,1) with 32 random bits each from ONE Philox block: the draws of a source ray (position and
// direction samples are resolved to 2^-32 -- 38 nm on the 163 m NSTTF source disc -- and cost half the generator work of
// two 53-bit pairs; the 40 v_mad_u64_u32 of two blocks were a quarter of the cycles of the generation kernel)
TRC_HD void trc_uniform_quad(uint64_t seed, uint64_t rid, uint32_t event, uint32_t block, double *u0, double *u1,
                             double *u2, double *u3) {
    uint32_t o[4];
    trc_philox4x32_10((uint32_t)rid, (uint32_t)(rid >> 32), event, block, (uint32_t)seed, (uint32_t)(seed >> 32), o);
    const double s = 1.0 / 4294967296.0;
    *u0 = ((double)o[0] + 0.5) * s;
    *u1 = ((double)o[1] + 0.5) * s;
    *u2 = ((double)o[2] + 0.5) * s;
    *u3 = ((double)o[3] + 0.5) * s;
}

// Box-Muller: two independent N(0,1) from two uniforms (1-u0 keeps the log argument in (0,1])
TRC_HD void trc_normal_pair(double u0, double u1, double *g0, double *g1) {
    double r = sqrt(-2.0 * log(1.0 - u0));
    double sa, ca;
    trc_sincos_2pi(u1, &sa, &ca);           // (sine and cosine of 2 pi u1 with the exact reduction of sincospi: a third of sincos(2 pi u1))
    *g0 = r * ca;
    *g1 = r * sa;
}

// Tangent of a small angle by its Maclaurin series (the slope errors of mirrors are milliradians): below 1/8 the first term left
// out, 6404582 / 10854718875 x^17, is under 3e-19 of x; beyond, the library's tangent with its argument reduction
TRC_HD double trc_tan_small(double x) {
    if (!(fabs(x) <= 0.125)) return tan(x);
    const double x2 = x * x;
    double p = 929569.0 / 638512875.0;
    p = p * x2 + 21844.0 / 6081075.0;
    p = p * x2 + 1382.0 / 155925.0;
    p = p * x2 + 62.0 / 2835.0;
    p = p * x2 + 17.0 / 315.0;
    p = p * x2 + 2.0 / 15.0;
    p = p * x2 + 1.0 / 3.0;
    return x + x * (x2 * p);
}

// sine and cosine of an angle that is usually small (the polar angle of a slope error): the series of trc_sincos_small up to
// 0.05 rad, the library beyond
TRC_HD void trc_sincos_mostly_small(double x, double *s, double *c) {
    if (fabs(x) <= 0.05) trc_sincos_small(x, s, c);
    else trc_sincos(x, s, c);
}

// stream id of the second ray created when an interaction splits a ray in two
// (RefractiveHomogenous, single_ray=False: optics_callables.py:1284-1294)
TRC_HD uint64_t trc_child_rid(uint64_t rid, uint32_t event) {
    return rid * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull * (uint64_t)(event + 1u);
}

// ---------------------------------------------------------------------------------------------
// small vector helpers
// ---------------------------------------------------------------------------------------------
TRC_HD double trc_dot3(double ax, double ay, double az, double bx, double by, double bz) {
    return ax * bx + ay * by + az * bz;
}

// global -> local: R^T (p - c)   (the reference multiplies by linalg.inv(frame); frames are rigid)
TRC_HD void trc_to_local_point(const double *rec, double px, double py, double pz, double *lx,
                               double *ly, double *lz) {
    double qx = px - rec[9], qy = py - rec[10], qz = pz - rec[11];
    *lx = rec[0] * qx + rec[3] * qy + rec[6] * qz;
    *ly = rec[1] * qx + rec[4] * qy + rec[7] * qz;
    *lz = rec[2] * qx + rec[5] * qy + rec[8] * qz;
}
TRC_HD void trc_to_local_dir(const double *rec, double dx, double dy, double dz, double *lx, double *ly,
                             double *lz) {
    *lx = rec[0] * dx + rec[3] * dy + rec[6] * dz;
    *ly = rec[1] * dx + rec[4] * dy + rec[7] * dz;
    *lz = rec[2] * dx + rec[5] * dy + rec[8] * dz;
}
TRC_HD void trc_to_global_dir(const double *rec, double lx, double ly, double lz, double *gx, double *gy,
                              double *gz) {
    *gx = rec[0] * lx + rec[1] * ly + rec[2] * lz;
    *gy = rec[3] * lx + rec[4] * ly + rec[5] * lz;
    *gz = rec[6] * lx + rec[7] * ly + rec[8] * lz;
}

// ---------------------------------------------------------------------------------------------
// G1/G2/G3 -- flat family.  flat_surface.py:33-60 (plane), :160-164 (local coords),
// :209 rect, :271-272 extruded, :373-375 perforated, :488-490 round, :558 cut,
// triangular_face.py:59-72 triangle.
// ---------------------------------------------------------------------------------------------
TRC_HD double trc_intersect_flat(int kind, const double *rec, const double *extra, double vx, double vy,
                                 double vz, double dx, double dy, double dz) {
    const double nx = rec[2], ny = rec[5], nz = rec[8];
    double dt = dx * nx + dy * ny + dz * nz;
    if (!(fabs(dt) > 1e-7)) return TRC_INF;                    // flat_surface.py:39
    double vt = nx * (vx - rec[9]) + ny * (vy - rec[10]) + nz * (vz - rec[11]);
    double t = -vt / dt;                                       // :47
    if (!(t >= 1e-7)) return TRC_INF;                          // :50-51 (params < 1e-7 -> inf)
    if (kind == TRC_GM_FLAT_INF) return t;
    double hx = vx + t * dx, hy = vy + t * dy, hz = vz + t * dz;   // :156
    const double *g = rec + TRC_REC_HDR;
    if (kind == TRC_GM_TRIANGLE) {
        // barycentric coordinates from Gram terms, triangular_face.py:59-72
        double e0x, e0y, e0z, e1x, e1y, e1z;
        trc_to_global_dir(rec, g[0], g[1], g[2], &e0x, &e0y, &e0z);
        trc_to_global_dir(rec, g[3], g[4], g[5], &e1x, &e1y, &e1z);
        double wx = hx - rec[9], wy = hy - rec[10], wz = hz - rec[11];
        double uv = g[0] * g[3] + g[1] * g[4] + g[2] * g[5];
        double n0 = g[0] * g[0] + g[1] * g[1] + g[2] * g[2];
        double n1 = g[3] * g[3] + g[4] * g[4] + g[5] * g[5];
        double r0 = wx * e0x + wy * e0y + wz * e0z;
        double r1 = wx * e1x + wy * e1y + wz * e1z;
        double den = uv * uv - n0 * n1;
        double bc0 = (uv * r1 - n1 * r0) / den;
        double bc1 = (uv * r0 - n0 * r1) / den;
        if (bc0 < 0.0 || bc1 < 0.0 || (bc0 + bc1) > 1.0) return TRC_INF;
        return t;
    }
    double lx, ly, lz;
    trc_to_local_point(rec, hx, hy, hz, &lx, &ly, &lz);
    (void)lz;
    switch (kind) {
    case TRC_GM_RECT:
        if (fabs(lx) > g[0] || fabs(ly) > g[1]) return TRC_INF;
        return t;
    case TRC_GM_RECT_EXTRUDED:
        if (fabs(lx) > g[0] || fabs(ly) > g[1]) return TRC_INF;
        if (fabs(lx - g[2]) < g[4] && fabs(ly - g[3]) < g[5]) return TRC_INF;
        return t;
    case TRC_GM_RECT_PERFORATED: {
        if (fabs(lx) > g[0] || fabs(ly) > g[1]) return TRC_INF;
        int off = trc_rec_extra_off(rec), len = trc_rec_extra_len(rec);
        for (int k = 0; k + 2 < len; k += 3) {
            double ex = lx - extra[off + k], ey = ly - extra[off + k + 1];
            if (sqrt(ex * ex + ey * ey) < extra[off + k + 2]) return TRC_INF;
        }
        return t;
    }
    case TRC_GM_ROUND:
    case TRC_GM_ROUND_CUT: {
        double r2 = lx * lx + ly * ly;
        if (r2 > g[0] * g[0]) return TRC_INF;
        if (g[1] >= 0.0 && r2 < g[1] * g[1]) return TRC_INF;
        if (kind == TRC_GM_ROUND_CUT && lx > g[2]) return TRC_INF;
        return t;
    }
    case TRC_GM_POLYGON: {
        // boundary-crossing count with the reference's segment rules (polygon.py:30-63): a segment wholly at x >= the point
        // counts when it straddles the point's y; one that straddles both x and y counts when its crossing abscissa
        // is >= the point's x; the rest do not.  "<=" puts a point level with a vertex on the vertex's lower-left side.
        const int n = (int)g[0], nh = (int)g[1];
        const double *xs = extra + trc_rec_extra_off(rec), *ys = xs + n;
        unsigned crossings = 0;
        for (int k = 0; k < n; ++k) {
            const int k1 = (k + 1 == n) ? 0 : k + 1;
            const double x0 = xs[k], y0 = ys[k], x1 = xs[k1], y1 = ys[k1];
            const bool xp0 = lx <= x0, xp1 = lx <= x1, yp0 = ly <= y0, yp1 = ly <= y1;
            if (yp0 == yp1) continue;
            if (xp0 && xp1) crossings += 1u;
            else if (xp0 != xp1) {
                const double a = (y1 - y0) / (x1 - x0);              // :57-61
                const double x_inter = (ly - (y0 - a * x0)) / a;
                if (x_inter >= lx) crossings += 1u;
            }
        }
        if (!(crossings & 1u)) return TRC_INF;
        const double *hole = ys + n;                                  // circular perforations, :192-195
        for (int k = 0; k < nh; ++k) {
            double ex = lx - hole[3 * k], ey = ly - hole[3 * k + 1];
            if (sqrt(ex * ex + ey * ey) < hole[3 * k + 2]) return TRC_INF;
        }
        return t;
    }
    default:
        return TRC_INF;
    }
}

// ---------------------------------------------------------------------------------------------
// G4..G9 -- quadric family.  quadric.py:54-101 (solver), :133-142 (default root choice) and each
// subclass' get_ABC / _select_coords.
// ---------------------------------------------------------------------------------------------
TRC_HD bool trc_quadric_aperture(int kind, const double *rec, const double *g, double lx, double ly,
                                 double lz) {
    switch (kind) {
    case TRC_GM_PARAB_DISH: return (lz <= g[2]) && (lz >= 0.0);                         // paraboloid.py:112
    case TRC_GM_PARAB_HEX: {                                                            // :213-216
        double ax = fabs(lx), ay = fabs(ly);
        bool outside = ax > sqrt(3.0) * g[2] / 2.0;
        outside = outside || (ay > g[2] - tan(3.14159265358979323846 / 6.0) * ax);
        return !outside;
    }
    case TRC_GM_PARAB_RECT: return !(fabs(lx) > g[2] || fabs(ly) > g[3]);               // :283-288
    case TRC_GM_PARAB_RECT_OFFAXIS: {                                                   // :279-281
        double qx = lx + g[13], qy = ly + g[14], qz = lz + g[15];
        double rx = g[4] * qx + g[5] * qy + g[6] * qz;
        double ry = g[7] * qx + g[8] * qy + g[9] * qz;
        return !(fabs(rx) > g[2] || fabs(ry) > g[3]);
    }
    case TRC_GM_PARAB_TROUGH: return (fabs(ly) <= g[1]) && (lz <= g[2]) && (lz >= 0.0); // :433-438
    case TRC_GM_HEMISPHERE: return lz <= 0.0;                                            // sphere_surface.py:133
    case TRC_GM_SPHERE_RECT: return (lz <= 0.0) && (fabs(lx) <= g[1]) && (fabs(ly) <= g[2]); // :222-223
    case TRC_GM_SPHERE_CUT: {                 // :194-199 with the bound carried in the surface's frame
        double qx = lx - g[11], qy = ly - g[12], qz = lz - g[13];
        int bound = (int)g[1];
        if (bound == 1) return (g[4] * qx + g[7] * qy + g[10] * qz) >= 0.0;             // boundary_shape.py:152-162
        if (bound == 2) return g[14] * g[14] >= qx * qx + qy * qy + qz * qz;             // :104-110
        if (bound == 3) {                                                                // :139-149
            double bx = g[2] * qx + g[5] * qy + g[8] * qz, by = g[3] * qx + g[6] * qy + g[9] * qz;
            return bx * bx + by * by <= g[14] * g[14];
        }
        return true;
    }
    case TRC_GM_CYL_FINITE: {                                                           // cylinder.py:97-103
        // a full turn of wall: the azimuth, brought to [0, 2 pi), passes whatever it is (and a nan hit point fails on its height
        // as it would on its angle) -- no arc tangent, twice per ray and cylinder
        if (g[2] <= 0.0 && g[3] >= TRC_TWO_PI) return fabs(lz) <= g[1];
        double ang = atan2(ly, lx);
        if (ang < 0.0) ang = TRC_TWO_PI + ang;
        return (fabs(lz) <= g[1]) && (ang >= g[2]) && (ang <= g[3]);
    }
    case TRC_GM_CYL_RECTCUT:                                                            // :185-190
        return (-g[1] <= lz) && (lz <= g[1]) && (fabs(lx) <= g[2]) && (fabs(ly) <= g[3]);
    case TRC_GM_CONE_FINITE: return (lz >= 0.0) && (lz <= g[2]);                          // cone.py:111
    case TRC_GM_FRUSTUM: return (g[2] <= lz) && (lz <= g[3]);                            // :311
    case TRC_GM_FRUSTUM_RECTCUT:                                                        // :381-386
        return (g[2] <= lz) && (lz <= g[3]) && (fabs(lx) <= g[4]) && (fabs(ly) <= g[5]);
    case TRC_GM_QUADRATIC_RECT: return !(fabs(lx) > g[6] || fabs(ly) > g[7]);            // quadratic_surface.py:95-98
    case TRC_GM_ELLIPSOID_CUT:                                                          // ellipsoid.py:102-110
        return (lx >= g[3]) && (lx <= g[4]) && (ly >= g[5]) && (ly <= g[6]) && (lz >= g[7]) && (lz <= g[8]);
    default: return true;
    }
    (void)rec;
}

// how a kind picks between the two roots
//   0: base class rule only (valid_k = t_k >= 1e-6)                      quadric.py:133-142
//   1: own rule: valid_k = aperture_k && t_k > eps                       e.g. paraboloid.py:104-117
//   2: base rule first, then restricted by inside_k = aperture_k && t_k > eps (xor -> that one,
//      neither -> miss, both -> keep the base choice)                    e.g. sphere_surface.py:128-137
TRC_HD int trc_quadric_select_mode(int kind, double *eps) {
    switch (kind) {
    case TRC_GM_PARAB_DISH: case TRC_GM_PARAB_TROUGH: case TRC_GM_CYL_FINITE: case TRC_GM_CYL_RECTCUT:
    case TRC_GM_FRUSTUM: case TRC_GM_FRUSTUM_RECTCUT:
        *eps = 1e-6; return 1;
    case TRC_GM_CONE_FINITE: *eps = 1e-9; return 1;            // cone.py:112
    case TRC_GM_ELLIPSOID_CUT: *eps = 1e-7; return 1;          // ellipsoid.py:96
    case TRC_GM_PARAB_HEX: *eps = 0.0; return 2;               // paraboloid.py:217
    case TRC_GM_PARAB_RECT: case TRC_GM_PARAB_RECT_OFFAXIS: case TRC_GM_HEMISPHERE: case TRC_GM_SPHERE_RECT:
    case TRC_GM_QUADRATIC_RECT: case TRC_GM_SPHERE_CUT:
        *eps = 1e-6; return 2;
    default: *eps = 1e-6; return 0;
    }
}

TRC_HD double trc_intersect_quadric(int kind, const double *rec, double vx, double vy, double vz, double dx,
                                    double dy, double dz) {
    const double *g = rec + TRC_REC_HDR;
    double A, B, C;
    if (kind == TRC_GM_SPHERE || kind == TRC_GM_HEMISPHERE || kind == TRC_GM_SPHERE_RECT || kind == TRC_GM_SPHERE_CUT) {
        // global frame, sphere_surface.py:58-66
        double qx = vx - rec[9], qy = vy - rec[10], qz = vz - rec[11];
        A = dx * dx + dy * dy + dz * dz;
        B = 2.0 * (dx * qx + dy * qy + dz * qz);
        C = (qx * qx + qy * qy + qz * qz) - g[0] * g[0];
    } else {
        double lx, ly, lz, ex, ey, ez;
        trc_to_local_point(rec, vx, vy, vz, &lx, &ly, &lz);
        trc_to_local_dir(rec, dx, dy, dz, &ex, &ey, &ez);
        switch (kind) {
        case TRC_GM_PARABOLOID: case TRC_GM_PARAB_DISH: case TRC_GM_PARAB_HEX: case TRC_GM_PARAB_RECT:
        case TRC_GM_PARAB_RECT_OFFAXIS:                           // paraboloid.py:39-41
            A = g[0] * ex * ex + g[1] * ey * ey;
            B = 2.0 * g[0] * ex * lx + 2.0 * g[1] * ey * ly - ez;
            C = g[0] * lx * lx + g[1] * ly * ly - lz;
            break;
        case TRC_GM_PARAB_CYL: case TRC_GM_PARAB_TROUGH:           // paraboloid.py:354-356
            A = g[0] * ex * ex;
            B = 2.0 * g[0] * ex * lx - ez;
            C = g[0] * lx * lx - lz;
            break;
        case TRC_GM_CYL_INF: case TRC_GM_CYL_FINITE: case TRC_GM_CYL_RECTCUT:   // cylinder.py:53-55
            A = ex * ex + ey * ey;
            B = 2.0 * (ex * lx + ey * ly);
            C = (lx * lx + ly * ly) - g[0] * g[0];
            break;
        case TRC_GM_CONE_INF: case TRC_GM_CONE_FINITE: case TRC_GM_FRUSTUM: case TRC_GM_FRUSTUM_RECTCUT: {
            double c = g[0], a = g[1];                            // cone.py:68-70
            A = ex * ex + ey * ey - (c * ez) * (c * ez);
            B = 2.0 * (lx * ex + ly * ey - c * c * (lz - a) * ez);
            C = lx * lx + ly * ly - (c * (lz - a)) * (c * (lz - a));
            break;
        }
        case TRC_GM_QUADRATIC: case TRC_GM_QUADRATIC_RECT:        // quadratic_surface.py:57-59
            A = g[0] * ex * ex + g[1] * ey * ey + g[2] * ex * ey;
            B = 2.0 * g[0] * ex * lx + 2.0 * g[1] * ey * ly + g[2] * (lx * ey + ly * ex) + g[3] * ex +
                g[4] * ey - ez;
            C = g[0] * lx * lx + g[1] * ly * ly + g[2] * lx * ly + g[3] * lx + g[4] * ly + g[5] - lz;
            break;
        case TRC_GM_ELLIPSOID: case TRC_GM_ELLIPSOID_CUT:         // ellipsoid.py:31-33
            A = g[0] * ex * ex + g[1] * ey * ey + g[2] * ez * ez;
            B = 2.0 * g[0] * ex * lx + 2.0 * g[1] * ey * ly + 2.0 * g[2] * ez * lz;
            C = g[0] * lx * lx + g[1] * ly * ly + g[2] * lz * lz - 1.0;
            break;
        default:
            return TRC_INF;
        }
    }
    double delta = B * B - 4.0 * A * C;
    if (!(delta >= 1e-6)) return TRC_INF;                         // quadric.py:57-58
    double t0, t1;
    if (A == 0.0) {                                               // :77-82
        if (B == 0.0) return TRC_INF;
        t0 = t1 = -C / B;
    } else if (B == 0.0) {                                        // :84-85
        t1 = sqrt(-C / A);
        t0 = -t1;
    } else {                                                      // :87-89
        double q = -0.5 * (B + (B > 0.0 ? 1.0 : -1.0) * sqrt(delta));
        t0 = q / A;
        t1 = C / q;
    }
    double eps;
    int mode = trc_quadric_select_mode(kind, &eps);
    bool b0 = t0 >= 1e-6, b1 = t1 >= 1e-6;                         // base rule
    int base_sel = (b0 && b1) ? 1 : (b0 ? 0 : (b1 ? 1 : -1));
    int sel;
    if (mode == 0) {
        sel = base_sel;
    } else {
        bool in0 = t0 > eps, in1 = t1 > eps;
        if (in0) {
            double lx, ly, lz;
            trc_to_local_point(rec, vx + dx * t0, vy + dy * t0, vz + dz * t0, &lx, &ly, &lz);
            in0 = trc_quadric_aperture(kind, rec, g, lx, ly, lz);
        }
        if (in1) {
            double lx, ly, lz;
            trc_to_local_point(rec, vx + dx * t1, vy + dy * t1, vz + dz * t1, &lx, &ly, &lz);
            in1 = trc_quadric_aperture(kind, rec, g, lx, ly, lz);
        }
        if (mode == 1) {
            sel = (in0 && in1) ? 1 : (in0 ? 0 : (in1 ? 1 : -1));
        } else {
            if (!(in0 || in1)) sel = -1;
            else if (in0 != in1) sel = in0 ? 0 : 1;
            else sel = base_sel;
        }
    }
    if (sel < 0) return TRC_INF;
    return sel ? t1 : t0;
}

TRC_HD bool trc_gm_is_flat(int kind) { return kind <= TRC_GM_TRIANGLE || kind == TRC_GM_POLYGON; }

// GeometryManager.find_intersections for one ray: parametric distance, +inf = miss
TRC_HD double trc_intersect(const double *rec, const double *extra, double vx, double vy, double vz,
                            double dx, double dy, double dz) {
    int kind = trc_rec_gm_kind(rec);
    if (trc_gm_is_flat(kind)) return trc_intersect_flat(kind, rec, extra, vx, vy, vz, dx, dy, dz);
    return trc_intersect_quadric(kind, rec, vx, vy, vz, dx, dy, dz);
}

// GeometryManager.get_normals for one hit: unit normal opposing the incident direction.
// flat_surface.py:84-91; paraboloid.py:57-67; sphere_surface.py:44-49; cylinder.py:26-30;
// cone.py:42-54; quadratic_surface.py:32-40; ellipsoid.py:49-57.
TRC_HD void trc_normal(const double *rec, double hx, double hy, double hz, double dx, double dy, double dz,
                       double *nx, double *ny, double *nz) {
    int kind = trc_rec_gm_kind(rec);
    const double *g = rec + TRC_REC_HDR;
    if (trc_gm_is_flat(kind)) {
        double ux = rec[2], uy = rec[5], uz = rec[8];
        double dt = dx * ux + dy * uy + dz * uz;
        if (dt > 0.0) { ux = -ux; uy = -uy; uz = -uz; }          // backside, flat_surface.py:57-60
        *nx = ux; *ny = uy; *nz = uz;
        return;
    }
    if (kind == TRC_GM_SPHERE || kind == TRC_GM_HEMISPHERE || kind == TRC_GM_SPHERE_RECT || kind == TRC_GM_SPHERE_CUT) {
        double ux = hx - rec[9], uy = hy - rec[10], uz = hz - rec[11];
        double sides = (-ux) * dx + (-uy) * dy + (-uz) * dz;
        if (sides < 0.0) { ux = -ux; uy = -uy; uz = -uz; }
        double inv = 1.0 / sqrt(ux * ux + uy * uy + uz * uz);
        *nx = ux * inv; *ny = uy * inv; *nz = uz * inv;
        return;
    }
    double lx, ly, lz, ex, ey, ez;
    trc_to_local_point(rec, hx, hy, hz, &lx, &ly, &lz);
    trc_to_local_dir(rec, dx, dy, dz, &ex, &ey, &ez);
    double ux, uy, uz;
    bool flip;
    switch (kind) {
    case TRC_GM_CYL_INF: case TRC_GM_CYL_FINITE: case TRC_GM_CYL_RECTCUT:
        ux = lx / g[0]; uy = ly / g[0]; uz = 0.0;
        flip = (ux * ex + uy * ey) > 0.0;
        break;
    case TRC_GM_CONE_INF: case TRC_GM_CONE_FINITE: case TRC_GM_FRUSTUM: case TRC_GM_FRUSTUM_RECTCUT: {
        ux = 2.0 * lx; uy = 2.0 * ly; uz = -2.0 * (lz - g[1]) * (g[0] * g[0]);
        double inv = 1.0 / sqrt(ux * ux + uy * uy + uz * uz);
        ux *= inv; uy *= inv; uz *= inv;
        flip = (ex * ux + ey * uy + ez * uz) > 1e-9;
        if (flip) { ux = -ux; uy = -uy; uz = -uz; }
        if (lz == g[1]) { ux = 0.0; uy = 0.0; uz = -1.0; }       // apex, cone.py:52-54
        trc_to_global_dir(rec, ux, uy, uz, nx, ny, nz);
        return;
    }
    default: {
        switch (kind) {
        case TRC_GM_PARAB_CYL: case TRC_GM_PARAB_TROUGH:
            ux = 2.0 * lx * g[0]; uy = 0.0; uz = -1.0; break;
        case TRC_GM_QUADRATIC: case TRC_GM_QUADRATIC_RECT:
            ux = 2.0 * lx * g[0] + g[2] * ly + g[3]; uy = 2.0 * ly * g[1] + g[2] * lx + g[4]; uz = -1.0; break;
        case TRC_GM_ELLIPSOID: case TRC_GM_ELLIPSOID_CUT:
            ux = 2.0 * lx * g[0]; uy = 2.0 * ly * g[1]; uz = 2.0 * lz * g[2]; break;
        default:  // paraboloid family
            ux = 2.0 * lx * g[0]; uy = 2.0 * ly * g[1]; uz = -1.0; break;
        }
        double inv = 1.0 / sqrt(ux * ux + uy * uy + uz * uz);
        ux *= inv; uy *= inv; uz *= inv;
        flip = (ex * ux + ey * uy + ez * uz) > 0.0;
        break;
    }
    }
    if (flip) { ux = -ux; uy = -uy; uz = -uz; }
    trc_to_global_dir(rec, ux, uy, uz, nx, ny, nz);
}

// ---------------------------------------------------------------------------------------------
// E2 -- nearest hit over all surfaces, brute force.  tracer_engine.py:45-63: surfaces in index
// order, t == 0 is not a hit, a later surface replaces only on strictly smaller t.
// ---------------------------------------------------------------------------------------------
TRC_HD void trc_nearest_brute(const double *recs, int stride, int n_surf, const double *extra, double vx,
                              double vy, double vz, double dx, double dy, double dz, double *t_best,
                              int *s_best) {
    double tb = TRC_INF;
    int sb = -1;
    for (int s = 0; s < n_surf; ++s) {
        double t = trc_intersect(recs + (size_t)s * stride, extra, vx, vy, vz, dx, dy, dz);
        if (t != 0.0 && t < tb) { tb = t; sb = s; }
    }
    *t_best = tb;
    *s_best = sb;
}

// ---------------------------------------------------------------------------------------------
// K2 -- Kd-tree traversal.  accel_tree.py:213-330.  The reference marks every leaf a ray crosses
// and lets intersect_ray pick the nearest hit among the marked surfaces; here the candidates of a
// leaf are tested as soon as the leaf is reached and the walk stops once the best hit is in front
// of everything still on the stack.  The surviving (t, surface) pair is the same: lowest surface
// index among the minimal t.
// ---------------------------------------------------------------------------------------------
struct trc_kd_view {
    const int32_t *node_a;   // per node: flag (2 low bits) | child or leaf_off << 2
    const int32_t *node_b;   // per node: leaf_cnt (leaf)
    const double *split;     // per node
    const int32_t *leaf_surfs;
    const int32_t *always;
    int32_t n_always;
    double bmin[3], bmax[3];
};

#ifndef TRC_KD_STACK
#define TRC_KD_STACK 32
#endif

template <class Stack>
TRC_HD void trc_nearest_kd(const trc_kd_view &kd, Stack &stk, const double *recs, int stride,
                           const double *extra, double vx, double vy, double vz, double dx, double dy,
                           double dz, double *t_best, int *s_best) {
    double tb = TRC_INF;
    int sb = -1;
    // surfaces of objects without boundaries are always candidates (accel_tree.py:59-73, :235)
    for (int k = 0; k < kd.n_always; ++k) {
        int s = kd.always[k];
        double t = trc_intersect(recs + (size_t)s * stride, extra, vx, vy, vz, dx, dy, dz);
        if (t != 0.0 && (t < tb || (t == tb && s < sb))) { tb = t; sb = s; }
    }
    // root slab test, accel_tree.py:314-330
    const double v[3] = {vx, vy, vz}, d[3] = {dx, dy, dz};
    double inv[3];
    double tmin = 0.0, tmax = TRC_INF;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        inv[i] = 1.0 / d[i];
        double lo = ((d[i] < 0.0 ? kd.bmax[i] : kd.bmin[i]) - v[i]) * inv[i];
        double hi = ((d[i] < 0.0 ? kd.bmin[i] : kd.bmax[i]) - v[i]) * inv[i];
        if (lo > hi) { double tmp = lo; lo = hi; hi = tmp; }
        tmin = fmax(tmin, lo);   // NaN (0*inf) is ignored: conservative
        tmax = fmin(tmax, hi);
    }
    if (tmax > 0.0 && !(tmin > tmax)) {
        int node = 0;
        int sp = 0;
        for (;;) {
            int a = kd.node_a[node];
            int flag = a & 3;
            if (flag != 3) {
                double split = kd.split[node];
                double pv = (flag == 0) ? v[0] : (flag == 1 ? v[1] : v[2]);
                double dv = (flag == 0) ? d[0] : (flag == 1 ? d[1] : d[2]);
                double iv = (flag == 0) ? inv[0] : (flag == 1 ? inv[1] : inv[2]);
                double tp = (split - pv) * iv;                      // accel_tree.py:255
                int c1 = a >> 2, c2 = c1 + 1;
                bool below = (pv < split) || (pv == split && dv <= 0.0);   // :259
                if (!below) { int tmp = c1; c1 = c2; c2 = tmp; }
                if (tp > tmax || tp <= 0.0) node = c1;              // :264
                else if (tp < tmin) node = c2;                      // :266
                else {                                               // :268-274
                    if (sp < TRC_KD_STACK) { stk.push(sp, c2, tmax); ++sp; }
                    node = c1;
                    tmax = tp;
                }
            } else {
                int off = a >> 2, cnt = kd.node_b[node];
                for (int k = 0; k < cnt; ++k) {
                    int s = kd.leaf_surfs[off + k];
                    double t = trc_intersect(recs + (size_t)s * stride, extra, vx, vy, vz, dx, dy, dz);
                    if (t != 0.0 && (t < tb || (t == tb && s < sb))) { tb = t; sb = s; }
                }
                if (sp == 0) break;
                --sp;
                // the far child was pushed with t_min = t_plane, which is exactly where the near
                // subtree just ended (the last leaf of a subtree inherits its interval's end)
                tmin = tmax;
                stk.pop(sp, &node, &tmax);
                if (tb < tmin) break;   // everything left starts behind the best hit
            }
        }
    }
    *t_best = tb;
    *s_best = sb;
}

// ---------------------------------------------------------------------------------------------
// K2 (fast form) -- single-precision CONSERVATIVE candidate search + exact float64 tests.
//
// Only the *choice of candidates* is done in float32; every candidate that survives is tested with the
// exact float64 trc_intersect above, and the winner is the lowest surface index among the minimal t, so
// the result equals brute force over all surfaces (= the reference) as long as no true hit is ever
// discarded.  That is guaranteed by construction:
//   * every bounded surface carries an axis-aligned box derived from its own geometry (not from the
//     user's BoundaryBox), inflated by `delta` >> float32 rounding of any coordinate in the scene;
//     unbounded surfaces (infinite plane, paraboloid, cylinder, cone ...) are always tested exactly;
//   * the ray is first advanced (in float64) to where it enters the scene box, so float32 coordinates are
//     bounded by the scene size, and all coordinates are taken relative to the scene centre;
//   * Kd split planes are treated as slabs of half-width delta: a child is skipped only if the ray's
//     interval lies entirely beyond the slab; when the origin is within delta of a plane both children are
//     visited; the walk stops early only when the best exact hit is nearer than the entry of everything
//     left on the stack by more than the margin.
// ---------------------------------------------------------------------------------------------
struct trc_accel_view {
    const float *sbox;         // n_surf * 6: lo xyz, hi xyz relative to cen, inflated; unbounded: -inf / +inf
    const uint32_t *nodes;     // 2 words per node: interior {float split (rel), child<<2|axis}; leaf {leaf_off, cnt<<2|3}
    const uint16_t *leaf_surfs;
    const int32_t *always;     // Kd always_relevant surfaces
    const int32_t *unbounded;  // surfaces without a box (tested exactly for every ray)
    int32_t n_always, n_unbounded, n_surf, has_kd;
    float root[6];             // Kd root box, relative, inflated
    float delta;
    double cen[3];
    double slo[3], shi[3];     // scene box (absolute, inflated): union of all finite surface boxes
};

struct trc_ray32 {
    float ox, oy, oz, dx, dy, dz, ix, iy, iz;
};

// does the ray (t' >= 0) cross the inflated box?  NaNs (0 * inf) are ignored by fminf/fmaxf: conservative
TRC_HD bool trc_box_hit32(const float *b, const trc_ray32 &r) {
    float ax = (b[0] - r.ox) * r.ix, bx = (b[3] - r.ox) * r.ix;
    float ay = (b[1] - r.oy) * r.iy, by = (b[4] - r.oy) * r.iy;
    float az = (b[2] - r.oz) * r.iz, bz = (b[5] - r.oz) * r.iz;
    float lo = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
    float hi = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    // parametric rounding: 1e-4 relative on the far end is far above float32 error and far below delta's effect
    return hi * 1.0001f + 1e-6f >= lo;
}

// The same question for the surface's own (oriented) box, a record of TRC_OBB_STRIDE floats made by trc_bounds.h:
// [A0 A1 A2 c0 | A3 A4 A5 c1 | A6 A7 A8 c2 | lo0 lo1 lo2 hi0 | hi1 hi2 - -], A = R^T, c = frame origin relative to the
// scene centre, lo / hi = the surface's box in its own frame inflated by delta.  (ox, oy, oz) is relative to the scene
// centre like trc_ray32's origin.  The ray is taken into the frame in float32: a point of the ray at distance t moves by
// ~1e-7 (|o - c| + t), which delta (>= 2.5e-5 of the scene extent) covers with two orders of magnitude to spare as long as
// the origin is no farther from the scene than a few times its size -- the callers advance it to the scene first.
TRC_HD bool trc_obb_hit32(const float *B, float ox, float oy, float oz, float dx, float dy, float dz) {
    const float rx = ox - B[3], ry = oy - B[7], rz = oz - B[11];
    const float lx = B[0] * rx + B[1] * ry + B[2] * rz, ly = B[4] * rx + B[5] * ry + B[6] * rz, lz = B[8] * rx + B[9] * ry + B[10] * rz;
    const float ex = B[0] * dx + B[1] * dy + B[2] * dz, ey = B[4] * dx + B[5] * dy + B[6] * dz, ez = B[8] * dx + B[9] * dy + B[10] * dz;
#if defined(__HIP_DEVICE_COMPILE__)
    const float ix = __builtin_amdgcn_rcpf(ex), iy = __builtin_amdgcn_rcpf(ey), iz = __builtin_amdgcn_rcpf(ez);
#else
    const float ix = 1.0f / ex, iy = 1.0f / ey, iz = 1.0f / ez;
#endif
    const float ax = (B[12] - lx) * ix, bx = (B[15] - lx) * ix;
    const float ay = (B[13] - ly) * iy, by = (B[16] - ly) * iy;
    const float az = (B[14] - lz) * iz, bz = (B[17] - lz) * iz;
    const float lo = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
    const float hi = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    return hi * 1.0001f + 1e-6f >= lo;       // NaNs (0 * inf) are ignored by fminf/fmaxf: conservative, as in trc_box_hit32
}

// The same question for a triangular face (triangular_face.py:59-72) from its corner and two edges in single precision
// (relative to the scene centre like the ray's origin): can the exact test succeed?  With T = o - v0, the scaled barycentric
// coordinates are u = T . (d x e2), v = d . (T x e1) over det = e1 . (d x e2); the point is inside when u, v >= 0 and u + v <= det
// (signs for det > 0).  A displacement of the ray by delta -- which covers single-precision rounding of the origin, the direction
// and the products with two orders of magnitude to spare, as for the boxes above -- changes u and v by at most delta * max|edge|:
// the test allows E = 1.5 delta max|edge| on every comparison, so that a ray is rejected only when the exact test rejects it too.
// A ray within 1e-5 of the face's plane is passed on without a decision (the sign of det is not reliable there; the exact test
// refuses |d . n| <= 1e-7 itself).  t * det = e2 . (T x e1) sorts out the faces behind the origin.
// ent: [index | v0 (3) | e1 (3) | E | e2 (3) | max|edge|^2]
TRC_HD bool trc_tri_hit32(float v0x, float v0y, float v0z, float e1x, float e1y, float e1z, float E, float e2x, float e2y, float e2z,
                          float emax2, float delta, const trc_ray32 &r) {
    const float tx = r.ox - v0x, ty = r.oy - v0y, tz = r.oz - v0z;
    const float px = fmaf(r.dy, e2z, -(r.dz * e2y)), py = fmaf(r.dz, e2x, -(r.dx * e2z)), pz = fmaf(r.dx, e2y, -(r.dy * e2x));
    const float det = fmaf(e1x, px, fmaf(e1y, py, e1z * pz));
    const float qx = fmaf(ty, e1z, -(tz * e1y)), qy = fmaf(tz, e1x, -(tx * e1z)), qz = fmaf(tx, e1y, -(ty * e1x));
    float u = fmaf(tx, px, fmaf(ty, py, tz * pz));
    float v = fmaf(r.dx, qx, fmaf(r.dy, qy, r.dz * qz));
    float tn = fmaf(e2x, qx, fmaf(e2y, qy, e2z * qz));
    if (det < 0.0f) { u = -u; v = -v; tn = -tn; }
    const float ad = fabsf(det);
    if (!(ad > 1e-5f * emax2)) return true;
    return u >= -E && v >= -E && u + v <= ad + 2.0f * E && tn >= -delta * (ad + emax2);
}

// One interior-node step of the conservative single-precision walk (used by trc_nearest_accel32 below and by the
// wave-cooperative kernel).  w0/w1: the packed node.  Returns the node to continue with; when *push is set the
// caller must push (*push_na = other child << 2 | axis code, *push_t = interval end of that child) and the
// interval of the continued child has been shortened.  Axis code 3 on the stack means "both children with the
// full interval" (origin within delta of the plane).
TRC_HD uint32_t trc_kd32_step(uint32_t w0, uint32_t w1, const trc_ray32 &r, float delta, float tmin, float *tmax,
                              bool *push, uint32_t *push_na, float *push_t) {
    uint32_t axis = w1 & 3u;
    float split = __builtin_bit_cast(float, w0);
    float o = axis == 0 ? r.ox : (axis == 1 ? r.oy : r.oz);
    float iv = axis == 0 ? r.ix : (axis == 1 ? r.iy : r.iz);
    float diff = split - o;
    uint32_t left = w1 >> 2, right = left + 1;
    *push = false;
    if (fabsf(diff) <= delta) {
        *push = true; *push_na = (right << 2) | 3u; *push_t = *tmax;
        return left;
    }
    uint32_t nearc = diff > 0.0f ? left : right, farc = diff > 0.0f ? right : left;
    float tp = diff * iv;
    float dt = delta * fabsf(iv);
    if (!(tp - dt <= *tmax) || tp + dt < 0.0f) return nearc;      // slab beyond the interval / behind the origin
    if (tp + dt < tmin) return farc;                              // interval starts after the slab
    *push = true; *push_na = (farc << 2) | axis; *push_t = *tmax;  // far child keeps the interval end
    *tmax = fminf(*tmax, tp + dt);
    return nearc;
}

// interval start of a popped child: the near subtree ended at (plane + dt), the far one starts at (plane - dt)
TRC_HD float trc_kd32_pop_tmin(uint32_t axis_code, const trc_ray32 &r, float delta, float tmax_now) {
    if (axis_code == 3u) return 0.0f;
    float iv = axis_code == 0 ? r.ix : (axis_code == 1 ? r.iy : r.iz);
    return fmaxf(0.0f, tmax_now - 2.0f * delta * fabsf(iv) - 1e-5f * fabsf(tmax_now));
}

// root interval of the walk
TRC_HD bool trc_kd32_root(const float *root, const trc_ray32 &r, float *tmin, float *tmax) {
    float ax = (root[0] - r.ox) * r.ix, bx = (root[3] - r.ox) * r.ix;
    float ay = (root[1] - r.oy) * r.iy, by = (root[4] - r.oy) * r.iy;
    float az = (root[2] - r.oz) * r.iz, bz = (root[5] - r.oz) * r.iz;
    *tmin = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
    *tmax = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz)) * 1.0001f + 1e-6f;
    return *tmax >= *tmin;
}

// ---------------------------------------------------------------------------------------------
// Uniform grid over the scene box (built by trc_bounds.h from the same inflated surface boxes): the streaming engine's
// candidate search.  A ray visits the cells of a 3-D DDA in single precision; every surface whose box, inflated by a
// further 2*delta, overlaps a cell is listed there, so that the float32 rounding of the walk (~1e-5 of the scene
// extent, 400 times below delta) cannot lose a surface the exact ray touches.  Cell planes are recomputed from the
// integer cell index at every step (no accumulated error).
// ---------------------------------------------------------------------------------------------
// OT / LT: the integer types of the offsets and of the lists -- 16 bits for a grid that lives in LDS (at most 8192 cells and
// 65535 entries), 32 bits for one in global memory (meshes of 1e5 faces and more)
template <class OT, class LT>
struct trc_grid_view_t {
    const OT *off;             // cells + 1 offsets into list
    const LT *list;            // surface indices
    int32_t nx, ny, nz;
    float lox, loy, loz;       // grid origin relative to cen
    float csx, csy, csz;       // cell size
    float ivx, ivy, ivz;       // 1 / cell size
};
typedef trc_grid_view_t<uint16_t, uint16_t> trc_grid_view;
typedef trc_grid_view_t<uint32_t, uint32_t> trc_grid_view32;

struct trc_dda {
    int32_t cx, cy, cz;
    float tnx, tny, tnz;       // parameter at which the ray leaves the current cell along each axis
};

TRC_HD float trc_dda_plane_t(float lo, float cs, int32_t c, float o, float iv) {
    // leaving plane of cell c along one axis: upper face when the ray goes up (iv >= 0), lower face otherwise
    if (!(fabsf(iv) < 3.0e38f)) return TRC_INF32;      // parallel to the axis: never leaves
    float plane = lo + (float)(c + (iv >= 0.0f ? 1 : 0)) * cs;
    return (plane - o) * iv;
}

template <class GV>
TRC_HD void trc_dda_start(const GV &G, const trc_ray32 &r, float tmin, trc_dda *s) {
    float px = r.ox + tmin * (1.0f / r.ix), py = r.oy + tmin * (1.0f / r.iy), pz = r.oz + tmin * (1.0f / r.iz);
    int32_t cx = (int32_t)floorf((px - G.lox) * G.ivx), cy = (int32_t)floorf((py - G.loy) * G.ivy),
            cz = (int32_t)floorf((pz - G.loz) * G.ivz);
    s->cx = cx < 0 ? 0 : (cx >= G.nx ? G.nx - 1 : cx);
    s->cy = cy < 0 ? 0 : (cy >= G.ny ? G.ny - 1 : cy);
    s->cz = cz < 0 ? 0 : (cz >= G.nz ? G.nz - 1 : cz);
    s->tnx = trc_dda_plane_t(G.lox, G.csx, s->cx, r.ox, r.ix);
    s->tny = trc_dda_plane_t(G.loy, G.csy, s->cy, r.oy, r.iy);
    s->tnz = trc_dda_plane_t(G.loz, G.csz, s->cz, r.oz, r.iz);
}

template <class GV>
TRC_HD int32_t trc_dda_cell(const GV &G, const trc_dda &s) { return (s.cz * G.ny + s.cy) * G.nx + s.cx; }

// moves to the next cell; false when the ray has left the grid or passed tmax
template <class GV>
TRC_HD bool trc_dda_next(const GV &G, const trc_ray32 &r, float tmax, trc_dda *s) {
    if (s->tnx <= s->tny && s->tnx <= s->tnz) {
        if (!(s->tnx <= tmax)) return false;
        s->cx += (r.ix >= 0.0f) ? 1 : -1;
        if ((uint32_t)s->cx >= (uint32_t)G.nx) return false;
        s->tnx = trc_dda_plane_t(G.lox, G.csx, s->cx, r.ox, r.ix);
    } else if (s->tny <= s->tnz) {
        if (!(s->tny <= tmax)) return false;
        s->cy += (r.iy >= 0.0f) ? 1 : -1;
        if ((uint32_t)s->cy >= (uint32_t)G.ny) return false;
        s->tny = trc_dda_plane_t(G.loy, G.csy, s->cy, r.oy, r.iy);
    } else {
        if (!(s->tnz <= tmax)) return false;
        s->cz += (r.iz >= 0.0f) ? 1 : -1;
        if ((uint32_t)s->cz >= (uint32_t)G.nz) return false;
        s->tnz = trc_dda_plane_t(G.loz, G.csz, s->cz, r.oz, r.iz);
    }
    return true;
}

// float64 entry into the scene box and the relative single-precision ray from there
TRC_HD bool trc_ray32_prepare(const double *slo, const double *shi, const double *cen, double vx, double vy, double vz,
                              double dx, double dy, double dz, trc_ray32 *r, double *t_entry) {
    double t0 = 0.0, t1 = TRC_INF;
    const double v[3] = {vx, vy, vz}, d[3] = {dx, dy, dz};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#if defined(__HIP_DEVICE_COMPILE__)
        // the entry distance only places the single-precision origin inside the (inflated) box: a float reciprocal and
        // one Newton step (4e-15 relative) instead of the ~28 instructions of an IEEE double division; 1/0 and 1/tiny
        // stay infinite with their sign
        float fi = __builtin_amdgcn_rcpf((float)d[i]);
        double inv = (double)fi;
        if (fabsf(fi) < 1e30f) inv = inv * (2.0 - d[i] * inv);
#else
        double inv = 1.0 / d[i];
#endif
        double a = (slo[i] - v[i]) * inv, b = (shi[i] - v[i]) * inv;
        t0 = fmax(t0, fmin(a, b));
        t1 = fmin(t1, fmax(a, b));
    }
    r->ox = (float)(vx + t0 * dx - cen[0]);
    r->oy = (float)(vy + t0 * dy - cen[1]);
    r->oz = (float)(vz + t0 * dz - cen[2]);
    r->dx = (float)dx; r->dy = (float)dy; r->dz = (float)dz;
#if defined(__HIP_DEVICE_COMPILE__)
    r->ix = __builtin_amdgcn_rcpf(r->dx); r->iy = __builtin_amdgcn_rcpf(r->dy); r->iz = __builtin_amdgcn_rcpf(r->dz);   // 1 ulp
#else
    r->ix = 1.0f / r->dx; r->iy = 1.0f / r->dy; r->iz = 1.0f / r->dz;
#endif
    *t_entry = t0;
    return t1 >= t0;
}

#define TRC_TEST_EXACT(S)                                                                                   \
    do {                                                                                                    \
        int _s = (S);                                                                                       \
        double _t = trc_intersect(recs + (size_t)_s * stride, extra, vx, vy, vz, dx, dy, dz);               \
        if (_t != 0.0 && (_t < tb || (_t == tb && _s < sb))) { tb = _t; sb = _s; }                          \
    } while (0)

// Stack: push(sp, node_and_axis, tmax) / pop(sp, &node_and_axis, &tmax)
template <class Stack>
TRC_HD void trc_nearest_accel32(const trc_accel_view &A, Stack &stk, const double *recs, int stride,
                                const double *extra, double vx, double vy, double vz, double dx, double dy,
                                double dz, bool use_kd, double *t_best, int *s_best) {
    double tb = TRC_INF;
    int sb = -1;
    for (int k = 0; k < A.n_unbounded; ++k) TRC_TEST_EXACT(A.unbounded[k]);
    trc_ray32 r;
    double t0;
    if (trc_ray32_prepare(A.slo, A.shi, A.cen, vx, vy, vz, dx, dy, dz, &r, &t0)) {
        if (!use_kd) {
            for (int s = 0; s < A.n_surf; ++s) {
                const float *b = A.sbox + 6 * (size_t)s;
                if (b[3] == TRC_INF && b[0] == -TRC_INF) continue;     // unbounded: already tested
                if (trc_box_hit32(b, r)) TRC_TEST_EXACT(s);
            }
        } else {
            for (int k = 0; k < A.n_always; ++k) {
                int s = A.always[k];
                const float *b = A.sbox + 6 * (size_t)s;
                if (b[3] == TRC_INF && b[0] == -TRC_INF) continue;
                if (trc_box_hit32(b, r)) TRC_TEST_EXACT(s);
            }
            float tmin, tmax;
            if (trc_kd32_root(A.root, r, &tmin, &tmax)) {
                uint32_t node = 0;
                int sp = 0;
                for (;;) {
                    uint32_t w0 = A.nodes[2 * node], w1 = A.nodes[2 * node + 1];
                    if ((w1 & 3u) != 3u) {
                        bool push;
                        uint32_t na;
                        float pt;
                        node = trc_kd32_step(w0, w1, r, A.delta, tmin, &tmax, &push, &na, &pt);
                        if (push) { stk.push(sp, na, pt); ++sp; }
                    } else {
                        uint32_t off = w0, cnt = w1 >> 2;
                        for (uint32_t k = 0; k < cnt; ++k) {
                            int s = A.leaf_surfs[off + k];
                            if (trc_box_hit32(A.sbox + 6 * (size_t)s, r)) TRC_TEST_EXACT(s);
                        }
                        // Next pending child.  One whose interval starts behind the best hit is passed over, not taken as the
                        // end of the walk: below it on the stack there may be the other child of a node whose plane lies within
                        // delta of the origin (code 3), and that one starts at the origin.  Passing over keeps the chain of
                        // interval ends that the next pop starts from.
                        bool pending = false;
                        while (sp > 0) {
                            --sp;
                            uint32_t na;
                            float tmax_far;
                            stk.pop(sp, &na, &tmax_far);
                            node = na >> 2;
                            tmin = trc_kd32_pop_tmin(na & 3u, r, A.delta, tmax);
                            tmax = tmax_far;
                            if (sb >= 0) {
                                float tbr = (float)(tb - t0);
                                if (tbr < tmin - (1e-3f + 1e-5f * fabsf(tbr))) continue;
                            }
                            pending = true;
                            break;
                        }
                        if (!pending) break;
                    }
                }
            }
        }
    }
    *t_best = tb;
    *s_best = sb;
}

// ---------------------------------------------------------------------------------------------
// O1..O6 -- optics.  One interaction of one ray; at most two outgoing rays.
// ---------------------------------------------------------------------------------------------
struct trc_ray_out {
    double dx, dy, dz;
    double e;
    double ref;
    int blk;   // 0: reflected block, 1: refracted block (ordering inside a surface's output bundle); scattering optics: 0 scattered,
               // 1 reflected, 2 refracted (optics_callables.py:1136-1168: "stacking together the scattered, reflected and refracted rays")
    double back;   // 0: the ray leaves from the hit point.  > 0: a volume event on the way -- the ray never reached the surface, it
                   // leaves from the point `back` before the hit along its old direction, and the surface records nothing
    double shift;  // the ray leaves from the hit point moved by `shift` along the oriented normal (PeriodicBoundary, :717); 0: from the hit
    double sf;     // factor on the spectrum a polychromatic ray carries (`outg._spectra *= ...` of the classes that have the line);
                   // read by the ordered engine and the per-surface protocol only
};

TRC_HD void trc_reflect(double dx, double dy, double dz, double nx, double ny, double nz, double *ox,
                        double *oy, double *oz) {
    double dn = dx * nx + dy * ny + dz * nz;                        // optics.py:156-157
    *ox = dx - 2.0 * (dn * nx);
    *oy = dy - 2.0 * (dn * ny);
    *oz = dz - 2.0 * (dn * nz);
}

// N.round(x, 14), spatial_geometry.py:18 (x * 1e-14 for x / 1e14: one unit of the last place apart at most, a float64 division less)
TRC_HD double trc_round14(double x) { return rint(x * 1e14) * 1e-14; }

// minimal rotation taking z to n, applied to e (ray_trace_utils/vector_manipulations.py:56-90,
// spatial_geometry.py:8-22).  Used for the mirror slope error.
TRC_HD void trc_rotate_z_to_normal(double ex, double ey, double ez, double nx, double ny, double nz,
                                   double *ox, double *oy, double *oz) {
    // angle = arccos(n_z) (:84); only its sine and cosine are used (:18): cos = n_z, sin = sqrt(1 - n_z^2) >= 0 -- an arc cosine and
    // a sine / cosine pair less per call, the same numbers to a unit of the last place before they are rounded to 14 decimals
    if (nz == 1.0) { *ox = ex; *oy = ey; *oz = ez; return; }          // (arccos gives 0 for 1 and nothing else)
    // axis = unit(z x n); undefined -> x axis
    double kx = -ny, ky = nx, kz = 0.0;
    double kn = sqrt(kx * kx + ky * ky);
    kx /= kn; ky /= kn;
    if (kx != kx) { kx = 1.0; ky = 0.0; kz = 0.0; }
    double s = sqrt((1.0 - nz) * (1.0 + nz)), c = nz;                // (nan for |n_z| > 1, as arccos)
    s = trc_round14(s); c = trc_round14(c);
    double vv = 1.0 - c;
    // M = outer(k,k)*v + I*c + [k]x*s
    double m00 = kx * kx * vv + c, m01 = kx * ky * vv - kz * s, m02 = kx * kz * vv + ky * s;
    double m10 = ky * kx * vv + kz * s, m11 = ky * ky * vv + c, m12 = ky * kz * vv - kx * s;
    double m20 = kz * kx * vv - ky * s, m21 = kz * ky * vv + kx * s, m22 = kz * kz * vv + c;
    *ox = m00 * ex + m01 * ey + m02 * ez;
    *oy = m10 * ex + m11 * ey + m12 * ez;
    *oz = m20 * ex + m21 * ey + m22 * ez;
}

// frame whose columns are (perp, n x perp, n), applied to a (spatial_geometry.py:41-48)
TRC_HD void trc_rotation_to_z_apply(double nx, double ny, double nz, double ax, double ay, double az,
                                    double *ox, double *oy, double *oz) {
    double px = ny, py = -nx, pz = 0.0;
    if (px == 0.0 && py == 0.0) { px = 1.0; py = 0.0; }
    double pn = sqrt(px * px + py * py + pz * pz);
    px /= pn; py /= pn;
    double cx = ny * pz - nz * py, cy = nz * px - nx * pz, cz = nx * py - ny * px;
    *ox = px * ax + cx * ay + nx * az;
    *oy = py * ax + cy * ay + ny * az;
    *oz = pz * ax + cz * ay + nz * az;
}

// cone-limited cosine-weighted direction about +z (sources.py:91-98)
TRC_HD void trc_pillbox_dir(double xi1, double xi2, double ang_range, double *ax, double *ay, double *az) {
    if (ang_range == 0.0) { *ax = 0.0; *ay = 0.0; *az = 1.0; return; }
    double s = sin(ang_range) * sqrt(xi2);
    double s1, c1;
    trc_sincos(xi1, &s1, &c1);
    *ax = c1 * s;
    *ay = s1 * s;
    *az = sqrt(1.0 - s * s);
}
// the same with the azimuth given as the uniform it is drawn from (azimuth = 2 pi u): sincospi instead of reducing 2 pi u, and
// no sine of the cone angle for the half space (sin(pi / 2) is 1 in float64 too)
TRC_HD void trc_pillbox_dir_u(double u, double xi2, double ang_range, double *ax, double *ay, double *az) {
    if (ang_range == 0.0) { *ax = 0.0; *ay = 0.0; *az = 1.0; return; }
    double s = (ang_range == 1.57079632679489661923 ? 1.0 : sin(ang_range)) * sqrt(xi2);
    double s1, c1;
    trc_sincos_2pi(u, &s1, &c1);
    *ax = c1 * s;
    *ay = s1 * s;
    *az = sqrt(1.0 - s * s);
}

// slope-error normal in the frame of the ideal normal (optics_callables.py:234-251)
TRC_HD void trc_slope_error_local(double sigma, bool bi_var, double g0, double g1, double u2, double *ex,
                                  double *ey, double *ez) {
    if (bi_var) {
        double tx = trc_tan_small(sigma * g0), ty = trc_tan_small(sigma * g1);
        double z = sqrt(1.0 / (1.0 + tx * tx + ty * ty));
        *ex = tx * z; *ey = ty * z; *ez = z;
    } else {
        double th = sigma * g0;
        double st, ct, sp, cp;
        trc_sincos_mostly_small(th, &st, &ct);
        trc_sincos_2pi(u2, &sp, &cp);
        *ez = ct;
        *ex = st * cp;
        *ey = st * sp;
    }
}

// unpolarised Fresnel reflectance, optics.py:28-38 (formula kept as written, through arccos/sin)
TRC_HD double trc_fresnel(double cos_abs, double n1, double n2) {
    // theta = arccos(|cos|) (:28) enters through its sine and cosine only: cos = |cos|, sin = sqrt(1 - cos^2)
    double foo = cos_abs, sth = sqrt((1.0 - cos_abs) * (1.0 + cos_abs));
    double sn = n1 / n2 * sth;
    double bar = sqrt(1.0 - sn * sn);
    double rs = (n1 * foo - n2 * bar) / (n1 * foo + n2 * bar);
    double rp = (n1 * bar - n2 * foo) / (n1 * bar + n2 * foo);
    return (rs * rs + rp * rp) / 2.0;
}

// N.interp: piecewise linear with clamped ends; table = n x then n y
TRC_HD double trc_interp_xy(const double *xs, const double *ys, int n, double x);
TRC_HD double trc_interp(const double *tab, int n, double x) { return trc_interp_xy(tab, tab + n, n, x); }
TRC_HD double trc_interp_xy(const double *xs, const double *ys, int n, double x) {
    if (!(x > xs[0])) return ys[0];
    if (x >= xs[n - 1]) return ys[n - 1];
    int lo = 0, hi = n - 1;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (xs[mid] <= x) lo = mid; else hi = mid;
    }
    double slope = (ys[lo + 1] - ys[lo]) / (xs[lo + 1] - xs[lo]);
    return slope * (x - xs[lo]) + ys[lo];
}

// RegularGridInterpolator (linear) on a (theta, lambda) grid, arguments clamped to the grid
// tab: n_theta, n_lambda, theta[n_theta], lambda[n_lambda], value[n_theta][n_lambda]
TRC_HD int trc_grid_cell(const double *xs, int n, double x, double *w) {
    if (!(x > xs[0])) { *w = 0.0; return 0; }
    if (x >= xs[n - 1]) { *w = 1.0; return n - 2; }
    int lo = 0, hi = n - 1;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (xs[mid] <= x) lo = mid; else hi = mid;
    }
    *w = (x - xs[lo]) / (xs[lo + 1] - xs[lo]);
    return lo;
}
TRC_HD double trc_interp2(const double *tab, double th, double lam) {
    int nt = (int)tab[0], nl = (int)tab[1];
    const double *ts = tab + 2, *ls = ts + nt, *v = ls + nl;
    double wt, wlam;
    int it = trc_grid_cell(ts, nt, th, &wt), il = trc_grid_cell(ls, nl, lam, &wlam);
    double v00 = v[it * nl + il], v01 = v[it * nl + il + 1], v10 = v[(it + 1) * nl + il], v11 = v[(it + 1) * nl + il + 1];
    return (1.0 - wt) * ((1.0 - wlam) * v00 + wlam * v01) + wt * ((1.0 - wlam) * v10 + wlam * v11);
}

// interface between a perfect dielectric and an absorbing medium, optics.py:63-81 (fresnel_to_attenuating):
// parallel / perpendicular reflectances and the refraction angle for an incidence angle th
TRC_HD void trc_fresnel_attenuating(double th, double n1, double n2, double k2, double *rp_out, double *rs_out,
                                    double *theta2) {
    double sth, cth;
    trc_sincos(th, &sth, &cth);
    double sn = n1 * sth;
    double b = n2 * n2 - k2 * k2 - sn * sn;
    double a = sqrt(b * b + 4.0 * (n2 * k2) * (n2 * k2));
    double p = sqrt(0.5 * (a + b)), q = sqrt(0.5 * (a - b));
    double c = n1 * cth;
    double rs = ((c - p) * (c - p) + q * q) / ((c + p) * (c + p) + q * q);
    double st = sn * tan(th);
    *rp_out = ((p - st) * (p - st) + q * q) / ((p + st) * (p + st) + q * q) * rs;
    *rs_out = rs;
    *theta2 = atan(sn / p);
}

// unpolarised mean used by FresnelConductorHomogenous (optics_callables.py:1523-1558)
TRC_HD double trc_fresnel_conductor(double cos_abs, double n1, double n2, double k2) {
    double rp, rs, t2;
    trc_fresnel_attenuating(acos(cos_abs), n1, n2, k2, &rp, &rs, &t2);
    return (rp + rs) / 2.0;
}

// optics(geometry, rays, selector) for one hit.
//   opt_kind / opt[8]: trc_surface_desc;  (ux,uy,uz): GeometryManager.up() = frame z axis;
//   (dx..), e, ref, wl: incident ray;  (nx..): oriented normal from trc_normal.
// Returns the number of outgoing rays (1 or 2) in out[].
// Incidence Angle Modifier of Martin and Ruiz (optics_callables.py:271-281): the reflected fraction (1 - absorptivity) is
// scaled by (1 - exp(-cos(theta)^c / a_r)) / (1 - exp(-1 / a_r)), theta the angle of incidence.  a_r == 0: no modifier.
TRC_HD double trc_iam(double a_r, double c, double dx, double dy, double dz, double nx, double ny, double nz) {
    if (a_r == 0.0) return 1.0;
    double dn = dx * nx + dy * ny + dz * nz;
    double wx = dn * nx, wy = dn * ny, wz = dn * nz;
    double cos_aoi = sqrt(wx * wx + wy * wy + wz * wz);
    return (1.0 - exp(-pow(cos_aoi, c) / a_r)) / (1.0 - exp(-1.0 / a_r));
}

// RefractiveHomogenous (optics_callables.py:1226-1296 on :836-858): the body of its case in trc_shade, shared with the
// scattering optics, whose rays that reach the surface are refracted like this
TRC_HD int trc_shade_refractive(const double *opt, double dx, double dy, double dz, double e, double ref, double path, double nx,
                                double ny, double nz, uint64_t seed, uint64_t rid, uint32_t event, trc_ray_out out[2]) {
    double na = opt[0], nb = opt[1];
    bool single = opt[2] != 0.0;
    double sigma = opt[3];
    double u0, u1, u2, u3;
    trc_uniform_pair(seed, rid, event, 0, &u0, &u1);
    trc_uniform_pair(seed, rid, event, 1, &u2, &u3);
    if (sigma >= 0.0) {                                         // normal perturbation :1227-1239
        double g0, g1;
        trc_normal_pair(u0, u1, &g0, &g1);
        double th = sigma * g0, phi = TRC_TWO_PI * u2;
        double st, ct, sp, cp;
        trc_sincos(th, &st, &ct);
        trc_sincos(phi, &sp, &cp);
        double ex = st * cp, ey = st * sp, ez = ct;
        double rx, ry, rz;
        trc_rotation_to_z_apply(nx, ny, nz, ex, ey, ez, &rx, &ry, &rz);
        nx = rx; ny = ry; nz = rz;
    }
    // attenuation in the medium the ray arrives through (RefractiveTransmissiveHomogenous :1326-1348 on Absorbant.attenuate
    // :874-889): coefficient opt[4] in the medium of index n1 = opt[0], opt[5] in the other, path scaled by opt[6]
    if (opt[7] != 0.0) e *= exp(-((ref == nb) ? opt[5] : opt[4]) * (path * opt[6]));
    double n1 = ref;
    double n2 = (n1 == na) ? nb : na;                           // :1217-1218
    double eta = n2 / n1;
    double cos1 = nx * dx + ny * dy + nz * dz;
    bool refracted = (cos1 * cos1) >= (1.0 - eta * eta);        // optics.py:180
    double R = 1.0;
    double tx = 0.0, ty = 0.0, tz = 0.0;
    if (refracted) {
        tx = (dx - cos1 * nx) / eta; ty = (dy - cos1 * ny) / eta; tz = (dz - cos1 * nz) / eta;   // :188
        double cos2 = sqrt(1.0 - 1.0 / (eta * eta) * (1.0 - cos1 * cos1));                        // :189
        double sg = (cos1 < 0.0) ? -1.0 : 1.0;
        tx += nx * cos2 * sg; ty += ny * cos2 * sg; tz += nz * cos2 * sg;                         // :190
        R = trc_fresnel(fabs(cos1), n1, n2);
    }
    if (single) {                                               // :1254-1280
        if (u3 <= R) {
            trc_reflect(dx, dy, dz, nx, ny, nz, &out[0].dx, &out[0].dy, &out[0].dz);
            out[0].e = e; out[0].ref = ref;
        } else {
            out[0].dx = tx; out[0].dy = ty; out[0].dz = tz;
            out[0].e = e; out[0].ref = n2; out[0].blk = 1;
        }
        return 1;
    }
    trc_reflect(dx, dy, dz, nx, ny, nz, &out[0].dx, &out[0].dy, &out[0].dz);     // :1284-1294
    out[0].e = e * R; out[0].ref = ref;
    if (!refracted) return 1;
    out[1].dx = tx; out[1].dy = ty; out[1].dz = tz;
    out[1].e = e * (1.0 - R); out[1].ref = n2;
    return 2;
}

// polar angle of a Henyey-Greenstein scattering event from its uniform (sampling.py:160-168; the closed form of the CDF)
TRC_HD double trc_hg_theta(double g, double Rv) {
    const double s = 2.0 * Rv - 1.0;
    if (g == 0.0) return acos(s);
    const double q = (1.0 - g * g) / (1.0 + g * s);
    double c = 1.0 / (2.0 * g) * (1.0 + g * g - q * q);
    c = c < -1.0 ? -1.0 : (c > 1.0 ? 1.0 : c);      // rounding at the ends of the range (the reference's arccos would give nan)
    return acos(c);
}

// KINDS: bit mask of the optics kinds the caller promises (the compiler drops the others); FULL = false further promises that no
// surface has an Incidence Angle Modifier and no Lambertian wall stands in an absorbing medium (a_r == 0, opt[2] == 0): the
// class-split shading kernels of the streaming engine are built from these.  The arithmetic of a kind is the same in every
// instance (a factor of exactly 1.0 is left out).
template <unsigned KINDS, bool FULL>
TRC_HD int trc_shade_k(int opt_kind, const double *opt, const double *extra, int extra_off, int extra_len,
                     double ux, double uy, double uz, double dx, double dy, double dz, double e, double ref,
                     double wl, double path, double nx, double ny, double nz, uint64_t seed, uint64_t rid,
                     uint32_t event, trc_ray_out out[2]) {
    // path: distance the ray travelled to this hit (Absorbant.attenuate, optics_callables.py:874-889: |hit - previous vertex|)
    if (opt_kind >= 0 && opt_kind < 32 && !((KINDS >> opt_kind) & 1u)) __builtin_unreachable();
    out[0].ref = ref;
    out[0].blk = 0;
    out[1].blk = 1;
    out[0].back = 0.0;
    out[1].back = 0.0;
    out[0].sf = 1.0;
    out[1].sf = 1.0;
    out[0].shift = 0.0;
    out[1].shift = 0.0;
    switch (opt_kind) {
    case TRC_OPT_PERIODIC_BOUNDARY:                                 // :703-723
        // out[0] is the ray that goes on (the one the fast engines follow); the ordered engine files the children by block, so
        // the stub (block 0, energy 0: culled) still stands before the moved ray (block 1) where the reference puts it
        out[0].dx = dx; out[0].dy = dy; out[0].dz = dz; out[0].e = e; out[0].blk = 1; out[0].shift = opt[0];
        out[1].dx = dx; out[1].dy = dy; out[1].dz = dz; out[1].e = 0.0; out[1].ref = ref; out[1].blk = 0; out[1].sf = 0.0;   // :710-713
        return 2;
    case TRC_OPT_TRANSPARENT:                                       // :106-113
        out[0].dx = dx; out[0].dy = dy; out[0].dz = dz; out[0].e = e;
        return 1;
    case TRC_OPT_REFLECTIVE:
    case TRC_OPT_ONE_SIDED_REFLECTIVE: {                            // :130-140, :201-212
        trc_reflect(dx, dy, dz, nx, ny, nz, &out[0].dx, &out[0].dy, &out[0].dz);
        double eo = e * (1.0 - opt[0]);
        if (FULL) eo *= trc_iam(opt[1], opt[2], dx, dy, dz, nx, ny, nz);       // Reflective_IAM :283-300
        if (opt_kind == TRC_OPT_ONE_SIDED_REFLECTIVE && (dx * ux + dy * uy + dz * uz) > 0.0) eo = 0.0;
        out[0].e = eo;
        out[0].sf = 1.0 - opt[0];                                   // :137-138
        return 1;
    }
    case TRC_OPT_REFLECTIVE_SPECTRAL: {                             // :183-193
        trc_reflect(dx, dy, dz, nx, ny, nz, &out[0].dx, &out[0].dy, &out[0].dz);
        int n = extra_len / 2;
        out[0].e = e * (1.0 - trc_interp(extra + extra_off, n, wl));
        return 1;
    }
    case TRC_OPT_REAL_REFLECTIVE:
    case TRC_OPT_ONE_SIDED_REAL_REFLECTIVE: {                       // :231-269, :498-504
        double sigma = opt[1];
        double rx = nx, ry = ny, rz = nz;
        if (sigma > 0.0) {
            double u0, u1, u2 = 0.0, u3, g0, g1;
            trc_uniform_pair(seed, rid, event, 0, &u0, &u1);
            trc_normal_pair(u0, u1, &g0, &g1);
            bool bi = opt[2] != 0.0;
            if (!bi) trc_uniform_pair(seed, rid, event, 1, &u2, &u3);
            double ex, ey, ez;
            trc_slope_error_local(sigma, bi, g0, g1, u2, &ex, &ey, &ez);
            trc_rotate_z_to_normal(ex, ey, ez, nx, ny, nz, &rx, &ry, &rz);
            double inv = 1.0 / sqrt(rx * rx + ry * ry + rz * rz);
            rx *= inv; ry *= inv; rz *= inv;
        }
        trc_reflect(dx, dy, dz, rx, ry, rz, &out[0].dx, &out[0].dy, &out[0].dz);
        double eo = e * (1.0 - opt[0]);
        if (FULL) eo *= trc_iam(opt[3], opt[4], dx, dy, dz, nx, ny, nz);       // RealReflective_IAM :320-329 (ideal normal)
        if (opt_kind == TRC_OPT_ONE_SIDED_REAL_REFLECTIVE && (dx * ux + dy * uy + dz * uz) > 0.0) eo = 0.0;
        out[0].e = eo;
        out[0].sf = 1.0 - opt[0];                                   // :266-267
        return 1;
    }
    case TRC_OPT_LAMBERTIAN: {                                      // :154-176
        double u0, u1, ax, ay, az;
        trc_uniform_pair(seed, rid, event, 0, &u0, &u1);
        trc_pillbox_dir_u(u0, u1, opt[1], &ax, &ay, &az);
        trc_rotation_to_z_apply(nx, ny, nz, ax, ay, az, &out[0].dx, &out[0].dy, &out[0].dz);
        if (FULL && opt[2] != 0.0) out[0].e = e * exp(-opt[2] * (path * opt[3])) * (1.0 - opt[0]);      // LambertianAbsorbant :895-906
        else {
            out[0].e = e * (1.0 - opt[0]);
            if (FULL) out[0].e *= trc_iam(opt[4], opt[5], dx, dy, dz, nx, ny, nz);   // Lambertian_IAM :302-318
            out[0].sf = 1.0 - opt[0];                               // :173-174 (LambertianAbsorbant builds its Lambertian with 0, :897)
        }
        return 1;
    }
    case TRC_OPT_SEMI_LAMBERTIAN: {                                 // :514-531 as documented (:507-509)
        // incidence angle from the oriented normal; glancing rays are mirrored (block 0), the others scattered (block 1):
        // `outg = specular + diffuse` (:531)
        double ang = acos(-(dx * nx + dy * ny + dz * nz));
        if (ang > opt[1]) {
            trc_reflect(dx, dy, dz, nx, ny, nz, &out[0].dx, &out[0].dy, &out[0].dz);
        } else {
            double u0, u1, ax, ay, az;
            trc_uniform_pair(seed, rid, event, 0, &u0, &u1);
            trc_pillbox_dir_u(u0, u1, opt[1], &ax, &ay, &az);
            trc_rotation_to_z_apply(nx, ny, nz, ax, ay, az, &out[0].dx, &out[0].dy, &out[0].dz);
            out[0].blk = 1;
        }
        out[0].e = e * (1.0 - opt[0]);
        return 1;
    }
    case TRC_OPT_LAMBERTIAN_SPECULAR: {                             // :561-585
        double u0, u1, u2, u3;
        trc_uniform_pair(seed, rid, event, 0, &u0, &u1);
        if (u0 < opt[1]) {
            trc_reflect(dx, dy, dz, nx, ny, nz, &out[0].dx, &out[0].dy, &out[0].dz);
        } else {
            double ax, ay, az;
            trc_uniform_pair(seed, rid, event, 1, &u2, &u3);
            trc_pillbox_dir_u(u1, u2, 1.57079632679489661923, &ax, &ay, &az);
            trc_rotation_to_z_apply(nx, ny, nz, ax, ay, az, &out[0].dx, &out[0].dy, &out[0].dz);
        }
        out[0].e = e * (1.0 - opt[0]);
        return 1;
    }
    case TRC_OPT_LAMBERTIAN_DIRECTIONAL:
    case TRC_OPT_LAMBERTIAN_DIRECTIONAL_SPECTRAL: {                 // :340-361, :373-391
        double dn = dx * nx + dy * ny + dz * nz;
        double wx = dn * nx, wy = dn * ny, wz = dn * nz;             // "vertical" component of the incident direction
        double th = acos(sqrt(wx * wx + wy * wy + wz * wz));
        // opt[0]: 0 Lambertian; 1 specular with probability opt[1] (LambertianSpecular_directional_..., :427-455); 2 with a
        // probability tabulated on the incidence angle too, third column of the table (Lambertian_piecewise_Specular_..., :457-487)
        const int mode = (opt_kind == TRC_OPT_LAMBERTIAN_DIRECTIONAL) ? (int)opt[0] : 0;
        const int ncol = mode == 2 ? 3 : 2;
        double ab = (opt_kind == TRC_OPT_LAMBERTIAN_DIRECTIONAL) ? trc_interp(extra + extra_off, extra_len / ncol, th)
                                                                  : trc_interp2(extra + extra_off, th, wl);
        double u0, u1, ax, ay, az;
        trc_uniform_pair(seed, rid, event, 0, &u0, &u1);
        if (mode == 0) {
            trc_pillbox_dir_u(u0, u1, 1.57079632679489661923, &ax, &ay, &az);
            trc_rotation_to_z_apply(nx, ny, nz, ax, ay, az, &out[0].dx, &out[0].dy, &out[0].dz);
        } else {
            const int n = extra_len / ncol;
            const double spec = mode == 1 ? opt[1] : trc_interp_xy(extra + extra_off, extra + extra_off + 2 * n, n, th);
            if (u0 < spec) {
                trc_reflect(dx, dy, dz, nx, ny, nz, &out[0].dx, &out[0].dy, &out[0].dz);
            } else {
                double u2, u3;
                trc_uniform_pair(seed, rid, event, 1, &u2, &u3);
                trc_pillbox_dir_u(u1, u2, 1.57079632679489661923, &ax, &ay, &az);
                trc_rotation_to_z_apply(nx, ny, nz, ax, ay, az, &out[0].dx, &out[0].dy, &out[0].dz);
            }
        }
        out[0].e = e * (1.0 - ab);
        if (opt_kind == TRC_OPT_LAMBERTIAN_DIRECTIONAL && mode == 0) out[0].sf = 1.0 - ab;       // :358-359
        return 1;
    }
    case TRC_OPT_FRESNEL_CONDUCTOR: {                               // :1536-1558
        int n = extra_len / 3;
        const double *tab = extra + extra_off;
        double n2, k2;
        {   // interp1d of the complex index: linear in n and k separately
            const double *xs = tab;
            double w;
            int i = trc_grid_cell(xs, n, wl, &w);
            n2 = (1.0 - w) * tab[n + i] + w * tab[n + i + 1];
            k2 = (1.0 - w) * tab[2 * n + i] + w * tab[2 * n + i + 1];
        }
        double R = trc_fresnel_conductor(fabs(dx * nx + dy * ny + dz * nz), opt[0], n2, k2);
        trc_reflect(dx, dy, dz, nx, ny, nz, &out[0].dx, &out[0].dy, &out[0].dz);
        out[0].e = e * R;
        return 1;
    }
    case TRC_OPT_REFRACTIVE_SCATTERING: {
        // Scattering in the medium the ray arrives through (optics_callables.py:946-1036 as its docstrings and the body kept in
        // comments at :1385-1470 describe it; the class itself does not run in the reference): a free path l = -ln(R) / s_c is
        // drawn (optics.py:214-239; s_c = 0 never scatters); l < path: the ray is scattered at prev + l d into a direction drawn
        // from the medium's Henyey-Greenstein function about d (sampling.py:150-168, rotate_z_to_normal as :1012-1013), energy
        // and index unchanged.  Otherwise it reaches the surface: RefractiveHomogenous below.  The medium is told by the index
        // the ray carries (the reference tells it by the scattering coefficient it carries, toggled with the index).
        // Draws: block 2 = (R, R_hg), block 3 = (azimuth, -); blocks 0-1 are the refraction's.
        const double *x = extra + extra_off;
        const int med = (ref == opt[0]) ? 0 : 1;
        const double s_c = extra_len >= 4 ? x[med] : 0.0, g = extra_len >= 4 ? x[2 + med] : 0.0;
        double r0, r1, r2, r3;
        trc_uniform_pair(seed, rid, event, 2, &r0, &r1);
        trc_uniform_pair(seed, rid, event, 3, &r2, &r3);
        (void)r3;
        const double l = s_c != 0.0 ? -log(r0) / s_c : path;        // optics.py:229-230
        if (l < path) {                                             // :233
            const double th = trc_hg_theta(g, r1);
            const double ph = TRC_TWO_PI * r2;
            double st, ct, sp, cp;
            trc_sincos(th, &st, &ct);
            trc_sincos(ph, &sp, &cp);
            trc_rotate_z_to_normal(st * cp, st * sp, ct, dx, dy, dz, &out[0].dx, &out[0].dy, &out[0].dz);
            out[0].e = e; out[0].ref = ref; out[0].blk = 0;
            out[0].back = path - l;
            return 1;
        }
        const int n_out = trc_shade_refractive(opt, dx, dy, dz, e, ref, path, nx, ny, nz, seed, rid, event, out);
        out[0].blk += 1; out[1].blk += 1;          // behind the scattered block
        return n_out;
    }
    case TRC_OPT_REFRACTIVE_HOMOGENOUS:                             // :1226-1296 on :836-858
        return trc_shade_refractive(opt, dx, dy, dz, e, ref, path, nx, ny, nz, seed, rid, event, out);
    default:
        out[0].dx = dx; out[0].dy = dy; out[0].dz = dz; out[0].e = 0.0;
        return 1;
    }
}

TRC_HD int trc_shade(int opt_kind, const double *opt, const double *extra, int extra_off, int extra_len,
                     double ux, double uy, double uz, double dx, double dy, double dz, double e, double ref,
                     double wl, double path, double nx, double ny, double nz, uint64_t seed, uint64_t rid,
                     uint32_t event, trc_ray_out out[2]) {
    return trc_shade_k<0xFFFFFFFFu, true>(opt_kind, opt, extra, extra_off, extra_len, ux, uy, uz, dx, dy, dz, e, ref, wl, path, nx, ny, nz,
                                          seed, rid, event, out);
}


// ---------------------------------------------------------------------------------------------
// Rays of the ordered engine and of the per-surface protocol carry more than the fast engines' record: the imaginary part of a
// complex refractive index (media that attenuate) and, for polychromatic bundles, a sampled spectrum.  trc_shade_x is trc_shade
// plus the optics that read them.
// ---------------------------------------------------------------------------------------------
struct trc_ray_ext {
    double ref_im;                 // Im of the index of the medium the ray travels in
    int W;                         // samples of the spectrum the ray carries (0: none)
    int n_mat;                     // materials evaluated at this ray's wavelength
    const double *wl, *spec;       // sample w at wl[w * stride], spec[w * stride]
    const double *mat;             // Re, Im of material k at mat[2k * stride], mat[(2k + 1) * stride]
    long long stride;
};

struct trc_cplx { double re, im; };
TRC_HD trc_cplx trc_c(double re, double im) { trc_cplx z; z.re = re; z.im = im; return z; }
TRC_HD trc_cplx trc_cadd(trc_cplx a, trc_cplx b) { return trc_c(a.re + b.re, a.im + b.im); }
TRC_HD trc_cplx trc_csub(trc_cplx a, trc_cplx b) { return trc_c(a.re - b.re, a.im - b.im); }
TRC_HD trc_cplx trc_cmul(trc_cplx a, trc_cplx b) { return trc_c(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
TRC_HD trc_cplx trc_cscale(trc_cplx a, double s) { return trc_c(a.re * s, a.im * s); }
TRC_HD trc_cplx trc_cdiv(trc_cplx a, trc_cplx b) {         // Smith's form, what numpy's complex division does
    if (fabs(b.re) >= fabs(b.im)) {
        if (b.re == 0.0 && b.im == 0.0) return trc_c(a.re / fabs(b.re), a.im / fabs(b.im));
        const double r = b.im / b.re, den = 1.0 / (b.re + b.im * r);
        return trc_c((a.re + a.im * r) * den, (a.im - a.re * r) * den);
    }
    const double r = b.re / b.im, den = 1.0 / (b.re * r + b.im);
    return trc_c((a.re * r + a.im) * den, (a.im * r - a.re) * den);
}
TRC_HD trc_cplx trc_csqrt(trc_cplx z) {                     // principal root
    if (z.im == 0.0) return z.re >= 0.0 ? trc_c(sqrt(z.re), z.im) : trc_c(0.0, copysign(sqrt(-z.re), z.im));
    const double m = hypot(z.re, z.im);
    if (z.re >= 0.0) { const double t = sqrt(0.5 * (m + z.re)); return trc_c(t, z.im / (2.0 * t)); }
    const double t = sqrt(0.5 * (m - z.re));
    return trc_c(fabs(z.im) / (2.0 * t), copysign(t, z.im));
}

// Re of optics.fresnel (optics.py:13-39) evaluated with complex indices, as Refractive._make_refraction_bundle does (:838-840:
// the complex reflectance is stored into a real array, numpy keeps its real part).  Squares, not squared moduli: reproduced.
TRC_HD double trc_fresnel_complex_re(double cos_abs, trc_cplx n1, trc_cplx n2) {
    const double th = acos(cos_abs);
    const double foo = cos(th), sn = sin(th);
    trc_cplx q = trc_cscale(trc_cdiv(n1, n2), sn);
    trc_cplx bar = trc_csqrt(trc_csub(trc_c(1.0, 0.0), trc_cmul(q, q)));
    trc_cplx a = trc_cscale(n1, foo), b = trc_cmul(n2, bar);
    trc_cplx rs = trc_cdiv(trc_csub(a, b), trc_cadd(a, b));
    trc_cplx c = trc_cmul(n1, bar), d = trc_cscale(n2, foo);
    trc_cplx rp = trc_cdiv(trc_csub(c, d), trc_cadd(c, d));
    rs = trc_cmul(rs, rs);
    rp = trc_cmul(rp, rp);
    return 0.5 * (rs.re + rp.re);
}

// Refractive (optics_callables.py:726-858) and RefractiveAbsorbant (:908-944)
// mat0, mat1: materials[0].m(lambda), materials[1].m(lambda) at this ray's wavelength (evaluated by the caller of the C-ABI with the
// material objects themselves, trc_rays.mat: tables, Sopra files and analytic models alike, and the comparison :750 stays exact)
TRC_HD int trc_shade_material(const double *opt, trc_cplx mat0, trc_cplx mat1, double dx, double dy, double dz, double e, double ref,
                              double ref_im, double wl, double path, double nx, double ny, double nz, uint64_t seed, uint64_t rid,
                              uint32_t event, trc_ray_out out[2], double out_im[2]) {
    const bool single = opt[0] != 0.0;
    const double sigma = opt[1];
    double u0, u1, u2, u3;
    trc_uniform_pair(seed, rid, event, 0, &u0, &u1);
    trc_uniform_pair(seed, rid, event, 1, &u2, &u3);
    if (sigma >= 0.0) {                                         // normal perturbation :767-781
        double g0, g1;
        trc_normal_pair(u0, u1, &g0, &g1);
        double th = sigma * g0, phi = TRC_TWO_PI * u2;
        double st, ct, sp, cp;
        trc_sincos(th, &st, &ct);
        trc_sincos(phi, &sp, &cp);
        double rx, ry, rz;
        trc_rotation_to_z_apply(nx, ny, nz, st * cp, st * sp, ct, &rx, &ry, &rz);
        nx = rx; ny = ry; nz = rz;
    }
    const trc_cplx m1 = trc_c(ref, ref_im);
    const trc_cplx m2 = (m1.re == mat0.re && m1.im == mat0.im) ? mat1 : mat0;     // toggle_ref_idx :750-751
    const double eta = m2.re / m1.re;                           // refractions(m1.real, m2.real, ...) :786
    const double cos1 = nx * dx + ny * dy + nz * dz;
    const bool refracted = (cos1 * cos1) >= (1.0 - eta * eta);
    double R = 1.0;
    double tx = 0.0, ty = 0.0, tz = 0.0;
    if (refracted) {
        tx = (dx - cos1 * nx) / eta; ty = (dy - cos1 * ny) / eta; tz = (dz - cos1 * nz) / eta;
        const double cos2 = sqrt(1.0 - 1.0 / (eta * eta) * (1.0 - cos1 * cos1));
        const double sg = (cos1 < 0.0) ? -1.0 : 1.0;
        tx += nx * cos2 * sg; ty += ny * cos2 * sg; tz += nz * cos2 * sg;
        R = trc_fresnel_complex_re(fabs(cos1), m1, m2);
    }
    int n_out;
    out[0].ref = ref; out_im[0] = ref_im;
    if (single) {                                               // :796-823
        if (u3 <= R) {
            trc_reflect(dx, dy, dz, nx, ny, nz, &out[0].dx, &out[0].dy, &out[0].dz);
            out[0].e = e;
        } else {
            out[0].dx = tx; out[0].dy = ty; out[0].dz = tz;
            out[0].e = e; out[0].ref = m2.re; out_im[0] = m2.im; out[0].blk = 1;
        }
        n_out = 1;
    } else {                                                    // :825-835
        trc_reflect(dx, dy, dz, nx, ny, nz, &out[0].dx, &out[0].dy, &out[0].dz);
        out[0].e = e * R;
        n_out = 1;
        if (refracted) {
            out[1].dx = tx; out[1].dy = ty; out[1].dz = tz;
            out[1].e = e * (1.0 - R); out[1].ref = m2.re; out_im[1] = m2.im;
            n_out = 2;
        }
    }
    if (opt[2] != 0.0)          // Absorbant.attenuate :874-882 with a_c None: k and the wavelength of the NEW ray, the path of the old
        for (int c = 0; c < n_out; ++c)
            out[c].e = exp(-4.0 * TRC_PI * (path * opt[3]) * out_im[c] / wl) * out[c].e;      // optics.py:210-211
    return n_out;
}

// absorptance of the polychromatic Lambertian wall at sample wavelength wl_w (RegularGridInterpolator over (theta, lambda), :399-408)
TRC_HD double trc_poly_absorptance(const double *tab, double th, double wl_w) { return trc_interp2(tab, th, wl_w); }

// One interaction with everything a ray can carry.  Returns the number of outgoing rays; out_im[c]: Im of their index;
// *poly_th >= 0: the surface is a polychromatic wall and sample w of the spectrum is scaled by 1 - trc_poly_absorptance(tab, *poly_th,
// wl_w) (the energy in out[0].e is already the trapezoid integral of that); otherwise the whole spectrum by out[c].sf.
TRC_HD int trc_shade_x(int opt_kind, const double *opt, const double *extra, int extra_off, int extra_len,
                       double ux, double uy, double uz, double dx, double dy, double dz, double e, double ref,
                       double wl, double path, double nx, double ny, double nz, uint64_t seed, uint64_t rid,
                       uint32_t event, const trc_ray_ext &X, trc_ray_out out[2], double out_im[2], double *poly_th) {
    *poly_th = -1.0;
    out_im[0] = out_im[1] = X.ref_im;
    if (opt_kind == TRC_OPT_REFRACTIVE_MATERIAL) {
        out[0].blk = 0; out[1].blk = 1; out[0].back = out[1].back = 0.0; out[0].sf = out[1].sf = 1.0; out[0].shift = out[1].shift = 0.0;
        const int k0 = (int)opt[4], k1 = (int)opt[5];
        trc_cplx m0 = trc_c(NAN, NAN), m1 = m0;
        if (X.mat && k0 < X.n_mat && k1 < X.n_mat) {
            m0 = trc_c(X.mat[(long long)(2 * k0) * X.stride], X.mat[(long long)(2 * k0 + 1) * X.stride]);
            m1 = trc_c(X.mat[(long long)(2 * k1) * X.stride], X.mat[(long long)(2 * k1 + 1) * X.stride]);
        }
        return trc_shade_material(opt, m0, m1, dx, dy, dz, e, ref, X.ref_im, wl, path, nx, ny, nz, seed, rid, event, out, out_im);
    }
    if (opt_kind == TRC_OPT_LAMBERTIAN_POLYCHROMATIC) {         // :406-425
        out[0].blk = 0; out[1].blk = 1; out[0].back = out[1].back = 0.0; out[0].sf = out[1].sf = 1.0; out[0].shift = out[1].shift = 0.0;
        out[0].ref = ref;
        const double dn = dx * nx + dy * ny + dz * nz;
        const double wx = dn * nx, wy = dn * ny, wz = dn * nz;
        const double th = acos(sqrt(wx * wx + wy * wy + wz * wz));
        const double *tab = extra + extra_off;
        double en = 0.0, y0 = 0.0, x0 = 0.0;                    // N.trapz(spectra, wavelengths, axis=0) :413
        for (int w = 0; w < X.W; ++w) {
            const double xw = X.wl[(long long)w * X.stride];
            const double yw = X.spec[(long long)w * X.stride] * (1.0 - trc_poly_absorptance(tab, th, xw));
            if (w > 0) en += (xw - x0) * (yw + y0) / 2.0;
            x0 = xw; y0 = yw;
        }
        double u0, u1, ax, ay, az;
        trc_uniform_pair(seed, rid, event, 0, &u0, &u1);
        trc_pillbox_dir_u(u0, u1, 1.57079632679489661923, &ax, &ay, &az);
        trc_rotation_to_z_apply(nx, ny, nz, ax, ay, az, &out[0].dx, &out[0].dy, &out[0].dz);
        out[0].e = en;
        *poly_th = th;
        return 1;
    }
    return trc_shade(opt_kind, opt, extra, extra_off, extra_len, ux, uy, uz, dx, dy, dz, e, ref, wl, path, nx, ny, nz, seed, rid,
                     event, out);
}

// ---------------------------------------------------------------------------------------------
// S1..S3 -- sources.  One ray from its four uniforms.
// ---------------------------------------------------------------------------------------------
// x^y for a positive normal x whose result is a normal number: log2 of the mantissa in [sqrt(1/2), sqrt(2)) by the atanh
// series, 2^t by the Taylor series of e^(f ln 2) on |f| <= 1/2.  Relative error below 1.5e-15 for |y log2 x| < 16 (checked
// against pow() in tests/hostcheck); ~70 instructions where the library pow(), with its special cases and its
// extended-precision logarithm, is ~750 -- and a wave pays for it whenever one of its 64 rays falls in the aureole.
TRC_HD double trc_pow_pos(double x, double y) {
    uint64_t bits = __builtin_bit_cast(uint64_t, x);
    int e = (int)(bits >> 52) - 1023;
    double m = __builtin_bit_cast(double, (bits & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);      // [1, 2)
    if (m > 1.4142135623730951) { m *= 0.5; e += 1; }
    double s = (m - 1.0) / (m + 1.0), s2 = s * s;                   // ln m = 2 atanh(s), |s| <= 0.1716
    double q = 1.0 / 23.0;
    q = q * s2 + 1.0 / 21.0; q = q * s2 + 1.0 / 19.0; q = q * s2 + 1.0 / 17.0; q = q * s2 + 1.0 / 15.0;
    q = q * s2 + 1.0 / 13.0; q = q * s2 + 1.0 / 11.0; q = q * s2 + 1.0 / 9.0;  q = q * s2 + 1.0 / 7.0;
    q = q * s2 + 1.0 / 5.0;  q = q * s2 + 1.0 / 3.0;
    double ln_m = 2.0 * s + 2.0 * s * s2 * q;
    double l2m = ln_m * 1.4426950408889634;
    double n = rint(y * ((double)e + l2m));                         // y log2 x = n + f, the exponent's part of f exactly
    double z = (fma(y, (double)e, -n) + y * l2m) * 0.6931471805599453;      // |z| <= 0.347
    double r = 1.0 / 6227020800.0;                                  // e^z, terms to z^13/13! (4e-18)
    r = r * z + 1.0 / 479001600.0; r = r * z + 1.0 / 39916800.0; r = r * z + 1.0 / 3628800.0; r = r * z + 1.0 / 362880.0;
    r = r * z + 1.0 / 40320.0; r = r * z + 1.0 / 5040.0; r = r * z + 1.0 / 720.0; r = r * z + 1.0 / 120.0;
    r = r * z + 1.0 / 24.0; r = r * z + 1.0 / 6.0; r = r * z + 0.5; r = r * z + 1.0; r = r * z + 1.0;
    return r * __builtin_bit_cast(double, (uint64_t)((int64_t)n + 1023) << 52);
}

// Buie sunshape polar angle from its uniform (sources.py:364-377).  tab: trc_source_desc.buie
// ray-independent parts of the aureole inversion (:377); kernels compute them once per block
TRC_HD void trc_buie_aureole_consts(const double *tab, double *c) {
    const double *sc = tab + 3 * (TRC_BUIE_NELEM + 1);
    double I_dni = sc[0], gamma = sc[1], kappa = sc[2], theta_dni = sc[3], theta_tot = sc[4];
    double gp2 = gamma + 2.0;
    c[0] = gp2 / (pow(10.0, 3.0 * gamma) * exp(kappa)) * I_dni - pow(theta_dni, gp2);
    c[1] = pow(theta_tot, gp2);
    c[2] = 1.0 / gp2;
}

// aur: the three values of trc_buie_aureole_consts, or null to compute them here
TRC_HD double trc_buie_theta(const double *tab, const double *aur, double Rv) {
    const int NE = TRC_BUIE_NELEM;
    const double *theta = tab, *g = tab + (NE + 1), *cdf = tab + 2 * (NE + 1);
    const double *sc = tab + 3 * (NE + 1);
    double I_dni = sc[0];
    bool csr_pos = sc[5] != 0.0;
    if (Rv < cdf[NE]) {
        int lo = 0, hi = NE;                       // largest i with cdf[i] <= R
        while (hi - lo > 1) {
            int mid = (lo + hi) >> 1;
            if (cdf[mid] <= Rv) lo = mid; else hi = mid;
        }
        int i = lo;
        double A = g[i], B = g[i + 1];
        double Cq = 2.0 * I_dni * (Rv - cdf[i]) * (theta[i + 1] - theta[i]);                   // :370
        double w = (theta[i] - theta[i + 1]) * A;
        return -(-A * theta[i + 1] + B * theta[i] + sqrt(w * w + Cq * (B - A))) / (A - B);     // :371
    }
    if (!csr_pos) return 0.0;                      // thetas stay 0 (sources.py:364, :376)
    double c[3];                                   // aureole, :377
    if (aur) { c[0] = aur[0]; c[1] = aur[1]; c[2] = aur[2]; }
    else trc_buie_aureole_consts(tab, c);
    double base = (Rv - 1.0) * c[0] + Rv * c[1];
    return trc_pow_pos(base, c[2]);
}

// The same inversion arranged for a kernel's LDS (the streaming engine's generation kernel): the bin is found from a
// 1024-entry first guess over the uniform and a short forward scan instead of an 8-step bisection of dependent LDS
// reads, and everything of :370-371 that does not depend on the ray is folded per bin into
//     theta = a_i + k_i sqrt(w2_i + q_i (R - cdf_i)),   a_i = (A th_{i+1} - B th_i)/(A - B),  k_i = -1/(A - B),
//     w2_i = ((th_i - th_{i+1}) A)^2,  q_i = 2 I_dni (th_{i+1} - th_i)(B - A)
// -- the reference's expression with its division replaced by a multiplication (a few ulp apart; compared with
// trc_buie_theta in tests/hostcheck).  122 -> ~45 VALU instructions per ray.
#define TRC_BUIE_GUESS 1024
struct trc_buie_fast {
    double cdf[TRC_BUIE_NELEM + 1];
    double rec[TRC_BUIE_NELEM][4];     // a, k, w2, q
    double aur[3];                     // trc_buie_aureole_consts
    double cdf_end;
    int csr_pos;
    uint8_t guess[TRC_BUIE_GUESS];     // largest bin i (<= NELEM-1) with cdf[i] <= k / TRC_BUIE_GUESS
};
static_assert(TRC_BUIE_NELEM <= 256, "bin numbers are stored in bytes");

// cooperative fill: thread `tid` of `nth` (host: 0 of 1).  Phase 0 copies the cdf, phase 1 (after a barrier on the device)
// derives the rest from it.
TRC_HD void trc_buie_fast_fill(const double *tab, trc_buie_fast *F, int phase, int tid, int nth) {
    const int NE = TRC_BUIE_NELEM;
    const double *theta = tab, *g = tab + (NE + 1), *cdf = tab + 2 * (NE + 1), *sc = tab + 3 * (NE + 1);
    if (phase == 0) {
        for (int i = tid; i <= NE; i += nth) F->cdf[i] = cdf[i];
        if (tid == 0) { trc_buie_aureole_consts(tab, F->aur); F->cdf_end = cdf[NE]; F->csr_pos = sc[5] != 0.0 ? 1 : 0; }
        return;
    }
    const double I_dni = sc[0];
    for (int i = tid; i < NE; i += nth) {
        double A = g[i], B = g[i + 1], t0 = theta[i], t1 = theta[i + 1];
        double w = (t0 - t1) * A;
        F->rec[i][0] = (A * t1 - B * t0) / (A - B);
        F->rec[i][1] = -1.0 / (A - B);
        F->rec[i][2] = w * w;
        F->rec[i][3] = 2.0 * I_dni * (t1 - t0) * (B - A);
    }
    for (int k = tid; k < TRC_BUIE_GUESS; k += nth) {
        const double Rv = (double)k * (1.0 / TRC_BUIE_GUESS);
        int lo = 0, hi = NE;                       // largest i in [0, NE-1] with cdf[i] <= Rv (cdf[0] = 0)
        while (hi - lo > 1) {
            int mid = (lo + hi) >> 1;
            if (F->cdf[mid] <= Rv) lo = mid; else hi = mid;
        }
        F->guess[k] = (uint8_t)lo;
    }
}

TRC_HD double trc_buie_theta_fast(const trc_buie_fast *F, double Rv) {
    if (Rv < F->cdf_end) {
        int i = F->guess[(int)(Rv * (double)TRC_BUIE_GUESS)];
        while (F->cdf[i + 1] <= Rv) ++i;           // ends: cdf[NELEM] = cdf_end > Rv
        const double *r = F->rec[i];
        return r[0] + r[1] * sqrt(r[2] + r[3] * (Rv - F->cdf[i]));
    }
    if (!F->csr_pos) return 0.0;
    return trc_pow_pos((Rv - 1.0) * F->aur[0] + Rv * F->aur[1], F->aur[2]);
}

// buie: the table of the descriptor (or its LDS copy); aur: trc_buie_aureole_consts of it, or null; bf: the
// trc_buie_fast form of the table when the caller has staged one (then buie and aur are not read).
// KIND >= 0 promises src->kind == KIND (the streaming engine compiles the Buie disc on its own: 158 instead of 237
// VGPRs); KIND < 0 reads the kind from the descriptor.
template <int KIND>
TRC_HD void trc_source_ray_t(const trc_source_desc *src, const double *buie, const double *aur, uint64_t seed, uint64_t rid,
                             double *px, double *py, double *pz, double *dx, double *dy, double *dz,
                             const trc_buie_fast *bf = nullptr) {
    double u0, u1, u2, u3;
    trc_uniform_quad(seed, rid, 0, 0, &u0, &u1, &u2, &u3);      // event 0 = source generation
    double lx, ly, lz = 0.0, ax, ay, az;
    const double *p = src->p;
    const int kind = KIND >= 0 ? KIND : src->kind;
    switch (kind) {
    case TRC_SRC_VF_CYLINDER:           // draws: zs, phi_s, dir phi, dir R (sources.py:737-746)
    case TRC_SRC_VF_FRUSTUM: {          // draws: dir phi, dir R, R, phi_s (sources.py:670-685)
        double phi, slope, sign, fx, fy, fz;
        if (kind == TRC_SRC_VF_CYLINDER) {
            lz = p[1] * u0 - p[1] / 2.0;
            phi = p[2] + (p[3] - p[2]) * u1;
            { double sp, cp; trc_sincos(phi, &sp, &cp); lx = p[0] * cp; ly = p[0] * sp; }
            trc_pillbox_dir_u(u2, u3, p[4], &fx, &fy, &fz);
            slope = 0.0; sign = p[5];
        } else {
            trc_pillbox_dir_u(u0, u1, p[5], &fx, &fy, &fz);
            slope = (p[1] - p[0]) / p[2];
            double rs = sqrt((p[1] * p[1] - p[0] * p[0]) * u2 + p[0] * p[0]);
            lz = (rs - p[0]) / slope;
            phi = p[3] + (p[4] - p[3]) * u3;
            { double sp, cp; trc_sincos(phi, &sp, &cp); lx = rs * cp; ly = rs * sp; }
            sign = p[6];
        }
        // local_unit = rotz(phi) . roty(-pi/2 + atan(slope)) . dir_flat   (:687-695, :748-753)
        double trot = -TRC_PI / 2.0 + atan(slope);
        double cy, sy, cz, sz;
        trc_sincos(trot, &sy, &cy);
        trc_sincos(phi, &sz, &cz);
        double rx = cy * fx + sy * fz, ry = fy, rz = -sy * fx + cy * fz;
        ax = sign * (cz * rx - sz * ry); ay = sign * (sz * rx + cz * ry); az = sign * rz;
        break;
    }
    case TRC_SRC_PILLBOX_DISK: {        // draws: dir phi, dir R, pos xi, pos theta (sources.py:200-213)
        trc_pillbox_dir_u(u0, u1, p[4], &ax, &ay, &az);
        double r = sqrt(p[1] * p[1] + u2 * (p[0] * p[0] - p[1] * p[1]));
        double th = p[2] + (p[3] - p[2]) * u3;
        { double st, ct; trc_sincos(th, &st, &ct); lx = r * ct; ly = r * st; }
        if (p[5] != 0.0) {              // x_cut: redraw the position until x < x_cut (rejection, sources.py:216-228)
            for (uint32_t blk = 2; !(lx < p[6]) && blk < 2 + 4096; ++blk) {
                trc_uniform_pair(seed, rid, 0, blk, &u2, &u3);
                r = sqrt(p[1] * p[1] + u2 * (p[0] * p[0] - p[1] * p[1]));
                th = p[2] + (p[3] - p[2]) * u3;
                double st, ct;
                trc_sincos(th, &st, &ct);
                lx = r * ct; ly = r * st;
            }
        }
        break;
    }
    case TRC_SRC_PILLBOX_RECT: {        // draws: dir phi, dir R, xs, ys (sources.py:243-256)
        trc_pillbox_dir_u(u0, u1, p[2], &ax, &ay, &az);
        double xs = -p[0] / 2.0 + p[0] * u2, ys = -p[1] / 2.0 + p[1] * u3;
        if (p[3] != 0.0) { double tmp = xs; xs = ys; ys = tmp; }
        lx = ys; ly = xs;               // vertices_local = (ys, xs, 0)
        break;
    }
    case TRC_SRC_BUIE_DISK: {           // draws: xv1, phiv, R_theta, xi (sources.py:431-434, :365, :380)
        double r = p[0] * sqrt(u0);
        double sph, cph, st, ct, sxi, cxi;
        trc_sincos_2pi(u1, &sph, &cph);
        lx = r * cph; ly = r * sph;
        double th = bf ? trc_buie_theta_fast(bf, u2) : trc_buie_theta(buie, aur, u2);
        trc_sincos_small(th, &st, &ct);
        trc_sincos_2pi(u3, &sxi, &cxi);
        ax = cxi * st; ay = sxi * st; az = ct;
        break;
    }
    case TRC_SRC_PILLBOX_TRIANGLE: {    // draws: r1, r2 (point picking), dir phi, dir R (sources.py:559-568)
        double sq = sqrt(u0);
        lx = sq * (1.0 - u1); ly = u1 * sq;     // A + sqrt(r1)(1-r2) AB + r2 sqrt(r1) AC; AB, AC are the columns of rot_pos
        trc_pillbox_dir_u(u2, u3, p[0], &ax, &ay, &az);
        break;
    }
    default: {                          // TRC_SRC_BUIE_RECT (sources.py:485-486)
        lx = p[0] * (u0 - 0.5); ly = p[1] * (u1 - 0.5);
        double th = bf ? trc_buie_theta_fast(bf, u2) : trc_buie_theta(buie, aur, u2);
        double st, ct, sxi, cxi;
        trc_sincos_small(th, &st, &ct);
        trc_sincos_2pi(u3, &sxi, &cxi);
        ax = cxi * st; ay = sxi * st; az = ct;
        break;
    }
    }
    const double *rp = src->rot_pos, *rd = src->rot_dir;
    if (kind >= TRC_SRC_VF_CYLINDER) {      // wall emitters have a third local coordinate
        *px = rp[0] * lx + rp[1] * ly + rp[2] * lz + src->center[0];
        *py = rp[3] * lx + rp[4] * ly + rp[5] * lz + src->center[1];
        *pz = rp[6] * lx + rp[7] * ly + rp[8] * lz + src->center[2];
    } else {
        *px = rp[0] * lx + rp[1] * ly + src->center[0];
        *py = rp[3] * lx + rp[4] * ly + src->center[1];
        *pz = rp[6] * lx + rp[7] * ly + src->center[2];
    }
    *dx = rd[0] * ax + rd[1] * ay + rd[2] * az;
    *dy = rd[3] * ax + rd[4] * ay + rd[5] * az;
    *dz = rd[6] * ax + rd[7] * ay + rd[8] * az;
}

TRC_HD void trc_source_ray(const trc_source_desc *src, const double *buie, const double *aur, uint64_t seed, uint64_t rid,
                           double *px, double *py, double *pz, double *dx, double *dy, double *dz) {
    trc_source_ray_t<-1>(src, buie, aur, seed, rid, px, py, pz, dx, dy, dz);
}

// ---------------------------------------------------------------------------------------------
// O8 -- flux-map bin of a coordinate for numpy.histogram edges (right edge of the last bin closed)
// ---------------------------------------------------------------------------------------------
TRC_HD int trc_bin_index(const double *edges, int nbins, double x) {
    if (!(x >= edges[0]) || !(x <= edges[nbins])) return -1;
    int lo = 0, hi = nbins;      // largest i with edges[i] <= x
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (edges[mid] <= x) lo = mid; else hi = mid;
    }
    return lo;
}

#endif  // TRC_CORE_H
