// hostcheck.cpp -- TEST-ONLY host build of the per-ray core (tracer_amd/csrc/trc_core.h).
//
// The same header that hipcc compiles into the gfx950 kernels is compiled here by g++ so that the core math
// can be checked against the golden fixtures and the oracle in the CPU-only test tier (and under sanitizers)
// before GPU minutes are spent.  This library is built by `make hostcheck` into tests/hostcheck/, is loaded
// only by tests/test_hostcheck.py, and is NOT a product path: nothing under tracer_amd/ knows it exists and
// the product fails loudly without a GPU.
#include <vector>
#include <cstring>
#include "../../tracer_amd/csrc/trc_core.h"
#include "../../tracer_amd/csrc/trc_bounds.h"
#include "../../tracer_amd/csrc/trc_footprint.h"

static void pack_record(const trc_surface_desc &s, double *rec, int stride) {
    for (int i = 0; i < stride; ++i) rec[i] = 0.0;
    for (int r = 0; r < 3; ++r) {
        for (int k = 0; k < 3; ++k) rec[3 * r + k] = s.frame[4 * r + k];
        rec[9 + r] = s.frame[4 * r + 3];
    }
    int32_t *h = (int32_t *)(rec + 12);
    h[0] = s.gm_kind; h[1] = s.optics_kind; h[2] = s.extra_off; h[3] = s.extra_len;
    int np = trc_gm_nparams(s.gm_kind);
    for (int i = 0; i < np; ++i) rec[TRC_REC_HDR + i] = s.gm[i];
}

struct HostKdStack {
    int node[TRC_KD_STACK];
    double tmax[TRC_KD_STACK];
    void push(int sp, int n, double t) { node[sp] = n; tmax[sp] = t; }
    void pop(int sp, int *n, double *t) { *n = node[sp]; *t = tmax[sp]; }
};

extern "C" {

int hc_intersect(const trc_surface_desc *s, const double *extra, long n, const double *x, const double *y, const double *z,
                 const double *dx, const double *dy, const double *dz, double *t) {
    double rec[TRC_REC_HDR + 16];
    pack_record(*s, rec, TRC_REC_HDR + 16);
    for (long i = 0; i < n; ++i) t[i] = trc_intersect(rec, extra, x[i], y[i], z[i], dx[i], dy[i], dz[i]);
    return 0;
}

int hc_normals(const trc_surface_desc *s, long n, const double *hx, const double *hy, const double *hz, const double *dx,
               const double *dy, const double *dz, double *nx, double *ny, double *nz) {
    double rec[TRC_REC_HDR + 16];
    pack_record(*s, rec, TRC_REC_HDR + 16);
    for (long i = 0; i < n; ++i) trc_normal(rec, hx[i], hy[i], hz[i], dx[i], dy[i], dz[i], &nx[i], &ny[i], &nz[i]);
    return 0;
}

// shade: outputs 2n slots (child 0 at i, child 1 at n+i), blk = -1 when empty
// path: distance travelled to the hit per ray (attenuating optics), or null for 0
int hc_shade_path(const trc_surface_desc *s, const double *extra, long n, const double *dx, const double *dy, const double *dz,
                  const double *e, const double *ref, const double *wl, const double *nx, const double *ny, const double *nz,
                  const uint64_t *rid, uint64_t seed, int event, double *odx, double *ody, double *odz, double *oe, double *oref,
                  int *oblk, const double *path) {
    for (long i = 0; i < n; ++i) {
        trc_ray_out out[2];
        int no = trc_shade(s->optics_kind, s->opt, extra, s->extra_off, s->extra_len, s->frame[2], s->frame[6], s->frame[10],
                           dx[i], dy[i], dz[i], e[i], ref[i], wl[i], path ? path[i] : 0.0, nx[i], ny[i], nz[i], seed, rid[i],
                           (uint32_t)event, out);
        for (int c = 0; c < 2; ++c) {
            long slot = c == 0 ? i : n + i;
            if (c < no) {
                odx[slot] = out[c].dx; ody[slot] = out[c].dy; odz[slot] = out[c].dz; oe[slot] = out[c].e; oref[slot] = out[c].ref;
                oblk[slot] = out[c].blk;
            } else oblk[slot] = -1;
        }
    }
    return 0;
}

// trc_shade_x: the same with what rays of the ordered engine carry.  ref_im (n) or null; mat (2 n_mat x n) or null; swl, spec (W x n) or
// null; outputs o_im (2n), o_spec (W x 2n).
int hc_shade_x(const trc_surface_desc *s, const double *extra, long n, const double *dx, const double *dy, const double *dz,
               const double *e, const double *ref, const double *wl, const double *nx, const double *ny, const double *nz,
               const uint64_t *rid, uint64_t seed, int event, double *odx, double *ody, double *odz, double *oe, double *oref,
               int *oblk, const double *path, const double *ref_im, int n_mat, const double *mat, int W, const double *swl,
               const double *spec, double *o_im, double *o_spec) {
    for (long i = 0; i < n; ++i) {
        trc_ray_out out[2];
        trc_ray_ext X;
        X.ref_im = ref_im ? ref_im[i] : 0.0; X.W = W; X.n_mat = n_mat; X.stride = n;
        X.mat = mat ? mat + i : nullptr; X.wl = swl ? swl + i : nullptr; X.spec = spec ? spec + i : nullptr;
        double out_im[2], poly_th;
        int no = trc_shade_x(s->optics_kind, s->opt, extra, s->extra_off, s->extra_len, s->frame[2], s->frame[6], s->frame[10],
                             dx[i], dy[i], dz[i], e[i], ref[i], wl[i], path ? path[i] : 0.0, nx[i], ny[i], nz[i], seed, rid[i],
                             (uint32_t)event, X, out, out_im, &poly_th);
        for (int c = 0; c < 2; ++c) {
            long slot = c == 0 ? i : n + i;
            if (c < no) {
                odx[slot] = out[c].dx; ody[slot] = out[c].dy; odz[slot] = out[c].dz; oe[slot] = out[c].e; oref[slot] = out[c].ref;
                oblk[slot] = out[c].blk;
                if (o_im) o_im[slot] = out_im[c];
                for (int w = 0; w < W && o_spec; ++w) {
                    const double f = poly_th >= 0.0 ? 1.0 - trc_poly_absorptance(extra + s->extra_off, poly_th, X.wl[(long)w * n]) : out[c].sf;
                    o_spec[(long)w * 2 * n + slot] = X.spec[(long)w * n] * f;
                }
            } else oblk[slot] = -1;
        }
    }
    return 0;
}

// sizes of the C-ABI structs as the compiler lays them out (tests/test_abi.py compares the ctypes mirrors)
long hc_sizeof(int which) {
    switch (which) {
    case 0: return (long)sizeof(trc_surface_desc);
    case 1: return (long)sizeof(trc_rays);
    case 2: return (long)sizeof(trc_source_desc);
    case 3: return (long)sizeof(trc_kdtree_desc);
    case 4: return (long)sizeof(trc_trace_stats);
    default: return -1;
    }
}

int hc_shade(const trc_surface_desc *s, const double *extra, long n, const double *dx, const double *dy, const double *dz,
             const double *e, const double *ref, const double *wl, const double *nx, const double *ny, const double *nz,
             const uint64_t *rid, uint64_t seed, int event, double *odx, double *ody, double *odz, double *oe, double *oref,
             int *oblk) {
    return hc_shade_path(s, extra, n, dx, dy, dz, e, ref, wl, nx, ny, nz, rid, seed, event, odx, ody, odz, oe, oref, oblk, nullptr);
}

int hc_source(const trc_source_desc *src, long n, uint64_t seed, uint64_t offset, double *x, double *y, double *z, double *dx,
              double *dy, double *dz) {
    for (long i = 0; i < n; ++i)
        trc_source_ray(src, src->buie, nullptr, seed, offset + (uint64_t)i, &x[i], &y[i], &z[i], &dx[i], &dy[i], &dz[i]);
    return 0;
}

// the aureole's restricted power function against the library's
int hc_pow_pos(long n, const double *x, const double *y, double *out) {
    for (long i = 0; i < n; ++i) out[i] = trc_pow_pos(x[i], y[i]);
    return 0;
}

// the LDS form of the Buie inversion (streaming generation kernel) and the plain one, same table, same uniforms
int hc_buie_theta(const double *tab, long n, const double *Rv, double *plain, double *staged) {
    static trc_buie_fast F;
    trc_buie_fast_fill(tab, &F, 0, 0, 1);
    trc_buie_fast_fill(tab, &F, 1, 0, 1);
    for (long i = 0; i < n; ++i) { plain[i] = trc_buie_theta(tab, nullptr, Rv[i]); staged[i] = trc_buie_theta_fast(&F, Rv[i]); }
    return 0;
}

// nearest hit over a whole scene, brute force and Kd-tree
int hc_nearest(int n_surf, const trc_surface_desc *surfs, const double *extra, const trc_kdtree_desc *kd, long n, const double *x,
               const double *y, const double *z, const double *dx, const double *dy, const double *dz, double *t_brute,
               int *s_brute, double *t_kd, int *s_kd) {
    int max_np = 0;
    for (int i = 0; i < n_surf; ++i) { int np = trc_gm_nparams(surfs[i].gm_kind); if (np > max_np) max_np = np; }
    int stride = TRC_REC_HDR + max_np;
    if ((stride & 1) == 0) stride += 1;
    std::vector<double> recs((size_t)n_surf * stride);
    for (int i = 0; i < n_surf; ++i) pack_record(surfs[i], recs.data() + (size_t)i * stride, stride);
    std::vector<int32_t> a, b;
    trc_kd_view view;
    memset(&view, 0, sizeof(view));
    if (kd) {
        a.resize(kd->n_nodes); b.resize(kd->n_nodes);
        for (int i = 0; i < kd->n_nodes; ++i) {
            if (kd->flag[i] == 3) { a[i] = (kd->leaf_off[i] << 2) | 3; b[i] = kd->leaf_cnt[i]; }
            else { a[i] = (kd->child[i] << 2) | kd->flag[i]; b[i] = 0; }
        }
        view.node_a = a.data(); view.node_b = b.data(); view.split = kd->split; view.leaf_surfs = kd->leaf_surfs;
        view.always = kd->always_relevant; view.n_always = kd->n_always;
        for (int i = 0; i < 3; ++i) { view.bmin[i] = kd->bounds[i]; view.bmax[i] = kd->bounds[3 + i]; }
    }
    for (long i = 0; i < n; ++i) {
        trc_nearest_brute(recs.data(), stride, n_surf, extra, x[i], y[i], z[i], dx[i], dy[i], dz[i], &t_brute[i], &s_brute[i]);
        if (kd) {
            HostKdStack stk;
            trc_nearest_kd(view, stk, recs.data(), stride, extra, x[i], y[i], z[i], dx[i], dy[i], dz[i], &t_kd[i], &s_kd[i]);
        }
    }
    return 0;
}

// the single-precision conservative candidate search + exact tests (the fast engine's hot path)
struct HostStack32 {
    uint32_t na[64];
    float tmax[64];
    void push(int sp, uint32_t n, float t) { na[sp] = n; tmax[sp] = t; }
    void pop(int sp, uint32_t *n, float *t) { *n = na[sp]; *t = tmax[sp]; }
};

int hc_nearest32(int n_surf, const trc_surface_desc *surfs, const double *extra, const trc_kdtree_desc *kd, long n, const double *x,
                 const double *y, const double *z, const double *dx, const double *dy, const double *dz, double *t_out, int *s_out) {
    int max_np = 0;
    for (int i = 0; i < n_surf; ++i) { int np = trc_gm_nparams(surfs[i].gm_kind); if (np > max_np) max_np = np; }
    int stride = TRC_REC_HDR + max_np;
    if ((stride & 1) == 0) stride += 1;
    std::vector<double> recs((size_t)n_surf * stride);
    for (int i = 0; i < n_surf; ++i) pack_record(surfs[i], recs.data() + (size_t)i * stride, stride);
    trc_accel_host H;
    trc_accel_build_surfaces(surfs, n_surf, H);
    if (kd && !trc_accel_build_kd(kd, H)) return -1;
    trc_accel_view A;
    memset(&A, 0, sizeof(A));
    A.sbox = H.sbox.data(); A.nodes = H.nodes.data(); A.leaf_surfs = H.leaf_surfs.data();
    A.always = kd ? kd->always_relevant : nullptr; A.n_always = kd ? kd->n_always : 0;
    A.unbounded = H.unbounded.data(); A.n_unbounded = (int)H.unbounded.size();
    A.n_surf = n_surf; A.has_kd = kd ? 1 : 0; A.delta = H.delta;
    for (int k = 0; k < 6; ++k) A.root[k] = H.root[k];
    for (int k = 0; k < 3; ++k) { A.cen[k] = H.cen[k]; A.slo[k] = H.slo[k]; A.shi[k] = H.shi[k]; }
    for (long i = 0; i < n; ++i) {
        HostStack32 stk;
        trc_nearest_accel32(A, stk, recs.data(), stride, extra, x[i], y[i], z[i], dx[i], dy[i], dz[i], kd != nullptr, &t_out[i], &s_out[i]);
    }
    return 0;
}

// the streaming engine's candidate search: uniform grid + DDA in single precision, box tests, exact tests.
// stats[0] = cells visited, [1] = box tests, [2] = exact tests (sums over the rays).  Returns -2 when no grid can be built.
int hc_nearest_grid(int n_surf, const trc_surface_desc *surfs, const double *extra, long n, const double *x, const double *y,
                    const double *z, const double *dx, const double *dy, const double *dz, double *t_out, int *s_out, double *stats) {
    int max_np = 0;
    for (int i = 0; i < n_surf; ++i) { int np = trc_gm_nparams(surfs[i].gm_kind); if (np > max_np) max_np = np; }
    int stride = TRC_REC_HDR + max_np;
    if ((stride & 1) == 0) stride += 1;
    std::vector<double> recs_v((size_t)n_surf * stride);
    for (int i = 0; i < n_surf; ++i) pack_record(surfs[i], recs_v.data() + (size_t)i * stride, stride);
    const double *recs = recs_v.data();
    trc_accel_host H;
    trc_accel_build_surfaces(surfs, n_surf, H);
    trc_accel_build_grid(H, n_surf);
    if (!H.grid_ok) return -2;
    trc_grid_view G;
    G.off = H.grid_off.data(); G.list = H.grid_list.data();
    G.nx = H.grid_dim[0]; G.ny = H.grid_dim[1]; G.nz = H.grid_dim[2];
    G.lox = H.grid_lo[0]; G.loy = H.grid_lo[1]; G.loz = H.grid_lo[2];
    G.csx = H.grid_cs[0]; G.csy = H.grid_cs[1]; G.csz = H.grid_cs[2];
    G.ivx = H.grid_inv[0]; G.ivy = H.grid_inv[1]; G.ivz = H.grid_inv[2];
    stats[0] = stats[1] = stats[2] = 0.0;
    stats[3] = (double)((long)G.nx * G.ny * G.nz); stats[4] = (double)H.grid_list.size();
    for (long i = 0; i < n; ++i) {
        const double vx = x[i], vy = y[i], vz = z[i];
        const double ddx = dx[i], ddy = dy[i], ddz = dz[i];
        double tb = TRC_INF;
        int sb = -1;
        {
            const double dx = ddx, dy = ddy, dz = ddz;
            for (size_t k = 0; k < H.unbounded.size(); ++k) TRC_TEST_EXACT(H.unbounded[k]);
            trc_ray32 r;
            double t0;
            float tmin, tmax;
            const bool in = trc_ray32_prepare(H.slo, H.shi, H.cen, vx, vy, vz, dx, dy, dz, &r, &t0);
            if (in)
                for (size_t k = 0; k < H.grid_apart.size(); ++k) {       // surfaces set apart from the grid
                    int sidx = H.grid_apart[k];
                    stats[1] += 1.0;
                    if (trc_box_hit32(H.sbox.data() + 6 * (size_t)sidx, r)) { stats[2] += 1.0; TRC_TEST_EXACT(sidx); }
                }
            if (in && trc_kd32_root(H.grid_root, r, &tmin, &tmax)) {
                trc_dda s;
                trc_dda_start(G, r, tmin, &s);
                do {
                    stats[0] += 1.0;
                    int c = trc_dda_cell(G, s);
                    for (int k = G.off[c]; k < G.off[c + 1]; ++k) {
                        int sidx = G.list[k];
                        stats[1] += 1.0;
                        if (trc_box_hit32(H.sbox.data() + 6 * (size_t)sidx, r)) { stats[2] += 1.0; TRC_TEST_EXACT(sidx); }
                    }
                } while (trc_dda_next(G, r, tmax, &s));
            }
        }
        t_out[i] = tb;
        s_out[i] = sb;
    }
    return 0;
}

// The footprint map of a source (trc_footprint.h) against brute force: rays of the source, generated in float64 like the
// kernels generate them; every ray that hits a surface and does not take the general path must have its mask bit set, the
// surface in the list of its cell, and pass the oriented-box test from its advanced origin.
// out[0] rays, [1] generic, [2] mask bit set, [3] hits, [4] VIOLATIONS, [5] listed candidates (sum over rays with the bit set),
// [6] candidates passing the oriented box, [7] coverage, [8] largest |float32 - float64| start point, [9] eps.
// Returns 0, or -3 when the map does not apply (reason in `why`).
int hc_footprint(int n_surf, const trc_surface_desc *surfs, const double *extra, const trc_source_desc *src, long n, uint64_t seed,
                 uint64_t offset, int M, double *out, char *why, int why_len) {
    int max_np = 0;
    for (int i = 0; i < n_surf; ++i) { int np = trc_gm_nparams(surfs[i].gm_kind); if (np > max_np) max_np = np; }
    int stride = TRC_REC_HDR + max_np;
    if ((stride & 1) == 0) stride += 1;
    std::vector<double> recs((size_t)n_surf * stride);
    for (int i = 0; i < n_surf; ++i) pack_record(surfs[i], recs.data() + (size_t)i * stride, stride);
    trc_accel_host H;
    trc_accel_build_surfaces(surfs, n_surf, H);
    trc_fp_host F;
    trc_fp_build(surfs, n_surf, H, *src, F, M);
    for (int k = 0; k < 10; ++k) out[k] = 0.0;
    if (!F.ok) { if (why && why_len > 0) { strncpy(why, F.why, (size_t)why_len - 1); why[why_len - 1] = 0; } return -3; }
    const trc_fp_params &P = F.P;
    out[7] = F.coverage;
    out[9] = TRC_FP_EPS_REL * (double)P.half;
    const double *rp = src->rot_pos;
    for (long i = 0; i < n; ++i) {
        const uint64_t rid = offset + (uint64_t)i;
        double px, py, pz, dx, dy, dz;
        trc_source_ray(src, src->buie, nullptr, seed, rid, &px, &py, &pz, &dx, &dy, &dz);
        double tb; int sb;
        trc_nearest_brute(recs.data(), stride, n_surf, extra, px, py, pz, dx, dy, dz, &tb, &sb);
        uint32_t o[4];
        trc_philox4x32_10((uint32_t)rid, (uint32_t)(rid >> 32), 0, 0, (uint32_t)seed, (uint32_t)(seed >> 32), o);
        float lx, ly;
        trc_fp_position32(P, o, &lx, &ly);
        // the float64 start point in the source's local coordinates
        const double vx = px - src->center[0], vy = py - src->center[1], vz = pz - src->center[2];
        const double ex = rp[0] * vx + rp[3] * vy + rp[6] * vz, ey = rp[1] * vx + rp[4] * vy + rp[7] * vz;
        out[8] = std::fmax(out[8], std::fmax(std::fabs(ex - (double)lx), std::fabs(ey - (double)ly)));
        int32_t ix, iy;
        trc_fp_cell(P, lx, ly, &ix, &iy);
        const bool bit = (F.mask[((size_t)iy * P.M + ix) >> 5] >> (ix & 31)) & 1u;
        const bool generic = trc_fp_generic(P, o);
        out[0] += 1.0;
        if (generic) out[1] += 1.0;
        if (bit) out[2] += 1.0;
        if (sb >= 0) out[3] += 1.0;
        const size_t c = (size_t)(iy >> TRC_FP_SHIFT) * P.Mc + (ix >> TRC_FP_SHIFT);
        const float ox = (float)(px + P.t_adv * dx - H.cen[0]), oy = (float)(py + P.t_adv * dy - H.cen[1]), oz = (float)(pz + P.t_adv * dz - H.cen[2]);
        bool listed = false, boxed = false;
        for (uint32_t k = F.coff[c]; k < F.coff[c + 1]; ++k) {
            const int s = F.clist[k];
            const bool hit = trc_obb_hit32(H.obb.data() + (size_t)TRC_OBB_STRIDE * s, ox, oy, oz, (float)dx, (float)dy, (float)dz);
            if (bit && !generic) { out[5] += 1.0; if (hit) out[6] += 1.0; }
            if (s == sb) { listed = true; boxed = hit; }
        }
        if (sb >= 0 && !generic && !(bit && listed && boxed)) out[4] += 1.0;
    }
    return 0;
}

// the large grid of scenes beyond LDS (a mesh): the lists of trc_accel_build_grid32 -- a triangular face only in the cells it touches --
// with their 48-byte entries, the occupancy bits, trc_tri_hit32 in front of the exact test, the walk that ends behind the best
// hit: what trc_nearest_grid32 / k_s_bounce<2> / k_s_bounce_coop do on the device, one ray after the other.
// stats[0] = cells, [1] = listed faces looked at, [2] = exact tests, [3] = cells of the grid, [4] = list entries.  -2: no grid.
int hc_nearest_grid32(int n_surf, const trc_surface_desc *surfs, const double *extra, long n, const double *x, const double *y,
                      const double *z, const double *dx, const double *dy, const double *dz, double *t_out, int *s_out, double *stats) {
    int max_np = 0;
    for (int i = 0; i < n_surf; ++i) { int np = trc_gm_nparams(surfs[i].gm_kind); if (np > max_np) max_np = np; }
    int stride = TRC_REC_HDR + max_np;
    if ((stride & 1) == 0) stride += 1;
    std::vector<double> recs_v((size_t)n_surf * stride);
    for (int i = 0; i < n_surf; ++i) pack_record(surfs[i], recs_v.data() + (size_t)i * stride, stride);
    const double *recs = recs_v.data();
    trc_accel_host H;
    trc_accel_build_surfaces(surfs, n_surf, H);
    trc_accel_build_grid32(surfs, n_surf, H);
    if (!H.big_ok) return -2;
    trc_grid_view32 G;
    G.off = H.big_off.data(); G.list = nullptr;
    G.nx = H.big_dim[0]; G.ny = H.big_dim[1]; G.nz = H.big_dim[2];
    G.lox = H.big_lo[0]; G.loy = H.big_lo[1]; G.loz = H.big_lo[2];
    G.csx = H.big_cs[0]; G.csy = H.big_cs[1]; G.csz = H.big_cs[2];
    G.ivx = H.big_inv[0]; G.ivy = H.big_inv[1]; G.ivz = H.big_inv[2];
    stats[0] = stats[1] = stats[2] = 0.0;
    stats[3] = (double)((long)G.nx * G.ny * G.nz); stats[4] = (double)H.big_list.size();
    for (long i = 0; i < n; ++i) {
        const double vx = x[i], vy = y[i], vz = z[i];
        const double ddx = dx[i], ddy = dy[i], ddz = dz[i];
        double tb = TRC_INF;
        int sb = -1;
        {
            const double dx = ddx, dy = ddy, dz = ddz;
            for (size_t k = 0; k < H.unbounded.size(); ++k) TRC_TEST_EXACT(H.unbounded[k]);
            trc_ray32 r;
            double t0;
            float tmin, tmax;
            const bool in = trc_ray32_prepare(H.slo, H.shi, H.cen, vx, vy, vz, dx, dy, dz, &r, &t0);
            if (in)
                for (size_t k = 0; k < H.big_apart.size(); ++k) {
                    const int sidx = H.big_apart[k];
                    if (trc_box_hit32(H.sbox.data() + 6 * (size_t)sidx, r) &&
                        trc_obb_hit32(H.obb.data() + (size_t)TRC_OBB_STRIDE * sidx, r.ox, r.oy, r.oz, r.dx, r.dy, r.dz)) { stats[2] += 1.0; TRC_TEST_EXACT(sidx); }
                }
            if (in && trc_kd32_root(H.big_root, r, &tmin, &tmax)) {
                trc_dda s;
                trc_dda_start(G, r, tmin, &s);
                float t_enter = tmin;
                bool walk = true;
                while (walk) {
                    if (sb >= 0) {
                        const float tbr = (float)(tb - t0);
                        if (tbr < t_enter - (1e-3f + 1e-4f * std::fabs(tbr))) break;
                    }
                    stats[0] += 1.0;
                    const int c = trc_dda_cell(G, s);
                    if ((H.big_occ[c >> 5] >> (c & 31)) & 1u)
                        for (uint32_t k = G.off[c]; k < G.off[c + 1]; ++k) {
                            const float *e = &H.big_ent[(size_t)TRC_BG_ENT * k];
                            uint32_t w;
                            std::memcpy(&w, e, 4);
                            const int sidx = (int)(w & 0x7FFFFFFFu);
                            stats[1] += 1.0;
                            bool pass;
                            if (!(w & 0x80000000u)) pass = trc_tri_hit32(e[1], e[2], e[3], e[4], e[5], e[6], e[7], e[8], e[9], e[10], e[11], H.delta, r);
                            else pass = trc_box_hit32(e + 1, r) && trc_obb_hit32(H.obb.data() + (size_t)TRC_OBB_STRIDE * sidx, r.ox, r.oy, r.oz, r.dx, r.dy, r.dz);
                            if (pass) { stats[2] += 1.0; TRC_TEST_EXACT(sidx); }
                        }
                    t_enter = std::fmin(s.tnx, std::fmin(s.tny, s.tnz));
                    walk = trc_dda_next(G, r, tmax, &s);
                }
            }
        }
        t_out[i] = tb;
        s_out[i] = sb;
    }
    return 0;
}

// the oriented-box test against the exact test on arbitrary rays (the walk kernel's use): a ray that hits surface s exactly must
// pass the box of s from an origin advanced to the scene box.  Returns the number of violations; *passed = rays passing the box.
long hc_obb(const trc_surface_desc *s, const double *extra, long n, const double *x, const double *y, const double *z, const double *dx,
            const double *dy, const double *dz, long *passed) {
    double rec[TRC_REC_HDR + 16];
    pack_record(*s, rec, TRC_REC_HDR + 16);
    trc_accel_host H;
    trc_accel_build_surfaces(s, 1, H);
    long bad = 0;
    *passed = 0;
    if (!H.unbounded.empty()) { *passed = -1; return 0; }      // no box: such surfaces are tested exactly for every ray
    for (long i = 0; i < n; ++i) {
        const double t = trc_intersect(rec, extra, x[i], y[i], z[i], dx[i], dy[i], dz[i]);
        trc_ray32 r;
        double t0;
        const bool in = trc_ray32_prepare(H.slo, H.shi, H.cen, x[i], y[i], z[i], dx[i], dy[i], dz[i], &r, &t0);
        const bool box = in && trc_obb_hit32(H.obb.data(), r.ox, r.oy, r.oz, r.dx, r.dy, r.dz);
        if (box) *passed += 1;
        if (t > 0.0 && t < TRC_INF && !box) ++bad;
    }
    return bad;
}

// Henyey-Greenstein polar angle of the scattering optics (trc_hg_theta) for the reference's recorded draws
int hc_hg_theta(double g, long n, const double *Rv, double *theta) {
    for (long i = 0; i < n; ++i) theta[i] = trc_hg_theta(g, Rv[i]);
    return 0;
}

}  // extern "C"
