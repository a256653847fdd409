// trc_device.h -- device-side types and helpers shared by the translation units of the library (trc_kernels.hip, trc_shade.hip):
// the device view of a scene, the parameter blocks of the fast engine and of its streaming form, the chunked list appends and the
// per-hit bookkeeping.  Moved here unchanged from trc_kernels.hip / trc_stream.inc when the class-split shading kernels got a
// translation unit of their own (round 3).
#ifndef TRC_DEVICE_H
#define TRC_DEVICE_H

#include <hip/hip_runtime.h>
#include "trc_core.h"
#include "trc_bounds.h"
#include "trc_footprint.h"

struct FluxMapDev {
    int32_t surf, nu, nv, pad;
    int64_t edges_u, edges_v;  // offsets into fm_edges
    int64_t bins;              // offset into the tally buffer
    double proj[12];           // global -> local (rows 0..2 of round(inv(frame), 9), surface.py:125)
};

// device view of a scene, passed by value to kernels
struct DScene {
    const double *recs;
    const double *opt;       // n_surf * 8
    const int32_t *sflags;   // n_surf
    const double *extra;
    int32_t stride, n_surf, n_extra, has_kd;
    // Kd-tree
    const int32_t *kd_a, *kd_b, *kd_leaf, *kd_always;
    const double *kd_split;
    int32_t kd_nodes, kd_nleaf, kd_nalways, kd_pad;
    double kd_bmin[3], kd_bmax[3];
    // single-precision acceleration data (trc_bounds.h)
    const float *a_sbox;
    const float *a_obb;        // TRC_OBB_STRIDE floats per surface: oriented boxes (trc_obb_hit32)
    const uint32_t *a_nodes;
    const uint16_t *a_leaf;
    const int32_t *a_unbounded;
    const uint16_t *a_bleaf;
    int32_t a_n_unbounded, a_kd_depth, a_ok, a_kd_ok;
    int32_t a_n_bleaf, a_pad2;
    uint32_t a_bnodes[2];
    float a_broot[6];
    float a_root[6];
    float a_delta, a_pad;
    double a_cen[3], a_slo[3], a_shi[3];
    // uniform grid (streaming engine)
    const uint16_t *a_goff, *a_glist;
    const int32_t *a_gapart;    // bounded surfaces kept out of the grid (box-tested for every ray)
    int32_t a_g_ok, a_g_ncell, a_g_nlist, a_g_napart, a_gdim[3];
    float a_glo[3], a_gcs[3], a_ginv[3], a_groot[6];
    // the grid of scenes too large for LDS (trc_accel_build_grid32): 32-bit offsets and lists in global memory
    const uint32_t *a_bg_off, *a_bg_occ;    // first entry of every cell; one bit per cell that lists anything
    const float *a_bg_ent;                  // TRC_BG_ENT floats per listed surface (trc_bounds.h)
    const int32_t *a_bg_apart;  // bounded surfaces kept out of that grid (box-tested for every ray)
    int32_t a_bg_napart, a_bg_pad;
    int32_t a_bg_ok, a_bg_dim[3];
    float a_bg_lo[3], a_bg_cs[3], a_bg_inv[3], a_bg_root[6];
    // tallies: [absorbed S | received S | count S | segments, hits | flux bins ... | transfer (S+1) x S]
    double *tally;
    long long tr_off;           // offset of the surface-to-surface transfer matrix in `tally`, -1 when it is not kept
    // flux maps
    int32_t n_fm, n_fm_edges;   // flux maps and the total length of their edge arrays
    const int32_t *fm_of_surf;  // n_surf, -1 = none
    const FluxMapDev *fms;
    const double *fm_edges;
    // hit capture
    unsigned long long *counters;  // [0] hit cursor, [1] hits dropped, [2] last cursor, [3] rays left
    double *energy_left;
    long long hit_cap;
    int32_t *h_surf;
    double *h_eabs, *h_ein, *h_px, *h_py, *h_pz, *h_dx, *h_dy, *h_dz;
};


// ================================================================================================
// device helpers
// ================================================================================================
struct LocalKdStack {
    int node[TRC_KD_STACK];
    float tmax[TRC_KD_STACK];
    __device__ __forceinline__ void push(int sp, int n, double t) {
        node[sp] = n;
        tmax[sp] = __double2float_ru(t);  // rounded up: the interval only ever grows (conservative)
    }
    __device__ __forceinline__ void pop(int sp, int *n, double *t) {
        *n = node[sp];
        *t = (double)tmax[sp];
    }
};

__device__ __forceinline__ trc_kd_view make_kd_view(const DScene &sc, const int32_t *a, const int32_t *b,
                                                    const double *split, const int32_t *leaf,
                                                    const int32_t *always) {
    trc_kd_view kd;
    kd.node_a = a; kd.node_b = b; kd.split = split; kd.leaf_surfs = leaf; kd.always = always;
    kd.n_always = sc.kd_nalways;
#pragma unroll
    for (int i = 0; i < 3; ++i) { kd.bmin[i] = sc.kd_bmin[i]; kd.bmax[i] = sc.kd_bmax[i]; }
    return kd;
}

__device__ __forceinline__ unsigned lane_id() { return __lane_id(); }

// wave-level sum of a double (64 lanes)
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Queue append without a hot atomic: a wave reserves CHUNK entries at a time from the global counter (one atomic per
// chunk; a single word sustains only ~88 returning atomics per microsecond) and fills them; what is left of a chunk
// when the wave moves on is marked invalid and skipped by the consumers.  base/used are wave-uniform.
#ifndef SQ_CHUNK
#define SQ_CHUNK 256
#endif
#define SQ_INVALID 0xFFFFFFFFu

#define SQ_CHUNK_MAX 1024        /* large launches reserve more per atomic; queue slack is sized for this */

struct WaveChunk {
    unsigned long long base;
    unsigned used;
    unsigned open;
    unsigned size;      // entries reserved per atomic (SQ_CHUNK, or up to SQ_CHUNK_MAX in the largest launches)
};

__device__ __forceinline__ WaveChunk chunk_init(unsigned size = SQ_CHUNK) {
    WaveChunk c;
    c.base = 0; c.used = size; c.open = 0; c.size = size;
    return c;
}

// The first chunk of a wave can be handed out without an atomic: chunk number `wave` of a queue whose counter the host
// starts at (number of waves) * size.  Every wave of a launch reserving its first chunk at the same moment is otherwise
// thousands of atomics on one word before any work starts (~88 per microsecond).  A wave that never appends leaves the
// whole chunk marked invalid (chunk_close).
__device__ __forceinline__ WaveChunk chunk_init_static(unsigned size, unsigned long long wave) {
    WaveChunk c;
    c.base = wave * size; c.used = 0; c.open = 1; c.size = size;
    return c;
}

// the same with the chunk's first entry given (chunks of different sizes pre-assigned in one list)
__device__ __forceinline__ WaveChunk chunk_init_static_at(unsigned size, unsigned long long base) {
    WaveChunk c;
    c.base = base; c.used = 0; c.open = 1; c.size = size;
    return c;
}

// after a chunk_append made by a subset of the lanes: every lane takes the state of `lane` (one that took part)
__device__ __forceinline__ void chunk_rebroadcast(WaveChunk &c, int lane) {
    c.base = __shfl(c.base, lane, 64);
    c.used = (unsigned)__shfl((int)c.used, lane, 64);
    c.open = (unsigned)__shfl((int)c.open, lane, 64);
    c.size = (unsigned)__shfl((int)c.size, lane, 64);
}

// marks the unused tail of the current chunk invalid (wave-uniform call)
__device__ __forceinline__ void chunk_close(WaveChunk &c, uint32_t *tag, long long cap) {
    if (c.open) {
        for (unsigned k = c.used + lane_id(); k < c.size; k += 64)
            if ((long long)(c.base + k) < cap) tag[c.base + k] = SQ_INVALID;
    }
    c.open = 0;
    c.used = c.size;
}

// returns this lane's index in the queue (meaningful when `want`).  Call it with the whole wave, or -- inside a
// divergent region -- re-broadcast the chunk afterwards from a lane that took part (chunk_rebroadcast).  A request that does not fit the
// open chunk fills it up and continues in a new one, so entries are only wasted at the end of a kernel (< CHUNK per wave).
__device__ __forceinline__ unsigned long long chunk_append(unsigned long long *counter, WaveChunk &c, bool want, uint32_t *tag,
                                                           long long cap) {
    unsigned long long m = __ballot(want);
    if (!m) return 0;
    const unsigned need = (unsigned)__popcll(m);
    const unsigned rank = (unsigned)__popcll(m & ((1ull << lane_id()) - 1ull));
    unsigned long long idx;
    if (c.used + need > c.size) {
        const unsigned rem = c.open ? c.size - c.used : 0u;
        const unsigned long long old_pos = c.base + c.used;
        const int leader = __ffsll((long long)m) - 1;          // a lane that is certainly active here
        unsigned long long b = 0;
        if ((int)lane_id() == leader) b = atomicAdd(counter, (unsigned long long)c.size);
        c.base = __shfl(b, leader, 64);
        c.used = need - rem;
        c.open = 1;
        idx = rank < rem ? old_pos + rank : c.base + (rank - rem);
    } else {
        idx = c.base + c.used + rank;
        c.used += need;
    }
    (void)tag; (void)cap;
    return idx;
}

#define SHADE_MAX_WAVES 5120     /* waves of one launch of a shading kernel (at most n_cu * 2 workgroups of 10 waves, n_cu <= 256) */
#define SQ_HIT_CHUNK 1024        /* entries of the hit buffer a wave of k_s_shade reserves per atomic: the cursor is one word (~88 returning
                                    atomics per microsecond), and at 256 the 12 000 reservations of an NSTTF batch were half of the kernel */
#define TRC_HIT_HOLDERS (4ll * SHADE_MAX_WAVES)   /* waves that can hold an open chunk of the hit buffer: SHADE_MAX_WAVES per slot (the shading
                                                    kernels of a slot run one after another, wave w of each continues the chunk wave w of the
                                                    one before left open), four slots at most */
#define TRC_SURF_TERMINAL 0x10000   /* device copy of the surface flags only: every ray that lands here ends here -- the optics absorb all of
                                       it whatever the angle (absorptivity 1, no incidence-angle factor): no direction needs to be drawn */

// per-hit bookkeeping shared by both engines: tallies, flux map, hit capture
template <bool LDS_TALLY>
__device__ __forceinline__ void record_hit(const DScene &sc, double *lds_tally, int s, double e_in,
                                           double e_abs, double hx, double hy, double hz, double dx,
                                           double dy, double dz, bool capture_enabled, int prev, WaveChunk *hc = nullptr,
                                           double *lds_fm = nullptr, bool volume = false, bool tallied = false,
                                           unsigned long long *slot_out = nullptr) {
    // slot_out: where the hit went in the hit buffer (~0: not captured) -- chunked appends only
    // tallied: the caller has added the three per-surface sums itself (k_s_absorb: once per wave)
    // volume: the ray was scattered in the medium before it reached the surface -- nothing is recorded, but the lane takes part
    // in the appends of the wave below (their bookkeeping is per wave)
    const int S = sc.n_surf;
    // energy carried from the surface the ray left (S = the source) to the one it lands on
    if (sc.tr_off >= 0 && !volume) atomicAdd(&sc.tally[sc.tr_off + (long long)prev * S + s], e_in);
    if (volume || tallied) {
    } else if (LDS_TALLY) {
        atomicAdd(&lds_tally[s], e_abs);
        atomicAdd(&lds_tally[S + s], e_in);
        atomicAdd(&lds_tally[2 * S + s], 1.0);
    } else {
        atomicAdd(&sc.tally[s], e_abs);
        atomicAdd(&sc.tally[S + s], e_in);
        atomicAdd(&sc.tally[2 * S + s], 1.0);
    }
    int fm = (!volume && sc.fm_of_surf) ? sc.fm_of_surf[s] : -1;
    if (fm >= 0) {
        const FluxMapDev &m = sc.fms[fm];
        double u = m.proj[0] * hx + m.proj[1] * hy + m.proj[2] * hz + m.proj[3];
        double v = m.proj[4] * hx + m.proj[5] * hy + m.proj[6] * hz + m.proj[7];
        int iu = trc_bin_index(sc.fm_edges + m.edges_u, m.nu, u);
        int iv = trc_bin_index(sc.fm_edges + m.edges_v, m.nv, v);
        if (iu >= 0 && iv >= 0) {
            // lds_fm: the workgroup's private copy of all flux-map bins (they follow the 3S+2 per-surface sums in the tally buffer);
            // scattered global float64 atomics run at ~1/17 of the rate of the contiguous ones the copy is flushed with
            if (lds_fm) atomicAdd(&lds_fm[m.bins - (3 * (int64_t)S + 2) + (int64_t)iu * m.nv + iv], e_abs);
            else atomicAdd(&sc.tally[m.bins + (int64_t)iu * m.nv + iv], e_abs);
        }
    }
    if (capture_enabled && hc) {
        // chunked append (streaming engine): one atomic per 256 captured hits instead of one per wave and iteration --
        // the cursor of the hit buffer is a single word, and a word sustains only ~88 returning atomics per microsecond
        const int fl = sc.sflags[s];
        bool want = !volume && (fl & TRC_SURF_CAPTURE_HITS) != 0;
        unsigned long long slot = chunk_append(&sc.counters[0], *hc, want, nullptr, 0);
        if (slot_out) *slot_out = (want && (long long)slot < sc.hit_cap) ? slot : ~0ull;
        if (want) {
            if ((long long)slot < sc.hit_cap) {
                sc.h_surf[slot] = s;
                sc.h_eabs[slot] = e_abs;
                sc.h_px[slot] = hx; sc.h_py[slot] = hy; sc.h_pz[slot] = hz;
                if (!(fl & TRC_SURF_CAPTURE_LEAN)) { sc.h_ein[slot] = e_in; sc.h_dx[slot] = dx; sc.h_dy[slot] = dy; sc.h_dz[slot] = dz; }
            } else {
                atomicAdd(&sc.counters[1], 1ull);
            }
        }
    } else if (capture_enabled) {
        // wave-aggregated append: one atomic per wave per iteration
        const int fl = sc.sflags[s];
        bool want = !volume && (fl & TRC_SURF_CAPTURE_HITS) != 0;
        unsigned long long mask = __ballot(want);
        if (mask) {
            int leader = __ffsll((long long)mask) - 1;
            unsigned long long base = 0;
            if ((int)lane_id() == leader) base = atomicAdd(&sc.counters[0], (unsigned long long)__popcll(mask));
            base = __shfl(base, leader, 64);
            if (want) {
                unsigned long long slot = base + __popcll(mask & ((1ull << lane_id()) - 1ull));
                if ((long long)slot < sc.hit_cap) {
                    sc.h_surf[slot] = s;
                    sc.h_eabs[slot] = e_abs;
                    sc.h_px[slot] = hx; sc.h_py[slot] = hy; sc.h_pz[slot] = hz;
                    if (!(fl & TRC_SURF_CAPTURE_LEAN)) { sc.h_ein[slot] = e_in; sc.h_dx[slot] = dx; sc.h_dy[slot] = dy; sc.h_dz[slot] = dz; }
                } else {
                    atomicAdd(&sc.counters[1], 1ull);
                }
            }
        }
    }
}

// ================================================================================================
// fast engine
// ================================================================================================
// what the rays of a given bundle carry beyond the fast engine's record (streaming form only, k_s_shade_x): Im of the refractive
// index (n), the materials at each ray's wavelength (2 n_mat x n), sample wavelengths and spectra of polychromatic bundles
// (n_spec x n).  Kept out of FastParams and at the end of StreamParams: the kernels of the other scenes never read it, and six
// more words in the middle of their arguments cost the NSTTF bench 4 % (scalar loads regrouped).
struct CarryIn {
    const double *ref_im, *mat, *spec_wl, *spec;
    int n_mat, n_spec;
    double *slot_spec;      // the slot's table of spectra under way: sample w of the ray in slot s at slot_spec[w * room + s] (null: none)
    double *hit_x;          // 3 n_spec columns beside the hit buffer, hit_x_cap entries each (null: no spectra per captured hit):
    long long hit_x_cap;    // sample wavelengths, spectrum that arrived, spectrum that left
};

struct FastParams {
    DScene sc;
    // given bundle (NULL when a source descriptor is used)
    const double *x, *y, *z, *dx, *dy, *dz, *e, *ref, *wl;
    const uint64_t *rid;
    const trc_source_desc *src;  // device copy
    long long n;
    int reps;
    int flags;
    double min_energy;
    unsigned long long seed, ray_offset;
    // rays left after `reps` bounces
    double *lx, *ly, *lz, *ldx, *ldy, *ldz, *le;
    long long last_cap;
    // LDS carve-up (in doubles / flags)
    int lds_scene;    // surfaces (+ Kd arrays) staged in LDS
    int lds_tally;    // tallies privatised in LDS
    int capture;      // some surface captures hits
};


// optics kinds of the "mirrors and diffuse walls" family: what a heliostat field, a dish or a cavity of opaque walls is made of.
// A scene of flat surfaces with only these gets an instance of k_s_shade that carries nothing else (SIMPLE below).
#define TRC_OPT_SIMPLE_MASK ((1u << TRC_OPT_TRANSPARENT) | (1u << TRC_OPT_REFLECTIVE) | (1u << TRC_OPT_ONE_SIDED_REFLECTIVE) | \
                             (1u << TRC_OPT_REAL_REFLECTIVE) | (1u << TRC_OPT_ONE_SIDED_REAL_REFLECTIVE) | (1u << TRC_OPT_LAMBERTIAN) | \
                             (1u << TRC_OPT_LAMBERTIAN_SPECULAR))

// shading + bookkeeping of one hit, shared by the two fast kernels.  Returns false when the ray stops here.
// SIMPLE promises a flat geometry kind and an optics kind of TRC_OPT_SIMPLE_MASK on every surface: the compiler drops the rest.
template <bool LDS_TALLY, bool SIMPLE = false>
__device__ __forceinline__ bool fast_shade(const FastParams &P, const double *recs, double *l_tally, double t, int s, double &px,
                                           double &py, double &pz, double &dx, double &dy, double &dz, double &e, double &ref,
                                           double wl, unsigned long long rid, int &bounce, int &prev, WaveChunk *hc = nullptr,
                                           double *lds_fm = nullptr, bool tallied = false, bool *was_volume = nullptr) {
    // tallied: the caller adds the three per-surface sums itself (k_s_shade: per wave), unless *was_volume comes back true
    const DScene &sc = P.sc;
    bounce += 1;
    const double *rec = recs + (size_t)s * sc.stride;
    double hx = px + t * dx, hy = py + t * dy, hz = pz + t * dz;
    double nx, ny, nz;
    if (SIMPLE && !trc_gm_is_flat(trc_rec_gm_kind(rec))) __builtin_unreachable();
    trc_normal(rec, hx, hy, hz, dx, dy, dz, &nx, &ny, &nz);
    trc_ray_out out[2];
    const double path = sqrt((hx - px) * (hx - px) + (hy - py) * (hy - py) + (hz - pz) * (hz - pz));
    if (SIMPLE && !((TRC_OPT_SIMPLE_MASK >> trc_rec_opt_kind(rec)) & 1u)) __builtin_unreachable();
    int n_out = trc_shade(trc_rec_opt_kind(rec), sc.opt + (size_t)s * 8, sc.extra, trc_rec_extra_off(rec), trc_rec_extra_len(rec),
                          rec[2], rec[5], rec[8], dx, dy, dz, e, ref, wl, path, nx, ny, nz, P.seed, rid, (uint32_t)bounce, out);
    (void)n_out;  // scenes whose optics split rays are routed to the ordered engine by the host
    // a volume event (scattering in the medium): the ray never reached the surface -- it goes on from a point before the hit,
    // the surface records nothing, the surface the ray left stays the one it left
    const bool volume = out[0].back > 0.0;
    if (was_volume) *was_volume = volume;
    if (volume) { hx -= out[0].back * dx; hy -= out[0].back * dy; hz -= out[0].back * dz; }
    double e_abs = e - out[0].e;
    record_hit<LDS_TALLY>(sc, l_tally, s, e, e_abs, hx, hy, hz, dx, dy, dz, P.capture != 0, prev, hc, lds_fm, volume, tallied);
    if (!volume) prev = s;
    px = hx + out[0].shift * nx; py = hy + out[0].shift * ny; pz = hz + out[0].shift * nz;      // (PeriodicBoundary: one period along the normal)
    dx = out[0].dx; dy = out[0].dy; dz = out[0].dz;
    e = out[0].e; ref = out[0].ref;
    if (e <= P.min_energy) return false;                      // tracer_engine.py:242
    if (bounce >= P.reps) {                                   // still alive after the last iteration
        atomicAdd(&sc.counters[3], 1ull);
        atomicAdd(sc.energy_left, e);
        if (P.flags & TRC_TRACE_KEEP_LAST) {
            unsigned long long slot = atomicAdd(&sc.counters[2], 1ull);
            if ((long long)slot < P.last_cap) {
                P.lx[slot] = px; P.ly[slot] = py; P.lz[slot] = pz;
                P.ldx[slot] = dx; P.ldy[slot] = dy; P.ldz[slot] = dz; P.le[slot] = e;
            }
        }
        return false;
    }
    return true;
}


// One ray of the table, split by use: what the search reads and updates is exactly one 64-byte sector, the rest
// (needed by shading only) another 32 bytes.  A structure-of-arrays table costs one sector per *field* on the gathers
// by slot of k_s_exact / k_s_shade (8x the useful bytes).
// Slots are handed out densely, in the order in which rays turn out to have a candidate (a chunked counter, CN(8)): of the
// 1e8 source rays of an NSTTF step 6.5e6 ever need a record, and a table indexed by ray number spread those over 6.4 GB --
// every access of the later bounces a DRAM page of its own.  The ray's number in its batch travels in the record (idx).
#define SQ_SKIP_SELF 0x80000000u    /* tail, high word: the ray cannot meet the (flat) surface it left again -- its exact test would
                                       return t < 1e-7 (flat_surface.py:50): the searches leave that surface out */
struct __attribute__((aligned(64))) SRayGeo {
    double px, py, pz, dx, dy, dz;
    uint32_t head;                  // index in Q3 of the last finite candidate linked so far, SQ_INVALID = none (general path)
    uint32_t idx;                   // the ray's number in its batch: stream id = ray_offset + base + idx
    unsigned long long tail;        // bounce (low word) and the surface the ray left (high word, | SQ_SKIP_SELF)
};
__device__ __forceinline__ unsigned long long sray_tail(int bounce, uint32_t prev) { return (unsigned long long)(uint32_t)bounce | ((unsigned long long)prev << 32); }
// result of one exact test, linked per ray
struct __attribute__((aligned(16))) SCand {
    double t;
    uint32_t surf;
    uint32_t next;                  // the ray's previous finite candidate (index in Q3) or SQ_INVALID
};
struct __attribute__((aligned(32))) SRayAux {
    double e, ref, wl;
    double pad;
};

// Workgroups of k_s_shade add into one of TALLY_PARTS copies of the tally buffer (copy = workgroup number mod TALLY_PARTS):
// the end-of-workgroup flush of ~3S sums and the flux-map bins would otherwise be thousands of atomics per 128-byte line,
// which serialise at ~88 per microsecond.
#ifndef TALLY_PARTS
#define TALLY_PARTS 16          /* (the mesh of 1e5 faces, sums added per hit in global memory: 2 copies 12.6 ms per 1e7 rays, 4: 10.1, 16: 8.8, 32: 8.6, 64: 8.8) */
#endif

// counters live 128 bytes apart: atomics on words of one cache line serialise in the same L2 bank
#define CN(k) ((k) << 4)
#define CN_WORDS (32 << 4)

struct StreamWs {
    long long cap;      // rays per batch
    long long room;     // entries of the ray table and of every list of slots / ray numbers: cap + room for the unused tails of the
                        // chunks (SQ_CHUNK_MAX * 16384 entries; TRC_STREAM_ROOM sets the whole: tests make the lists overflow with it)
    long long q3_cap;   // candidate pairs per bounce
    long long act_room; // entries of the two active lists: several shading kernels append to one list, each behind chunks sized for what
                        // it is expected to keep alive -- room for every ray of the batch behind pre-assigned chunks for every ray
    SRayGeo *geo;
    SRayAux *aux;
    uint32_t *q1_slot;
    float4 *q1_a, *q1_b;   // (ox, oy, oz, ix), (iy, iz, tmin, tmax)
    uint32_t *q3_slot, *q3_surf;
    SCand *q3n;
    uint32_t *hit_slot;        // the hit list: slot of every ray that hit something, with -- when the kernel that found the hit
    uint32_t *hit_surf;        // resolved it itself (k_s_fresh, k_s_bounce) -- the surface and the distance; SQ_INVALID in hit_surf:
    double *hit_t;             // k_s_shade picks the nearest of the ray's linked candidates (general path, k_s_exact)
    uint32_t *gen_list;        // fresh rays that k_s_cull leaves to the general path (ray numbers in the batch)
    uint32_t *fq_ray, *fq_cell;   // fresh rays that start inside a footprint: ray number, mask cell (k_s_cull -> k_s_fresh)
    uint32_t *act[2];
    double *tally_part;        // TALLY_PARTS private copies of the scene's tally buffer (merged at the end of the call)
    long long tally_n;
    unsigned long long *hit_state;   // per wave of k_s_shade: open chunk of the scene's hit buffer, kept across launches
    // the hit list of a bounce parted by optics class (k_s_partition; scenes with more than one shading class): list c holds the
    // hits on surfaces of class c, resolved (surface and distance), pl_room entries each; null for classes the scene does not have
    uint32_t *pl_slot[3], *pl_surf[3];
    double *pl_t[3];
    long long pl_room;
    unsigned long long *cnt;   // CN(k): [0] Q1, [1] Q3, [2] hit list, [3] next active list, [4] overflow flag, [5] entries of the
                               // active list coming in, [6] hits and [7] rays going on (real counts), [8] slots of the ray table
                               // handed out, [9] general-path list and [10] footprint list of k_s_cull, [11] hits on terminal surfaces
                               // (k_s_bounce -> k_s_absorb) and [12] their real count; [13..15] hits shaded per class; [16..22] SW_STATS;
                               // [24..26] entries of the class lists of k_s_partition
};

// the footprint map of the call's source on the device (trc_footprint.h)
struct FpDev {
    trc_fp_params P;
    const uint32_t *mask, *coff, *clist;     // lists: 32-bit in global memory; k_s_fresh's LDS copies are 16-bit
    long long n_list;          // entries of clist
};

struct StreamParams {
    FastParams P;
    StreamWs W;
    long long base;      // index of the batch's first ray in the call
    long long nb;        // rays in the batch
    const uint32_t *act_in;
    uint32_t *act_out;
    unsigned hit_epoch;  // generation of the scene's hit buffer: open chunks of an older one are stale
    unsigned chunk_q1, chunk_q3;   // entries reserved per atomic in the walker / candidate queues of this launch
    long long q3_gen_chunk0;   // >= 0: k_s_gen<false> takes pre-assigned first chunks of Q3 from this chunk number on (behind k_s_walk's)
    const uint32_t *gen_list;  // k_s_gen<true>: the rays to generate (numbers in the batch, CN(9) entries), null = all nb rays
    // A single counter word sustains ~88 returning atomics per microsecond, and the waves of a kernel run out of their chunks at
    // about the same time: 8192 waves fetching a second chunk are 0.1 ms.  So every list that a bounce fills gets ONE pre-assigned
    // chunk per appending wave, sized by the host for what the wave is expected to append (+ margin); the atomic is the rare path.
    unsigned chunk_hit;        // hit list: k_s_fresh / k_s_bounce
    unsigned chunk_slot;       // slots of the ray table: k_s_fresh
    unsigned chunk_act;        // next active list: k_s_shade
    unsigned chunk_first;      // slots and hit list: k_s_bounce<.., FRESH>
    long long slot_base0, hit_base0;   // first entry of the pre-assigned chunks of k_s_gen<true> (slots) and of k_s_exact (hit list):
                                       // behind those of k_s_fresh
    int static_general;        // the kernels of the general path have pre-assigned first chunks (not behind k_s_cull: few rays)
    FpDev fp;
    int static_first;    // the first chunk of every wave is pre-assigned (chunk_init_static): Q1 in k_s_gen, Q3 in k_s_walk,
                         // hit list in k_s_exact, active list in k_s_shade; the host starts the counters accordingly
    int lds_recs;        // k_s_exact / k_s_shade stage the surface records in LDS
    int lds_tables;      // k_s_shade also stages optics parameters, flux-map tables and capture flags
    int lds_fm_bins;     // k_s_shade: flux-map bins privatised in LDS (all of them, or 0)
    int lds_extra;       // k_s_shade: the scene's table values (optics tables: absorptance over angle / wavelength, complex indices) in LDS
                         // too -- an interpolation is a binary search, every step a dependent load
    int bounce_no;       // the bounce this launch belongs to (every ray of a launch is at the same bounce)
    int bg_occ_words;    // k_s_bounce<2>: the occupancy bits of the large grid staged in LDS (this many words), 0 = read from global memory
    int split_terminal;  // k_s_bounce lists the hits on surfaces that end every ray (TRC_SURF_TERMINAL) apart, for k_s_absorb: CN(11),
                         // entries in the arrays of the walker queue, which is idle in a bounce that k_s_bounce serves
    unsigned chunk_thit; // ... one pre-assigned chunk per wave of k_s_bounce
    int search;          // candidate search: 0 all boxes (one leaf), 1 packed Kd-tree, 2 uniform grid
    // Shading is split by optics class (trc_shade.hip): every shading kernel of a bounce walks the same hit list and takes the
    // hits on surfaces of its class (TRC_SURF_CLS_* bits of the device copy of the surface flags).
    int shade_cls;             // the class this launch of a shading kernel serves (k_s_shade: TRC_CLS_GENERAL)
    int shade_term_cls;        // >= 0: hits on surfaces that end every ray (TRC_SURF_TERMINAL) go to the kernel of this class whatever
                               // their own, and are finished there without optics (e_out = 0 <= min_energy); -1: no such shortcut
    long long act_base0;       // first entry of the pre-assigned chunks of the active list that belong to this launch (the shading
                               // kernels of a bounce append to one list; wave w starts at act_base0 + w * chunk_act)
    unsigned chunk_hitbuf;     // entries of the scene's hit buffer a wave reserves per atomic (<= SQ_HIT_CHUNK; small buffers: less)
    // the list a shading kernel walks: the bounce's hit list (counter CN(2)), or its class's part of it (k_s_partition, CN(24 + class))
    const uint32_t *hl_slot, *hl_surf;
    const double *hl_t;
    long long hl_room;
    int hl_cn;                 // index of the counter word that holds the list's length (CN(...) applied)
    // k_s_partition: per class, entries pre-assigned to every wave (0: the waves reserve as they go) and where those chunks start
    unsigned part_chunk[3];
    int part_static[3];
    CarryIn carry;      // (last: see CarryIn)
};

// optics classes of the shading stage.  MIRROR and DIFFUSE are served by lean kernels (trc_shade.hip: <= 128 registers, four
// waves per SIMD and more); everything else -- refraction, media, conductors, incidence-angle modifiers -- by k_s_shade.
#define TRC_CLS_MIRROR 0       /* Transparent, Reflective, OneSidedReflective, RealReflective, OneSidedRealReflective (no IAM) */
#define TRC_CLS_DIFFUSE 1      /* Lambertian (no IAM, no absorbing medium), LambertianSpecular, SemiLambertian, Reflective_spectral, the
                                  Lambertian_directional_axisymmetric_piecewise family */
#define TRC_CLS_GENERAL 2
#define TRC_CLS_COUNT 3
#define TRC_SURF_CLS_SHIFT 17  /* device copy of the surface flags only: bits 17-18 = class */
#define TRC_SURF_CLS_MASK 0x3
#define TRC_CLS_MIRROR_KINDS ((1u << TRC_OPT_TRANSPARENT) | (1u << TRC_OPT_REFLECTIVE) | (1u << TRC_OPT_ONE_SIDED_REFLECTIVE) | \
                              (1u << TRC_OPT_REAL_REFLECTIVE) | (1u << TRC_OPT_ONE_SIDED_REAL_REFLECTIVE))
#define TRC_CLS_DIFFUSE_KINDS ((1u << TRC_OPT_LAMBERTIAN) | (1u << TRC_OPT_LAMBERTIAN_SPECULAR) | (1u << TRC_OPT_SEMI_LAMBERTIAN) | \
                               (1u << TRC_OPT_REFLECTIVE_SPECTRAL) | (1u << TRC_OPT_LAMBERTIAN_DIRECTIONAL) |                     \
                               (1u << TRC_OPT_LAMBERTIAN_DIRECTIONAL_SPECTRAL) | (1u << TRC_OPT_FRESNEL_CONDUCTOR))

// class of a surface from its optics kind and parameters (host side, when the flags are uploaded)
static inline int trc_shade_class_of(const trc_surface_desc &sd) {
    const int ok = sd.optics_kind;
    if (ok < 0 || ok >= 32) return TRC_CLS_GENERAL;
    if ((TRC_CLS_MIRROR_KINDS >> ok) & 1u) {
        const bool iam = ((ok == TRC_OPT_REFLECTIVE || ok == TRC_OPT_ONE_SIDED_REFLECTIVE) && sd.opt[1] != 0.0) ||
                         ((ok == TRC_OPT_REAL_REFLECTIVE || ok == TRC_OPT_ONE_SIDED_REAL_REFLECTIVE) && sd.opt[3] != 0.0);
        return iam ? TRC_CLS_GENERAL : TRC_CLS_MIRROR;
    }
    if ((TRC_CLS_DIFFUSE_KINDS >> ok) & 1u) {
        if (ok == TRC_OPT_LAMBERTIAN && (sd.opt[2] != 0.0 || sd.opt[4] != 0.0)) return TRC_CLS_GENERAL;      // absorbing medium / IAM
        return TRC_CLS_DIFFUSE;
    }
    return TRC_CLS_GENERAL;
}

// the lean shading kernels (trc_shade.hip): kernel of a class for a scene of flat surfaces only (flat) with its tables in LDS (lds)
const void *trc_shade_carry_kernel(bool lds);      // k_s_shade_x (trc_shade.hip): every optics kind, with what the rays carry
const void *trc_shade_lean_kernel(int cls, bool flat, bool lds);
#ifndef SHC_THREADS
#define SHC_THREADS 1024
#endif

__device__ __forceinline__ trc_accel_view stream_accel_global(const DScene &sc, int mode) {
    const bool kd32 = mode == 1;
    trc_accel_view A;
    A.sbox = sc.a_sbox; A.nodes = kd32 ? sc.a_nodes : nullptr; A.leaf_surfs = kd32 ? sc.a_leaf : sc.a_bleaf;
    // surfaces tested for every ray besides the search structure: the Kd-tree's always-relevant ones, or those set apart
    // from the grid
    A.always = mode == 2 ? sc.a_gapart : sc.kd_always; A.unbounded = sc.a_unbounded;
    A.n_always = kd32 ? sc.kd_nalways : (mode == 2 ? sc.a_g_napart : 0); A.n_unbounded = sc.a_n_unbounded; A.n_surf = sc.n_surf; A.has_kd = 1;
#pragma unroll
    for (int i = 0; i < 6; ++i) A.root[i] = kd32 ? sc.a_root[i] : (mode == 2 ? sc.a_groot[i] : sc.a_broot[i]);
    A.delta = sc.a_delta;
#pragma unroll
    for (int i = 0; i < 3; ++i) { A.cen[i] = sc.a_cen[i]; A.slo[i] = sc.a_slo[i]; A.shi[i] = sc.a_shi[i]; }
    return A;
}

// Nearest hit of ONE ray through the large grid, one lane per ray (the ordered engine on a scene that stands on the large grid:
// a mesh of 1e5 faces, where testing every surface for every ray -- what the engine did -- took 370 ms per 2e5 rays).  The cells
// of the DDA with the occupancy bits, the listed faces from their 48-byte entries (trc_tri_hit32 / box and oriented box), the exact
// test at once; the walk ends at the first cell behind the nearest hit.  Same candidates, same exact test and tie rule as
// k_s_bounce<2> and trc_nearest_brute: t > 0, the lowest index among equal distances.  *s_out < 0: no hit.
__device__ __forceinline__ void trc_nearest_grid32(const DScene &sc, double px, double py, double pz, double dx, double dy, double dz,
                                                    double *t_out, int *s_out) {
    trc_accel_view A = stream_accel_global(sc, 0);
#pragma unroll
    for (int i = 0; i < 6; ++i) A.root[i] = sc.a_bg_root[i];
    trc_grid_view32 G;
    memset(&G, 0, sizeof(G));
    G.off = sc.a_bg_off; G.list = nullptr;
    G.nx = sc.a_bg_dim[0]; G.ny = sc.a_bg_dim[1]; G.nz = sc.a_bg_dim[2];
    G.lox = sc.a_bg_lo[0]; G.loy = sc.a_bg_lo[1]; G.loz = sc.a_bg_lo[2];
    G.csx = sc.a_bg_cs[0]; G.csy = sc.a_bg_cs[1]; G.csz = sc.a_bg_cs[2];
    G.ivx = sc.a_bg_inv[0]; G.ivy = sc.a_bg_inv[1]; G.ivz = sc.a_bg_inv[2];
    double tb = TRC_INF;
    int sb = 0x7FFFFFFF;
    auto exact = [&](int s) __attribute__((always_inline)) {
        const double t = trc_intersect(sc.recs + (size_t)s * sc.stride, sc.extra, px, py, pz, dx, dy, dz);
        if (t > 0.0 && t < TRC_INF && (t < tb || (t == tb && s < sb))) { tb = t; sb = s; }
    };
    for (int k = 0; k < A.n_unbounded; ++k) exact(A.unbounded[k]);
    trc_ray32 r;
    double t0 = 0.0;
    if (trc_ray32_prepare(A.slo, A.shi, A.cen, px, py, pz, dx, dy, dz, &r, &t0)) {
        for (int k = 0; k < sc.a_bg_napart; ++k) {        // surfaces set apart from the grid
            const int sidx = sc.a_bg_apart[k];
            const float *b = A.sbox + 6 * (size_t)sidx;
            if (b[3] == TRC_INF && b[0] == -TRC_INF) continue;
            if (trc_box_hit32(b, r) && trc_obb_hit32(sc.a_obb + (size_t)TRC_OBB_STRIDE * sidx, r.ox, r.oy, r.oz, r.dx, r.dy, r.dz)) exact(sidx);
        }
        float tmin = 1.0f, tmax = 0.0f;
        if (trc_kd32_root(A.root, r, &tmin, &tmax)) {
            const float4 *ent = (const float4 *)sc.a_bg_ent;
            trc_dda dd;
            trc_dda_start(G, r, tmin, &dd);
            float t_enter = tmin;
            bool walk = true;
            while (walk) {
                if (sb != 0x7FFFFFFF) {
                    const float tbr = (float)(tb - t0);
                    if (tbr < t_enter - (1e-3f + 1e-4f * fabsf(tbr))) break;      // every cell from here on starts behind the best hit
                }
                const int cell = trc_dda_cell(G, dd);
                if ((sc.a_bg_occ[cell >> 5] >> (cell & 31)) & 1u) {
                    const uint32_t k1 = G.off[cell + 1];
                    for (uint32_t k = G.off[cell]; k < k1; ++k) {
                        const float4 c0 = ent[3 * (size_t)k], c1 = ent[3 * (size_t)k + 1], c2 = ent[3 * (size_t)k + 2];
                        const uint32_t w = __float_as_uint(c0.x);
                        const int sidx = (int)(w & 0x7FFFFFFFu);
                        bool pass;
                        if (!(w & 0x80000000u)) pass = trc_tri_hit32(c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, c2.x, c2.y, c2.z, c2.w, A.delta, r);
                        else {
                            const float b[6] = {c0.y, c0.z, c0.w, c1.x, c1.y, c1.z};
                            pass = trc_box_hit32(b, r) && trc_obb_hit32(sc.a_obb + (size_t)TRC_OBB_STRIDE * sidx, r.ox, r.oy, r.oz, r.dx, r.dy, r.dz);
                        }
                        if (pass) exact(sidx);
                    }
                }
                t_enter = fminf(dd.tnx, fminf(dd.tny, dd.tnz));
                walk = trc_dda_next(G, r, tmax, &dd);
            }
        }
    }
    *t_out = sb != 0x7FFFFFFF ? tb : 0.0;
    *s_out = sb != 0x7FFFFFFF ? sb : -1;
}

// ---------------------------------------------------------------------------------------------------------------
// capacity of the ray table and of the lists of slots / ray numbers: rays per batch + room for the unused tails of the chunks
#define SQ_ROOM(W) ((W).room)

#endif
