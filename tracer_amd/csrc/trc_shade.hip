// trc_shade.hip -- the shading stage of the streaming engine, split by optics class.
//
// k_s_shade (trc_stream.inc) carries every optics kind of optics_callables.py in one kernel: 243-256 registers, two waves per SIMD,
// more than half of its wave cycles waiting.  The registers are not held by any one kind -- a mirror with slope error
// (optics_callables.py:214-269 on ray_trace_utils/vector_manipulations.py:56-74) needs 66 -- but by the union of them and by what
// was built around them to live with two waves (entries and ray records of the next two hits fetched ahead in registers).
// Here one kernel per CLASS of optics walks the bounce's hit list and takes the hits on surfaces of its class; what a class does
// not need is not compiled into its kernel (trc_shade_k<KINDS, false>), nothing is fetched ahead, and the tables the host passes
// are few: the instances fit 128 registers and less, so a CU holds sixteen waves of them and their loads hide each other.
//
//   TRC_CLS_MIRROR    Transparent, Reflective, OneSidedReflective, RealReflective, OneSidedRealReflective (:93-140, :195-269, :492-504)
//   TRC_CLS_DIFFUSE   Lambertian, LambertianSpecular, SemiLambertian, Reflective_spectral and the table-driven
//                     Lambertian_directional_axisymmetric_piecewise family (:143-193, :331-391, :427-487, :506-585), and
//                     FresnelConductorHomogenous (:1536-1558: a mirror whose reflectance comes from a tabulated complex index -- the metal
//                     ring of the cavity receiver, 4.5 % of its hits, was the only reason for a third kernel and a third list there)
//   (TRC_CLS_GENERAL  everything else stays with k_s_shade)
//
// Per hit the arithmetic is the one of the other engines: trc_normal, trc_shade_k = trc_shade with the other kinds left out.
#include "trc_device.h"

// FLATN: every surface of the scene is flat (its normal is the z axis of its frame, flat_surface.py:84-91)
// LDS:   tallies, records, optics parameters, flux-map tables, flags (and the optics tables for the DIFFUSE class) are staged in
//        LDS -- all of them or none, so that every access has a known address space (see k_s_shade)
template <int CLS, bool FLATN, bool LDS>
__global__ __launch_bounds__(SHC_THREADS) void k_s_shade_c(StreamParams S) {
    extern __shared__ double lds[];
    const FastParams &P = S.P;
    const DScene &sc = P.sc;
    const StreamWs &W = S.W;
    const int Sn = sc.n_surf;
    if (W.cnt[CN(4)]) return;
    constexpr unsigned KINDS = CLS == TRC_CLS_MIRROR ? TRC_CLS_MIRROR_KINDS : TRC_CLS_DIFFUSE_KINDS;
    DScene L = sc;
    // (one of TALLY_PARTS copies of the tally buffer per workgroup, also where the tables are beyond LDS and the sums are added per
    // hit: a single copy -- 2.4 MB for a mesh of 1e5 faces, against 38 MB -- measured slower, 33.8 against 26.9 ms per 1e7 rays)
    L.tally = W.tally_part + (size_t)(blockIdx.x % TALLY_PARTS) * (size_t)W.tally_n;
    double *l_tally = lds;
    double *cur = lds + (LDS ? 3 * Sn + 2 : 2);
    for (int i = threadIdx.x; i < (LDS ? 3 * Sn + 2 : 2); i += blockDim.x) l_tally[i] = 0.0;
    if (LDS) {
        double *l_recs = cur; cur += Sn * sc.stride;
        for (int i = threadIdx.x; i < Sn * sc.stride; i += blockDim.x) l_recs[i] = sc.recs[i];
        double *l_opt = cur; cur += 8 * Sn;
        for (int i = threadIdx.x; i < 8 * Sn; i += blockDim.x) l_opt[i] = sc.opt[i];
        double *l_edges = cur; cur += sc.n_fm_edges;
        for (int i = threadIdx.x; i < sc.n_fm_edges; i += blockDim.x) l_edges[i] = sc.fm_edges[i];
        FluxMapDev *l_fms = (FluxMapDev *)cur; cur += (sc.n_fm * sizeof(FluxMapDev) + 7) / 8;
        for (int i = threadIdx.x; i < sc.n_fm; i += blockDim.x) l_fms[i] = sc.fms[i];
        int32_t *l_fm_of = (int32_t *)cur;
        int32_t *l_flags = l_fm_of + Sn;
        for (int i = threadIdx.x; i < Sn; i += blockDim.x) { l_fm_of[i] = sc.fm_of_surf ? sc.fm_of_surf[i] : -1; l_flags[i] = sc.sflags[i]; }
        L.recs = l_recs; L.opt = l_opt; L.fm_edges = l_edges; L.fms = l_fms; L.fm_of_surf = l_fm_of; L.sflags = l_flags;
        cur = (double *)(((uintptr_t)(l_flags + Sn) + 7) & ~(uintptr_t)7);
        if (CLS != TRC_CLS_MIRROR) {        // (no mirror reads a table)
            double *l_extra = cur; cur += sc.n_extra;
            for (int i = threadIdx.x; i < sc.n_extra; i += blockDim.x) l_extra[i] = sc.extra[i];
            L.extra = l_extra;
        }
    }
    double *l_fm = nullptr;         // private copy of the flux-map bins (S.lds_fm_bins of them)
    if (S.lds_fm_bins > 0) {
        l_fm = cur; cur += S.lds_fm_bins;
        for (int i = threadIdx.x; i < S.lds_fm_bins; i += blockDim.x) l_fm[i] = 0.0;
    }
    const double src_energy = P.src ? P.src->energy : 0.0;
    __syncthreads();
    long long nh = (long long)W.cnt[S.hl_cn];
    if (nh > S.hl_room) nh = S.hl_room;     // never read beyond the allocation, whatever the counter says
    const long long padded = (nh + 63) & ~63ll;
    const unsigned wave_g = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    WaveChunk ca = S.static_first ? chunk_init_static_at(S.chunk_act, (unsigned long long)S.act_base0 + (unsigned long long)wave_g * S.chunk_act) : chunk_init(S.chunk_act);
    // this wave's open chunk of the scene's hit buffer, carried over from earlier launches (see k_s_shade)
    WaveChunk hc = chunk_init(S.chunk_hitbuf);
    unsigned long long *hstate = W.hit_state;
    if (P.capture && wave_g < SHADE_MAX_WAVES) {
        const unsigned long long st = hstate[2 * wave_g + 1];
        if ((unsigned)(st >> 32) == S.hit_epoch && (st & 1ull)) { hc.base = hstate[2 * wave_g]; hc.used = (unsigned)(st >> 1) & 0x7FFFFFFFu; hc.open = 1; }
    }
    unsigned n_hit = 0, n_alive = 0;
    const int bounce0 = S.bounce_no;          // every ray of a launch is at the same bounce
    const bool aux_in = !P.src || bounce0 > 0;     // fresh rays of a source carry (energy, 1, 0): nothing was written for them
    const long long stride = (long long)gridDim.x * blockDim.x;
#if SHC_PREFETCH == 2
    // Two loads stand in front of every hit -- its entry of the list, then the ray record the entry points to.  Both are fetched
    // ahead, in two stages (as in k_s_shade): while hit i is worked on, the record of hit i + 1 (whose entry arrived an iteration
    // ago) and the entry of hit i + 2 are on their way.  22 registers.
    uint32_t slot_a = SQ_INVALID, hs_a = SQ_INVALID, slot_n = SQ_INVALID, hs_n = SQ_INVALID;
    double t_a = TRC_INF, t_n = TRC_INF, ae_n = 0.0, aw_n = 0.0;
    SRayGeo g_n;
    g_n.px = g_n.py = g_n.pz = g_n.dx = g_n.dy = 0.0; g_n.dz = 1.0; g_n.head = SQ_INVALID; g_n.idx = 0u; g_n.tail = 0ull;
    {
        const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
        if (i0 < nh) { slot_n = S.hl_slot[i0]; hs_n = S.hl_surf[i0]; t_n = S.hl_t[i0]; }
        if (i0 + stride < nh) { slot_a = S.hl_slot[i0 + stride]; hs_a = S.hl_surf[i0 + stride]; t_a = S.hl_t[i0 + stride]; }
        if (slot_n != SQ_INVALID) {
            g_n = W.geo[slot_n];
            if (aux_in) { ae_n = W.aux[slot_n].e; if (CLS != TRC_CLS_MIRROR) aw_n = W.aux[slot_n].wl; }
        }
    }
#elif defined(SHC_PREFETCH)
    // the entry of the next iteration is on its way while this one is worked on
    uint32_t slot_n = SQ_INVALID, hs_n = SQ_INVALID;
    double t_n = TRC_INF;
    { const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x; if (i0 < nh) { slot_n = S.hl_slot[i0]; hs_n = S.hl_surf[i0]; t_n = S.hl_t[i0]; } }
#endif
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < padded; i += stride) {
        uint32_t slot = SQ_INVALID, hs = SQ_INVALID;
        double t = TRC_INF;
        SRayGeo g;
        g.px = g.py = g.pz = g.dx = g.dy = 0.0; g.dz = 1.0; g.head = SQ_INVALID; g.idx = 0u; g.tail = 0ull;
        bool have = false;
        double ae = 0.0, aw = 0.0;
#if SHC_PREFETCH == 2
        slot = slot_n; hs = hs_n; t = t_n; g = g_n; ae = ae_n; aw = aw_n;
        have = slot != SQ_INVALID;
        // (everything fetched for this hit is made to arrive here, before the loads for the next ones are issued: see k_s_shade)
        asm volatile("" : "+v"(slot), "+v"(hs), "+v"(t), "+v"(slot_a), "+v"(hs_a), "+v"(t_a));
        asm volatile("" : "+v"(g.px), "+v"(g.py), "+v"(g.pz), "+v"(g.dx), "+v"(g.dy), "+v"(g.dz), "+v"(g.head), "+v"(g.idx), "+v"(g.tail));
        asm volatile("" : "+v"(ae), "+v"(aw));
        slot_n = slot_a; hs_n = hs_a; t_n = t_a;
        if (slot_n != SQ_INVALID) {
            g_n = W.geo[slot_n];
            if (aux_in) { ae_n = W.aux[slot_n].e; if (CLS != TRC_CLS_MIRROR) aw_n = W.aux[slot_n].wl; }
        }
        slot_a = SQ_INVALID; hs_a = SQ_INVALID; t_a = TRC_INF;
        if (i + 2 * stride < nh) { slot_a = S.hl_slot[i + 2 * stride]; hs_a = S.hl_surf[i + 2 * stride]; t_a = S.hl_t[i + 2 * stride]; }
#elif defined(SHC_PREFETCH)
        slot = slot_n; hs = hs_n; t = t_n;
        asm volatile("" : "+v"(slot), "+v"(hs), "+v"(t));
        slot_n = SQ_INVALID; hs_n = SQ_INVALID; t_n = TRC_INF;
        if (i + stride < nh) { slot_n = S.hl_slot[i + stride]; hs_n = S.hl_surf[i + stride]; t_n = S.hl_t[i + stride]; }
#else
        if (i < nh) { slot = S.hl_slot[i]; hs = S.hl_surf[i]; t = S.hl_t[i]; }
#endif
        bool mine = false;
        int s = 0, fl = 0;
        if (slot != SQ_INVALID) {
            if (hs == SQ_INVALID) {
                // general path: the nearest of the ray's linked hits; on equal t the lowest surface index (tracer_engine.py:58-63)
                if (!have) g = W.geo[slot];
                have = true;
                t = TRC_INF;
                int sb = 0x7FFFFFFF;
                for (uint32_t k = g.head; k != SQ_INVALID;) {
                    const SCand c = W.q3n[k];
                    if (c.t < t || (c.t == t && (int)c.surf < sb)) { t = c.t; sb = (int)c.surf; }
                    k = c.next;
                }
                s = sb;
            } else s = (int)hs;
            if ((unsigned)s < (unsigned)Sn) {
                fl = L.sflags[s];
                const int cls = (fl >> TRC_SURF_CLS_SHIFT) & TRC_SURF_CLS_MASK;
                mine = (S.shade_term_cls >= 0 && (fl & TRC_SURF_TERMINAL)) ? (S.shade_term_cls == CLS) : (cls == CLS);
            }
            if (mine && !have) g = W.geo[slot];
        }
        const unsigned long long my_lanes = __ballot(mine);
        if (!my_lanes) continue;            // the unused tail of a chunk of the list, or hits of other classes only
        bool alive = false;
        int ts = -1;                         // the surface this lane's hit is tallied on, with absorbed and incident energy
        double tea = 0.0, tei = 0.0;
        if (mine) {
            n_hit += 1;
            const int prev = bounce0 == 0 ? Sn : (int)((uint32_t)(g.tail >> 32) & ~SQ_SKIP_SELF);      // the surface the ray left; Sn = the source
            double e = src_energy, wl = 0.0;
#if SHC_PREFETCH == 2
            if (aux_in) { e = ae; wl = aw; }
#else
            if (aux_in) { e = W.aux[slot].e; if (CLS != TRC_CLS_MIRROR) wl = W.aux[slot].wl; }
            (void)ae; (void)aw;
#endif
            const double *rec = L.recs + (size_t)s * sc.stride;
            const double hx = g.px + t * g.dx, hy = g.py + t * g.dy, hz = g.pz + t * g.dz;
            double ox = g.dx, oy = g.dy, oz = g.dz, e_out = 0.0;
            if (!(S.shade_term_cls >= 0 && (fl & TRC_SURF_TERMINAL))) {
                double nx, ny, nz;
                if (FLATN) {
                    if (!trc_gm_is_flat(trc_rec_gm_kind(rec))) __builtin_unreachable();
                }
                trc_normal(rec, hx, hy, hz, g.dx, g.dy, g.dz, &nx, &ny, &nz);
                trc_ray_out out[2];
                const unsigned long long rid = P.rid ? P.rid[S.base + (long long)g.idx] : (P.ray_offset + (unsigned long long)(S.base + (long long)g.idx));
                trc_shade_k<KINDS, false>(trc_rec_opt_kind(rec), L.opt + (size_t)s * 8, L.extra, trc_rec_extra_off(rec), trc_rec_extra_len(rec),
                                          rec[2], rec[5], rec[8], g.dx, g.dy, g.dz, e, 1.0, wl, 0.0, nx, ny, nz, P.seed, rid, (uint32_t)(bounce0 + 1), out);
                ox = out[0].dx; oy = out[0].dy; oz = out[0].dz; e_out = out[0].e;
            }
            const double e_abs = e - e_out;
            record_hit<LDS>(L, l_tally, s, e, e_abs, hx, hy, hz, g.dx, g.dy, g.dz, P.capture != 0, prev, &hc, l_fm, false, true);
            ts = s; tea = e_abs; tei = e;        // (the three sums of the surface: below, per wave)
            if (e_out > P.min_energy) {                               // tracer_engine.py:242
                if (bounce0 + 1 >= P.reps) {                          // still alive after the last iteration
                    atomicAdd(&sc.counters[3], 1ull);
                    atomicAdd(sc.energy_left, e_out);
                    if (P.flags & TRC_TRACE_KEEP_LAST) {
                        const unsigned long long q = atomicAdd(&sc.counters[2], 1ull);
                        if ((long long)q < P.last_cap) {
                            P.lx[q] = hx; P.ly[q] = hy; P.lz[q] = hz;
                            P.ldx[q] = ox; P.ldy[q] = oy; P.ldz[q] = oz; P.le[q] = e_out;
                        }
                    }
                } else {
                    alive = true;
                    SRayGeo go;
                    go.px = hx; go.py = hy; go.pz = hz; go.dx = ox; go.dy = oy; go.dz = oz;
                    go.head = SQ_INVALID;          // ready for the next bounce's search
                    go.idx = g.idx;
                    // leaving a flat surface the ray cannot meet it again when its own plane test is certain to give t < 1e-7
                    // (flat_surface.py:39-51: t = -((p - c).n) / (d.n), the hit point p is on the plane up to rounding)
                    uint32_t pw = (uint32_t)s;
                    if (trc_gm_is_flat(trc_rec_gm_kind(rec))) {
                        const double dtn = ox * rec[2] + oy * rec[5] + oz * rec[8];
                        const double vt = rec[2] * (hx - rec[9]) + rec[5] * (hy - rec[10]) + rec[8] * (hz - rec[11]);
                        const double scale = 1.0 + fabs(hx) + fabs(hy) + fabs(hz) + fabs(rec[9]) + fabs(rec[10]) + fabs(rec[11]);
                        if (fabs(dtn) > 1e-6 && fabs(vt) + 1e-12 * scale < 5e-8 * fabs(dtn)) pw |= SQ_SKIP_SELF;
                    }
                    go.tail = sray_tail(bounce0 + 1, pw);
                    W.geo[slot] = go;
                    if (aux_in) W.aux[slot].e = e_out;        // (index and wavelength stay as they are: no optics of these classes changes them)
                    else { SRayAux ao; ao.e = e_out; ao.ref = 1.0; ao.wl = 0.0; ao.pad = 0.0; W.aux[slot] = ao; }
                }
            }
        }
        {
            // The three sums per surface: lanes that share the surface of the first lane still to be served are summed in registers
            // and added once while at least 8 of them do (see k_s_shade); the others add for themselves.  In LDS, or -- a scene whose
            // tables do not fit it: a mesh -- in the workgroup's copy of the tally buffer in global memory, where 64 lanes on one
            // word are served one after the other (the lid over the mesh of 1e5 faces: 10 ms per 5e6 hits on it).
            double *tl = LDS ? l_tally : L.tally;
            unsigned long long todo = (!LDS || Sn <= 64) ? __ballot(ts >= 0) : 0ull;
            for (int round = 0; round < 6 && todo; ++round) {
                const int s0 = __shfl(ts, __ffsll((long long)todo) - 1, 64);
                const bool in = ts == s0;
                const unsigned long long m = __ballot(in);
                if (__popcll(m) < 8) break;
                const double a = wave_sum(in ? tea : 0.0), b = wave_sum(in ? tei : 0.0);
                if (lane_id() == 0) { atomicAdd(&tl[s0], a); atomicAdd(&tl[Sn + s0], b); atomicAdd(&tl[2 * Sn + s0], (double)__popcll(m)); }
                if (in) ts = -1;
                todo &= ~m;
            }
#ifdef SHC_DIAG_NO_TALLY        /* diagnostic builds only: what the scattered sums beyond LDS cost */
            if (LDS)
#endif
            if (ts >= 0) { atomicAdd(&tl[ts], tea); atomicAdd(&tl[Sn + ts], tei); atomicAdd(&tl[2 * Sn + ts], 1.0); }
        }
        if (P.capture) chunk_rebroadcast(hc, __ffsll((long long)my_lanes) - 1);   // hc was advanced by the lanes with a hit only
        const unsigned long long q = chunk_append(&W.cnt[CN(3)], ca, alive, S.act_out, W.act_room);
        if (alive) { if ((long long)q < W.act_room) S.act_out[q] = slot; else W.cnt[CN(4)] = 2ull; n_alive += 1; }
    }
    chunk_close(ca, S.act_out, W.act_room);
    if (P.capture && wave_g < SHADE_MAX_WAVES && lane_id() == 0) {
        hstate[2 * wave_g] = hc.base;
        hstate[2 * wave_g + 1] = ((unsigned long long)S.hit_epoch << 32) | ((unsigned long long)hc.used << 1) | (hc.open ? 1ull : 0ull);
    }
    // real (unpadded) counts of this bounce: hits and rays that go on -- one pair of atomics per workgroup
    {
        const double h = wave_sum((double)n_hit), a = wave_sum((double)n_alive);
        double *spare = l_tally + (LDS ? 3 * Sn : 0);
        if (lane_id() == 0) { atomicAdd(&spare[0], h); atomicAdd(&spare[1], a); }
        __syncthreads();
        if (threadIdx.x == 0) {
            if (spare[0] > 0.0) { atomicAdd(&W.cnt[CN(6)], (unsigned long long)(spare[0] + 0.5)); atomicAdd(&W.cnt[CN(13 + CLS)], (unsigned long long)(spare[0] + 0.5)); }
            if (spare[1] > 0.0) atomicAdd(&W.cnt[CN(7)], (unsigned long long)(spare[1] + 0.5));
        }
    }
    if (LDS)
        for (int i = threadIdx.x; i < 3 * Sn; i += blockDim.x) {
            const double v = l_tally[i];
            if (v != 0.0) atomicAdd(&L.tally[i], v);
        }
    if (l_fm) {
        double *gt = L.tally + 3 * Sn + 2;
        for (int i = threadIdx.x; i < S.lds_fm_bins; i += blockDim.x) {
            const double v = l_fm[i];
            if (v != 0.0) atomicAdd(&gt[i], v);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// k_s_shade_x: the shading stage for rays that carry more than the record of the fast engine -- the imaginary part of a complex
// refractive index, the scene's materials evaluated at the ray's wavelength (Refractive / RefractiveAbsorbant,
// optics_callables.py:726-858, :908-944), a sampled spectrum (LambertianReceiver's polychromatic form, :393-425).  One kernel for
// every optics kind (trc_shade_x = trc_shade + the kinds that read those), all hits of the bounce's list; nothing fetched ahead.
//   Im of the index       the fourth word of the ray's SRayAux, from the bundle's column for a ray not shaded yet
//   materials, sample     do not change along a ray: read from the bundle's columns by the ray's number (SRayGeo.idx)
//   wavelengths
//   spectrum              one value per sample and slot in CarryIn.slot_spec (sample-major, `room` apart); from the bundle's column at the
//                         first interaction.  Scaled per sample by 1 - absorptance(theta, lambda_w) at a polychromatic wall, as a
//                         whole by the optics' factor elsewhere -- what k_ord_bounce does for the ordered engine.
// Given bundles only (a source descriptor makes rays of energy, index 1 and no wavelength).
template <bool LDS>
__global__ __launch_bounds__(SHC_THREADS) void k_s_shade_x(StreamParams S) {
    extern __shared__ double lds[];
    const FastParams &P = S.P;
    const DScene &sc = P.sc;
    const StreamWs &W = S.W;
    const int Sn = sc.n_surf;
    if (W.cnt[CN(4)]) return;
    DScene L = sc;
    L.tally = W.tally_part + (size_t)(blockIdx.x % TALLY_PARTS) * (size_t)W.tally_n;
    double *l_tally = lds;
    double *cur = lds + (LDS ? 3 * Sn + 2 : 2);
    for (int i = threadIdx.x; i < (LDS ? 3 * Sn + 2 : 2); i += blockDim.x) l_tally[i] = 0.0;
    if (LDS) {
        double *l_recs = cur; cur += Sn * sc.stride;
        for (int i = threadIdx.x; i < Sn * sc.stride; i += blockDim.x) l_recs[i] = sc.recs[i];
        double *l_opt = cur; cur += 8 * Sn;
        for (int i = threadIdx.x; i < 8 * Sn; i += blockDim.x) l_opt[i] = sc.opt[i];
        double *l_edges = cur; cur += sc.n_fm_edges;
        for (int i = threadIdx.x; i < sc.n_fm_edges; i += blockDim.x) l_edges[i] = sc.fm_edges[i];
        FluxMapDev *l_fms = (FluxMapDev *)cur; cur += (sc.n_fm * sizeof(FluxMapDev) + 7) / 8;
        for (int i = threadIdx.x; i < sc.n_fm; i += blockDim.x) l_fms[i] = sc.fms[i];
        int32_t *l_fm_of = (int32_t *)cur;
        int32_t *l_flags = l_fm_of + Sn;
        for (int i = threadIdx.x; i < Sn; i += blockDim.x) { l_fm_of[i] = sc.fm_of_surf ? sc.fm_of_surf[i] : -1; l_flags[i] = sc.sflags[i]; }
        L.recs = l_recs; L.opt = l_opt; L.fm_edges = l_edges; L.fms = l_fms; L.fm_of_surf = l_fm_of; L.sflags = l_flags;
        cur = (double *)(((uintptr_t)(l_flags + Sn) + 7) & ~(uintptr_t)7);
        double *l_extra = cur; cur += sc.n_extra;
        for (int i = threadIdx.x; i < sc.n_extra; i += blockDim.x) l_extra[i] = sc.extra[i];
        L.extra = l_extra;
    }
    double *l_fm = nullptr;
    if (S.lds_fm_bins > 0) {
        l_fm = cur; cur += S.lds_fm_bins;
        for (int i = threadIdx.x; i < S.lds_fm_bins; i += blockDim.x) l_fm[i] = 0.0;
    }
    const double src_energy = P.src ? P.src->energy : 0.0;
    __syncthreads();
    long long nh = (long long)W.cnt[S.hl_cn];
    if (nh > S.hl_room) nh = S.hl_room;
    const long long padded = (nh + 63) & ~63ll;
    const unsigned wave_g = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    WaveChunk ca = S.static_first ? chunk_init_static_at(S.chunk_act, (unsigned long long)S.act_base0 + (unsigned long long)wave_g * S.chunk_act) : chunk_init(S.chunk_act);
    WaveChunk hc = chunk_init(S.chunk_hitbuf);
    unsigned long long *hstate = W.hit_state;
    if (P.capture && wave_g < SHADE_MAX_WAVES) {
        const unsigned long long st = hstate[2 * wave_g + 1];
        if ((unsigned)(st >> 32) == S.hit_epoch && (st & 1ull)) { hc.base = hstate[2 * wave_g]; hc.used = (unsigned)(st >> 1) & 0x7FFFFFFFu; hc.open = 1; }
    }
    unsigned n_hit = 0, n_alive = 0;
    const int bounce0 = S.bounce_no;
    const bool aux_in = !P.src || bounce0 > 0;
    const CarryIn &C = S.carry;
    const int nW = C.slot_spec ? C.n_spec : 0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < padded; i += stride) {
        uint32_t slot = SQ_INVALID, hs = SQ_INVALID;
        double t = TRC_INF;
        if (i < nh) { slot = S.hl_slot[i]; hs = S.hl_surf[i]; t = S.hl_t[i]; }
        SRayGeo g;
        g.px = g.py = g.pz = g.dx = g.dy = 0.0; g.dz = 1.0; g.head = SQ_INVALID; g.idx = 0u; g.tail = 0ull;
        bool mine = false;
        int s = 0;
        if (slot != SQ_INVALID) {
            g = W.geo[slot];
            if (hs == SQ_INVALID) {         // general path: the nearest of the ray's linked hits; on equal t the lowest surface index
                t = TRC_INF;
                int sb = 0x7FFFFFFF;
                for (uint32_t k = g.head; k != SQ_INVALID;) {
                    const SCand c = W.q3n[k];
                    if (c.t < t || (c.t == t && (int)c.surf < sb)) { t = c.t; sb = (int)c.surf; }
                    k = c.next;
                }
                s = sb;
            } else s = (int)hs;
            mine = (unsigned)s < (unsigned)Sn;
        }
        const unsigned long long my_lanes = __ballot(mine);
        if (!my_lanes) continue;
        bool alive = false;
        int ts = -1;
        double tea = 0.0, tei = 0.0;
        if (mine) {
            n_hit += 1;
            const bool first = bounce0 == 0;
            const int prev = first ? Sn : (int)((uint32_t)(g.tail >> 32) & ~SQ_SKIP_SELF);
            const long long ray = S.base + (long long)g.idx;            // the ray's place in the bundle's columns
            double e = src_energy, ref = 1.0, wl = 0.0, ref_im = 0.0;
            if (aux_in) { const SRayAux a = W.aux[slot]; e = a.e; ref = a.ref; wl = a.wl; ref_im = a.pad; }
            if (first) ref_im = C.ref_im ? C.ref_im[ray] : 0.0;
            const double *rec = L.recs + (size_t)s * sc.stride;
            double hx = g.px + t * g.dx, hy = g.py + t * g.dy, hz = g.pz + t * g.dz;
            double nx, ny, nz;
            trc_normal(rec, hx, hy, hz, g.dx, g.dy, g.dz, &nx, &ny, &nz);
            const double path = sqrt((hx - g.px) * (hx - g.px) + (hy - g.py) * (hy - g.py) + (hz - g.pz) * (hz - g.pz));
            trc_ray_ext X;
            X.ref_im = ref_im; X.W = nW; X.n_mat = C.mat ? C.n_mat : 0; X.stride = P.n;
            X.mat = C.mat ? C.mat + ray : nullptr;
            X.wl = nW ? C.spec_wl + ray : nullptr;
            X.spec = nW ? C.spec + ray : nullptr;
            long long spec_stride = P.n;                                 // ... of the spectrum as it reaches this hit
            if (nW && !first) { X.spec = C.slot_spec + slot; spec_stride = W.room; }
            trc_ray_out out[2];
            double out_im[2], poly_th;
            const unsigned long long rid = P.rid ? P.rid[ray] : (P.ray_offset + (unsigned long long)ray);
            // (sample wavelengths and spectrum have different strides once the spectrum lives in the slot table: the polychromatic
            // wall reads both, so its integral is taken here with the two strides, the optics get the wavelengths' stride)
            int n_out;
            if (nW && !first && trc_rec_opt_kind(rec) == TRC_OPT_LAMBERTIAN_POLYCHROMATIC) {
                // trc_shade_x's branch (optics_callables.py:406-425) with the spectrum read from the slot table
                out[0].blk = 0; out[1].blk = 1; out[0].back = out[1].back = 0.0; out[0].sf = out[1].sf = 1.0; out[0].shift = out[1].shift = 0.0;
                out[0].ref = ref;
                out_im[0] = out_im[1] = ref_im;
                const double dn = g.dx * nx + g.dy * ny + g.dz * nz;
                const double wx = dn * nx, wy = dn * ny, wz = dn * nz;
                const double th = acos(sqrt(wx * wx + wy * wy + wz * wz));
                const double *tab = L.extra + trc_rec_extra_off(rec);
                double en = 0.0, y0 = 0.0, x0 = 0.0;
                for (int w = 0; w < nW; ++w) {
                    const double xw = X.wl[(long long)w * P.n];
                    const double yw = X.spec[(long long)w * spec_stride] * (1.0 - trc_poly_absorptance(tab, th, xw));
                    if (w > 0) en += (xw - x0) * (yw + y0) / 2.0;
                    x0 = xw; y0 = yw;
                }
                double u0, u1, ax, ay, az;
                trc_uniform_pair(P.seed, rid, (uint32_t)(bounce0 + 1), 0, &u0, &u1);
                trc_pillbox_dir_u(u0, u1, 1.57079632679489661923, &ax, &ay, &az);
                trc_rotation_to_z_apply(nx, ny, nz, ax, ay, az, &out[0].dx, &out[0].dy, &out[0].dz);
                out[0].e = en;
                poly_th = th;
                n_out = 1;
            } else
                n_out = trc_shade_x(trc_rec_opt_kind(rec), L.opt + (size_t)s * 8, L.extra, trc_rec_extra_off(rec), trc_rec_extra_len(rec),
                                    rec[2], rec[5], rec[8], g.dx, g.dy, g.dz, e, ref, wl, path, nx, ny, nz, P.seed, rid, (uint32_t)(bounce0 + 1),
                                    X, out, out_im, &poly_th);
            (void)n_out;        // (scenes whose optics split rays go to the ordered engine)
            const bool volume = out[0].back > 0.0;          // scattered in the medium before the surface (see fast_shade)
            if (volume) { hx -= out[0].back * g.dx; hy -= out[0].back * g.dy; hz -= out[0].back * g.dz; }
            const double e_out = out[0].e;
            const double e_abs = e - e_out;
            unsigned long long hslot = ~0ull;
            record_hit<LDS>(L, l_tally, s, e, e_abs, hx, hy, hz, g.dx, g.dy, g.dz, P.capture != 0, prev, &hc, l_fm, volume, true, &hslot);
            if (!volume) { ts = s; tea = e_abs; tei = e; }
            if (nW && C.hit_x && hslot != ~0ull) {        // a captured hit of a polychromatic ray: wavelengths, spectrum in, spectrum out
                const double *tab = L.extra + trc_rec_extra_off(rec);
                for (int w = 0; w < nW; ++w) {
                    const double xw = X.wl[(long long)w * P.n], yin = X.spec[(long long)w * spec_stride];
                    const double f = poly_th >= 0.0 ? 1.0 - trc_poly_absorptance(tab, poly_th, xw) : out[0].sf;
                    C.hit_x[(long long)w * C.hit_x_cap + (long long)hslot] = xw;
                    C.hit_x[(long long)(nW + w) * C.hit_x_cap + (long long)hslot] = yin;
                    C.hit_x[(long long)(2 * nW + w) * C.hit_x_cap + (long long)hslot] = yin * f;
                }
            }
            const double ox = out[0].dx, oy = out[0].dy, oz = out[0].dz;
            if (e_out > P.min_energy) {                               // tracer_engine.py:242
                if (bounce0 + 1 >= P.reps) {
                    atomicAdd(&sc.counters[3], 1ull);
                    atomicAdd(sc.energy_left, e_out);
                    if (P.flags & TRC_TRACE_KEEP_LAST) {
                        const unsigned long long q = atomicAdd(&sc.counters[2], 1ull);
                        if ((long long)q < P.last_cap) {
                            P.lx[q] = hx + out[0].shift * nx; P.ly[q] = hy + out[0].shift * ny; P.lz[q] = hz + out[0].shift * nz;
                            P.ldx[q] = ox; P.ldy[q] = oy; P.ldz[q] = oz; P.le[q] = e_out;
                        }
                    }
                } else {
                    alive = true;
                    SRayGeo go;
                    go.px = hx + out[0].shift * nx; go.py = hy + out[0].shift * ny; go.pz = hz + out[0].shift * nz;
                    go.dx = ox; go.dy = oy; go.dz = oz;
                    go.head = SQ_INVALID;
                    go.idx = g.idx;
                    uint32_t pw = volume ? (uint32_t)prev : (uint32_t)s;        // the surface the ray leaves (a volume event: still the one before)
                    if (!volume && out[0].shift == 0.0 && trc_gm_is_flat(trc_rec_gm_kind(rec))) {
                        const double dtn = ox * rec[2] + oy * rec[5] + oz * rec[8];
                        const double vt = rec[2] * (hx - rec[9]) + rec[5] * (hy - rec[10]) + rec[8] * (hz - rec[11]);
                        const double scale = 1.0 + fabs(hx) + fabs(hy) + fabs(hz) + fabs(rec[9]) + fabs(rec[10]) + fabs(rec[11]);
                        if (fabs(dtn) > 1e-6 && fabs(vt) + 1e-12 * scale < 5e-8 * fabs(dtn)) pw |= SQ_SKIP_SELF;
                    }
                    go.tail = sray_tail(bounce0 + 1, pw);
                    W.geo[slot] = go;
                    SRayAux ao;
                    ao.e = e_out; ao.ref = out[0].ref; ao.wl = wl; ao.pad = out_im[0];
                    W.aux[slot] = ao;
                    if (nW) {       // the spectrum goes on with the ray (RayBundle.inherit, ray_bundle.py:117-143), the optics' changes applied
                        const double *tab = L.extra + trc_rec_extra_off(rec);
                        for (int w = 0; w < nW; ++w) {
                            const double f = poly_th >= 0.0 ? 1.0 - trc_poly_absorptance(tab, poly_th, X.wl[(long long)w * P.n]) : out[0].sf;
                            C.slot_spec[(long long)w * W.room + slot] = X.spec[(long long)w * spec_stride] * f;
                        }
                    }
                }
            }
        }
        {   // the three sums per surface, per wave where lanes share a surface (see k_s_shade_c)
            double *tl = LDS ? l_tally : L.tally;
            unsigned long long todo = (!LDS || Sn <= 64) ? __ballot(ts >= 0) : 0ull;
            for (int round = 0; round < 6 && todo; ++round) {
                const int s0 = __shfl(ts, __ffsll((long long)todo) - 1, 64);
                const bool in = ts == s0;
                const unsigned long long m = __ballot(in);
                if (__popcll(m) < 8) break;
                const double a = wave_sum(in ? tea : 0.0), b = wave_sum(in ? tei : 0.0);
                if (lane_id() == 0) { atomicAdd(&tl[s0], a); atomicAdd(&tl[Sn + s0], b); atomicAdd(&tl[2 * Sn + s0], (double)__popcll(m)); }
                if (in) ts = -1;
                todo &= ~m;
            }
            if (ts >= 0) { atomicAdd(&tl[ts], tea); atomicAdd(&tl[Sn + ts], tei); atomicAdd(&tl[2 * Sn + ts], 1.0); }
        }
        if (P.capture) chunk_rebroadcast(hc, __ffsll((long long)my_lanes) - 1);
        const unsigned long long q = chunk_append(&W.cnt[CN(3)], ca, alive, S.act_out, W.act_room);
        if (alive) { if ((long long)q < W.act_room) S.act_out[q] = slot; else W.cnt[CN(4)] = 2ull; n_alive += 1; }
    }
    chunk_close(ca, S.act_out, W.act_room);
    if (P.capture && wave_g < SHADE_MAX_WAVES && lane_id() == 0) {
        hstate[2 * wave_g] = hc.base;
        hstate[2 * wave_g + 1] = ((unsigned long long)S.hit_epoch << 32) | ((unsigned long long)hc.used << 1) | (hc.open ? 1ull : 0ull);
    }
    {
        const double h = wave_sum((double)n_hit), a = wave_sum((double)n_alive);
        double *spare = l_tally + (LDS ? 3 * Sn : 0);
        if (lane_id() == 0) { atomicAdd(&spare[0], h); atomicAdd(&spare[1], a); }
        __syncthreads();
        if (threadIdx.x == 0) {
            if (spare[0] > 0.0) { atomicAdd(&W.cnt[CN(6)], (unsigned long long)(spare[0] + 0.5)); atomicAdd(&W.cnt[CN(13 + TRC_CLS_GENERAL)], (unsigned long long)(spare[0] + 0.5)); }
            if (spare[1] > 0.0) atomicAdd(&W.cnt[CN(7)], (unsigned long long)(spare[1] + 0.5));
        }
    }
    if (LDS)
        for (int i = threadIdx.x; i < 3 * Sn; i += blockDim.x) {
            const double v = l_tally[i];
            if (v != 0.0) atomicAdd(&L.tally[i], v);
        }
    if (l_fm) {
        double *gt = L.tally + 3 * Sn + 2;
        for (int i = threadIdx.x; i < S.lds_fm_bins; i += blockDim.x) {
            const double v = l_fm[i];
            if (v != 0.0) atomicAdd(&gt[i], v);
        }
    }
}

const void *trc_shade_carry_kernel(bool lds) { return lds ? (const void *)k_s_shade_x<true> : (const void *)k_s_shade_x<false>; }

const void *trc_shade_lean_kernel(int cls, bool flat, bool lds) {
#define SHC_PICK(C) (flat ? (lds ? (const void *)k_s_shade_c<C, true, true> : (const void *)k_s_shade_c<C, true, false>) \
                          : (lds ? (const void *)k_s_shade_c<C, false, true> : (const void *)k_s_shade_c<C, false, false>))
    if (cls == TRC_CLS_MIRROR) return SHC_PICK(TRC_CLS_MIRROR);
    if (cls == TRC_CLS_DIFFUSE) return SHC_PICK(TRC_CLS_DIFFUSE);
#undef SHC_PICK
    return nullptr;
}
