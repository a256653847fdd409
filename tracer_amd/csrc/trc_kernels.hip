// trc_kernels.hip -- CDNA4 (gfx950) kernels and the C-ABI of include/tracer_amd.h.
//
// Two engines over the same per-ray core (trc_core.h):
//   * fast engine  (k_trace_fast): persistent wavefronts.  A lane carries one ray through all of its
//     bounces in registers; when a lane's ray dies (miss, absorbed, culled) the wave refills its dead
//     lanes from its slice of the ray-id range (ballot + prefix rank), so every iteration of the wave
//     loop traces 64 live segments.  Surface frames, Kd nodes and the Buie table are staged in LDS;
//     per-surface tallies are privatised in LDS and flushed once per workgroup.
//   * ordered engine (k_ord_*): one launch per bounce, reproducing the reference's bundle order and
//     parent indices for RayTree (tracer_engine.py:218-274); dead-ray compaction by wavefront ballot +
//     prefix sum, surface-major ordering by a stable radix sort of (culled, surface, block) keys.
//
// No CPU fallback lives here: without a GPU every entry point fails with TRC_ERR_DEVICE.

#include <cstring>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <new>
#include <map>
#include <mutex>
#include <unordered_map>
#include <deque>
#include <utility>

#include "trc_core.h"
#include "trc_bounds.h"
#include "trc_footprint.h"
#include "trc_device.h"

// ================================================================================================
// error handling
// ================================================================================================
static thread_local std::string g_last_error;

static int trc_fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess)                                                                           \
            return trc_fail(TRC_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),      \
                            __FILE__, __LINE__);                                                        \
    } while (0)

#define TRC_TRY(expr)                   \
    do {                                \
        int _s = (expr);                \
        if (_s != TRC_OK) return _s;    \
    } while (0)

extern "C" const char *trc_last_error(void) { return g_last_error.c_str(); }
extern "C" int trc_abi_version(void) { return TRC_ABI_VERSION; }

// ================================================================================================
// host-side objects
// ================================================================================================
struct trc_ctx {
    int device;
    hipStream_t stream;
    hipEvent_t ev0, ev1;
    int n_cu;
};

struct trc_scene {
    trc_ctx *ctx;
    int32_t n_surf, stride, n_extra;
    std::vector<trc_surface_desc> surfs;
    std::vector<double> extra_h;
    bool splits;  // some optics can emit two rays per hit
    bool carries; // some optics read what only rays of the ordered engine carry (complex indices, spectra)
    // device buffers
    double *d_recs, *d_opt, *d_extra;
    int32_t *d_sflags;
    int32_t *d_kd_a, *d_kd_b, *d_kd_leaf, *d_kd_always;
    double *d_kd_split;
    int32_t kd_nodes, kd_nleaf, kd_nalways;
    double kd_bounds[6];
    bool has_kd;
    trc_accel_host accel;
    bool accel_ok, accel_kd_ok;
    float *d_a_sbox;
    float *d_a_obb;
    uint64_t geom_version;     // bumped whenever the surfaces are (re)uploaded: tables derived from their poses are stale
    uint32_t *d_a_nodes;
    uint16_t *d_a_leaf;
    int32_t *d_a_unbounded;
    uint16_t *d_a_bleaf;
    uint16_t *d_a_goff, *d_a_glist;
    uint32_t *d_a_bg_off, *d_a_bg_occ;
    float *d_a_bg_ent;
    int32_t *d_a_bg_apart;
    int32_t *d_a_gapart;
    struct StreamEngine *stream_eng;   // slots of the streaming fast engine (trc_stream.inc), allocated on first use
    struct OrdScratch *ord_scratch;    // scratch of the ordered engine's bounce loop, kept between calls
    double *d_tally;
    int64_t tally_n;
    bool transfer_on;          // keep the transfer matrix (trc_scene_enable_transfer)
    int64_t tr_off;
    std::vector<FluxMapDev> fms_h;
    std::vector<double> fm_edges_h;
    std::vector<int32_t> fm_of_surf_h;
    int32_t *d_fm_of_surf;
    FluxMapDev *d_fms;
    double *d_fm_edges;
    unsigned long long *d_counters;
    double *d_energy_left;     // = (double *)(d_counters + 5)
    trc_source_desc *d_src_buf; // device copy of the source descriptor of the call in progress (kept between calls)
    trc_source_desc src_host;   // ... and what it holds (src_host_ok): a Monte-Carlo loop hands over the same descriptor every call
    bool src_host_ok;
    unsigned long long cnt_host[8];   // host copy of d_counters as trc_trace_fast left them (cnt_host_ok): the next call does not read
    bool cnt_host_ok;                 // them back before it starts.  Every other writer of d_counters updates or drops the copy.
    double *d_last[7];                // device side of trc_trace_fast's `last` bundle, kept between calls (seven hipMalloc / hipFree per
    int64_t d_last_cap;               // call were a millisecond of a Monte-Carlo loop's 1e6-ray calls)
    int64_t hit_dirty_to;             // entries [0, hit_dirty_to) of the hit buffer may have been written since it was last emptied: emptying
                                      // 2e8 entries for the 6e6 a trace used was 0.9 GB of memset, twice per call of the public entry point
    int64_t hit_cap;      // entries allocated: the capacity asked for + TRC_HIT_SLACK
    int64_t hit_cap_user;
    uint32_t hit_epoch;   // bumped whenever the cursor is reset: chunks left open by earlier launches are stale
    uint32_t hit_chunk;   // entries a wave of the streaming engine's shading kernels reserves per atomic (set with the capacity)
    int32_t *d_h_surf;
    double *d_h[8];
    double *d_hx;         // polychromatic hits: hx_cols more columns of the hit buffer (column k at d_hx + k * hx_cap): per hit the W sample
    int hx_cols;          // wavelengths, the W samples of the spectrum that arrived and the W that left (k_s_shade_x)
    int64_t hx_cap;
};

// what rays of the ordered engine carry beyond the nine columns: rows of one matrix `pay` (row r of ray i at pay[r * n + i]):
// [Im of the refractive index] [Re, Im of each material at the ray's wavelength] [wavelengths of the spectrum] [spectrum]
struct PayLayout {
    int has_im = 0, n_mat = 0, W = 0;
    __host__ __device__ int rows() const { return has_im + 2 * n_mat + 2 * W; }
    __host__ __device__ int r_mat() const { return has_im; }
    __host__ __device__ int r_wl() const { return has_im + 2 * n_mat; }
    __host__ __device__ int r_spec() const { return has_im + 2 * n_mat + W; }
};

struct Level {
    int64_t n_total, n_live;
    double *x, *y, *z, *dx, *dy, *dz, *e, *ref, *wl;
    uint64_t *rid;
    int64_t *parent;
    int32_t *surf;
    double *pay;
    char *slab;       // the one allocation the columns above are carved from
};

struct trc_result {
    trc_ctx *ctx;
    std::vector<Level> levels;
    PayLayout lay;
};

// Device memory.  A freed block of up to POOL_BLOCK_MAX bytes is kept for the next request of its size class instead of going back
// to the driver: scripts build an engine per run, and an ordered trace of 1e5 rays spent 3 of its 10 ms in hipMalloc / hipFree (a
// block of a few MB costs ~0.3 ms to map and as much to release).  Size classes are an eighth of an octave apart (at most 12.5 %
// over the request); at most POOL_KEEP bytes wait idle, and a failed hipMalloc empties the pool and tries again.  Larger requests
// (the hit buffers and ray tables of 1e7-ray runs and beyond) go to hipMalloc as they are and come straight back: their cost is
// nothing next to the run, and hipMalloc of a *rounded* large size was seen to stall for 1.2 s.  Keeping a block waits for the
// device as hipFree would have.  TRC_DEV_POOL=0 turns the pool off.
struct DevPool {
    std::mutex mu;
    std::unordered_map<void *, std::pair<size_t, int>> live;      // block handed out -> (bytes allocated, device)
    std::multimap<std::pair<int, size_t>, void *> idle;           // (device, bytes) -> block waiting
    size_t idle_bytes = 0;
    size_t idle_large = 0;          // bytes of idle blocks beyond POOL_BLOCK_MAX (kept by exact size)
    std::deque<void *> large_order; // ... in the order they became idle
    int enabled = -1;
};
static DevPool g_pool;
static const size_t POOL_KEEP = (size_t)2 << 30;
static const size_t POOL_BLOCK_MAX = (size_t)128 << 20;
static const size_t POOL_KEEP_LARGE = (size_t)2 << 30;       // idle bytes of blocks beyond POOL_BLOCK_MAX (the oldest make room for a new one)

static bool pool_enabled() {
    if (g_pool.enabled < 0) {
        const char *e = getenv("TRC_DEV_POOL");
        g_pool.enabled = (e && e[0] == '0') ? 0 : 1;
    }
    return g_pool.enabled == 1;
}

static size_t pool_class(size_t bytes) {
    if (bytes <= 256) return 256;
    const int k = 63 - __builtin_clzll((unsigned long long)bytes);      // 2^k <= bytes
    const size_t step = (size_t)1 << (k - 3);
    return (bytes + step - 1) & ~(step - 1);
}

static void pool_trim() {
    std::vector<void *> gone;
    {
        std::lock_guard<std::mutex> g(g_pool.mu);
        for (auto &kv : g_pool.idle) gone.push_back(kv.second);
        g_pool.idle.clear();
        g_pool.large_order.clear();
        g_pool.idle_bytes = 0;
        g_pool.idle_large = 0;
    }
    for (void *p : gone) (void)hipFree(p);
}

static hipError_t pool_alloc(void **out, size_t bytes) {
    *out = nullptr;
    // large requests are not rounded -- hipMalloc of a rounded size (2 GiB, 4 GiB, 7 x 256 MiB ...) was seen to take 1.2 s where
    // the exact size takes milliseconds -- but a freed large block waits for the next request of exactly its size: a Monte-Carlo
    // loop asks for the same 0.9 GB level of 1e7 source rays call after call, and mapping it anew was 10 ms of each
    if (!pool_enabled()) return hipMalloc(out, bytes);
    if (bytes > POOL_BLOCK_MAX) {
        int dev_l = 0;
        (void)hipGetDevice(&dev_l);
        {
            std::lock_guard<std::mutex> g(g_pool.mu);
            auto it = g_pool.idle.find(std::make_pair(dev_l, bytes));
            if (it != g_pool.idle.end()) {
                *out = it->second;
                g_pool.idle.erase(it);
                g_pool.idle_large -= bytes;
                for (auto q = g_pool.large_order.begin(); q != g_pool.large_order.end(); ++q)
                    if (*q == *out) { g_pool.large_order.erase(q); break; }
                g_pool.live[*out] = std::make_pair(bytes, dev_l);
                return hipSuccess;
            }
        }
        hipError_t el = hipMalloc(out, bytes);
        if (el != hipSuccess) {
            (void)hipGetLastError();
            pool_trim();
            el = hipMalloc(out, bytes);
            if (el != hipSuccess) return el;
        }
        std::lock_guard<std::mutex> g(g_pool.mu);
        g_pool.live[*out] = std::make_pair(bytes, dev_l);
        return hipSuccess;
    }
    const size_t cls = pool_class(bytes);
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> g(g_pool.mu);
        auto it = g_pool.idle.find(std::make_pair(dev, cls));
        if (it != g_pool.idle.end()) {
            *out = it->second;
            g_pool.idle.erase(it);
            g_pool.idle_bytes -= cls;
            g_pool.live[*out] = std::make_pair(cls, dev);
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(out, cls);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        pool_trim();
        e = hipMalloc(out, cls);
        if (e != hipSuccess) return e;
    }
    std::lock_guard<std::mutex> g(g_pool.mu);
    g_pool.live[*out] = std::make_pair(cls, dev);
    return hipSuccess;
}

static void pool_free(void *p) {
    size_t cls = 0;
    int dev = 0;
    bool keep = false;
    {
        std::lock_guard<std::mutex> g(g_pool.mu);
        auto it = g_pool.live.find(p);
        if (it != g_pool.live.end()) {
            cls = it->second.first;
            dev = it->second.second;
            g_pool.live.erase(it);
            keep = pool_enabled() && (cls <= POOL_BLOCK_MAX ? g_pool.idle_bytes + cls <= POOL_KEEP : cls <= POOL_KEEP_LARGE);
        }
    }
    if (!keep) { (void)hipFree(p); return; }
    (void)hipDeviceSynchronize();        // nothing in flight may still use the block when the next owner gets it
    std::vector<void *> evict;
    {
        std::lock_guard<std::mutex> g(g_pool.mu);
        if (cls > POOL_BLOCK_MAX) {
            // large blocks wait by exact size; the oldest of them make room
            while (g_pool.idle_large + cls > POOL_KEEP_LARGE && !g_pool.large_order.empty()) {
                void *old = g_pool.large_order.front();
                g_pool.large_order.pop_front();
                for (auto it = g_pool.idle.begin(); it != g_pool.idle.end(); ++it)
                    if (it->second == old) { g_pool.idle_large -= it->first.second; g_pool.idle.erase(it); evict.push_back(old); break; }
            }
            g_pool.large_order.push_back(p);
            g_pool.idle_large += cls;
        } else g_pool.idle_bytes += cls;
        g_pool.idle.insert(std::make_pair(std::make_pair(dev, cls), p));
    }
    for (void *q : evict) (void)hipFree(q);
}

template <class T>
static int dev_alloc(T **p, size_t n) {
    *p = nullptr;
    if (n == 0) n = 1;
    hipError_t e = pool_alloc((void **)p, n * sizeof(T));
    if (e != hipSuccess)
        return trc_fail(TRC_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", n * sizeof(T), hipGetErrorString(e));
    return TRC_OK;
}
template <class T>
static void dev_free(T *&p) {
    if (p) pool_free((void *)p);
    p = nullptr;
}

// ------------------------------------------------------------------------------------------------
// Wave-cooperative fast path (k_trace_coop).  Same candidates, same exact float64 tests and the same winner as
// trc_nearest_accel32 / brute force -- the work is only distributed differently over the 64 lanes:
//   refill   dead lanes take fresh rays; a ray with no candidate at all (misses the Kd root box and the boxes of
//            the always-relevant surfaces) is finished on the spot and its lane refilled again, up to 3 passes,
//            so that the lanes entering the walk are (almost) all doing useful work;
//   walk     each lane walks the tree for its own ray (single precision, packed 4-byte LDS stack) and only lists
//            the leaves it crosses;
//   boxes    all lanes drain the leaf lists together: one (ray, leaf) item per lane per round (prefix sum +
//            binary search), box test of every surface of the leaf;
//   exact    survivors are queued; all lanes drain that queue with trc_intersect (float64) and combine per ray
//            with LDS atomics: minimal t, then the lowest surface index among equal t (tracer_engine.py:58-63).
// ------------------------------------------------------------------------------------------------
#define COOP_LEAFCAP 16      // leaves listed per lane between two drains
#define COOP_E 128           // exact-test queue entries per wave
#define COOP_MAX_NODES 16384 // stack entries are 4 bytes: node (14 bits) | axis code (2 bits) | interval end (16 bits)
#define COOP_MAX_DEPTH 24
#define COOP_FIXED_BYTES (COOP_E * 8 + 64 * 8 + COOP_E * 4 + 64 * 4 + 64 * 4 + 64 * 4 + COOP_LEAFCAP * 64 * 2)
#define COOP_WAVE_BYTES(DEPTH) ((size_t)(DEPTH) * 64 * 4 + COOP_FIXED_BYTES)

struct CoopLds {
    double *exq_t;                // [COOP_E]
    unsigned long long *best_t;   // [64]
    uint32_t *exq;                // [COOP_E]
    int *best_s;                  // [64]
    int *dirty;                   // [64]
    uint32_t *pre;                // [64] inclusive prefix of the leaf-list lengths
    uint16_t *lst;                // [COOP_LEAFCAP][64] leaves crossed by each lane's ray
    uint32_t *stack;              // [depth][64]
};

__device__ __forceinline__ CoopLds coop_carve(char *base) {
    CoopLds W;
    W.exq_t = (double *)base; base += COOP_E * 8;
    W.best_t = (unsigned long long *)base; base += 64 * 8;
    W.exq = (uint32_t *)base; base += COOP_E * 4;
    W.best_s = (int *)base; base += 64 * 4;
    W.dirty = (int *)base; base += 64 * 4;
    W.pre = (uint32_t *)base; base += 64 * 4;
    W.lst = (uint16_t *)base; base += COOP_LEAFCAP * 64 * 2;
    W.stack = (uint32_t *)base;
    return W;
}

// LDS traffic between lanes of one wave: order it for the compiler and the hardware
#define WAVE_SYNC()                                              \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
    } while (0)

// the float64 ray of every lane, kept in the owner's registers and read by other lanes with cross-lane shuffles
struct CoopRay {
    double px, py, pz, dx, dy, dz;
};

// drain the exact-test queue (wave-uniform call)
__device__ __forceinline__ void coop_drain_exact(const CoopLds &W, int ecount, const CoopRay &ray, const double *recs, int stride,
                                                 const double *extra, unsigned lane) {
    if (ecount == 0) return;
    WAVE_SYNC();
    for (int base = 0; base < ecount; base += 64) {
        int i = base + (int)lane;
        const bool valid = i < ecount;
        uint32_t en = valid ? W.exq[i] : 0u;
        int L = (int)(en >> 16), sidx = (int)(en & 0xFFFFu);
        // all lanes shuffle (the source lane must be active), only the valid ones compute
        double qx = __shfl(ray.px, L, 64), qy = __shfl(ray.py, L, 64), qz = __shfl(ray.pz, L, 64);
        double ex = __shfl(ray.dx, L, 64), ey = __shfl(ray.dy, L, 64), ez = __shfl(ray.dz, L, 64);
        if (valid) {
            double t = trc_intersect(recs + (size_t)sidx * stride, extra, qx, qy, qz, ex, ey, ez);
            if (!(t > 0.0) || !(t < TRC_INF)) t = TRC_INF;    // t == 0 is not a hit (tracer_engine.py:58)
            W.exq_t[i] = t;
            if (t < TRC_INF) {
                unsigned long long bits = (unsigned long long)__double_as_longlong(t);   // t > 0: bit order = numeric order
                unsigned long long old = atomicMin(&W.best_t[L], bits);
                if (bits < old) W.dirty[L] = 1;
            }
        }
    }
    WAVE_SYNC();
    if (W.dirty[lane]) { W.best_s[lane] = 0x7FFFFFFF; W.dirty[lane] = 0; }   // a strictly nearer hit: forget the old surface
    WAVE_SYNC();
    for (int base = 0; base < ecount; base += 64) {
        int i = base + (int)lane;
        if (i < ecount) {
            double t = W.exq_t[i];
            if (t < TRC_INF) {
                uint32_t en = W.exq[i];
                int L = (int)(en >> 16), sidx = (int)(en & 0xFFFFu);
                if ((unsigned long long)__double_as_longlong(t) == W.best_t[L]) atomicMin(&W.best_s[L], sidx);   // lowest index wins ties
            }
        }
    }
    WAVE_SYNC();
}

// append (ray lane, surface) to the exact queue for the lanes whose `want` is set (wave-uniform call)
__device__ __forceinline__ void coop_push_exact(const CoopLds &W, int &ecount, bool want, uint32_t entry, const CoopRay &ray,
                                                const double *recs, int stride, const double *extra, unsigned lane) {
    unsigned long long m = __ballot(want);
    if (!m) return;
    int add = __popcll(m);
    if (ecount + add > COOP_E) { coop_drain_exact(W, ecount, ray, recs, stride, extra, lane); ecount = 0; }
    if (want) W.exq[ecount + __popcll(m & ((1ull << lane) - 1ull))] = entry;
    ecount += add;
}

// drain the per-lane leaf lists: items (ray lane, leaf) are dealt to the 64 lanes round by round (wave-uniform call)
__device__ __forceinline__ void coop_drain_leaves(const trc_accel_view &A, const CoopLds &W, unsigned cnt, int &ecount,
                                                  const trc_ray32 &mine, const CoopRay &ray, const double *recs, int stride,
                                                  const double *extra, unsigned lane) {
    // inclusive prefix sum of the list lengths
    unsigned incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        unsigned o = __shfl_up(incl, off, 64);
        if ((int)lane >= off) incl += o;
    }
    const unsigned total = __shfl(incl, 63, 64);
    if (total == 0) return;
    W.pre[lane] = incl;
    WAVE_SYNC();
    for (unsigned base = 0; base < total; base += 64) {
        const unsigned e = base + lane;
        const bool valid = e < total;
        unsigned lc = 0, off = 0, L = 0;
        trc_ray32 r;
        r.ox = r.oy = r.oz = r.ix = r.iy = r.iz = r.dx = r.dy = r.dz = 0.0f;
        if (valid) {
            // first lane whose inclusive prefix exceeds e
            unsigned lo = 0, hi = 63;
            while (lo < hi) {
                unsigned mid = (lo + hi) >> 1;
                if (W.pre[mid] > e) hi = mid; else lo = mid + 1;
            }
            L = lo;
            unsigned j = e - (L ? W.pre[L - 1] : 0u);
            unsigned node = W.lst[j * 64 + L];
            off = A.nodes[2 * node];
            lc = A.nodes[2 * node + 1] >> 2;
        }
        r.ox = __shfl(mine.ox, (int)L, 64); r.oy = __shfl(mine.oy, (int)L, 64); r.oz = __shfl(mine.oz, (int)L, 64);
        r.ix = __shfl(mine.ix, (int)L, 64); r.iy = __shfl(mine.iy, (int)L, 64); r.iz = __shfl(mine.iz, (int)L, 64);
        // box tests of the leaf's surfaces, four at a time (independent LDS loads in flight), hits kept as a bit mask;
        // the wave-level queue append happens once per round, for the (rare) set bits
        for (unsigned kb = 0; __ballot(kb < lc); kb += 32) {        // the hit mask holds 32 surfaces at a time
            unsigned hits = 0;
            for (unsigned k0 = kb; k0 < kb + 32 && __ballot(k0 < lc); k0 += 4) {
                unsigned sid[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) sid[q] = (k0 + q < lc) ? (unsigned)A.leaf_surfs[off + k0 + q] : 0u;
                bool h[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) h[q] = trc_box_hit32(A.sbox + 6 * (size_t)sid[q], r);
#pragma unroll
                for (int q = 0; q < 4; ++q) if (h[q] && k0 + q < lc) hits |= 1u << (k0 + q - kb);
            }
            while (__ballot(hits != 0)) {
                bool want = hits != 0;
                unsigned k = want ? kb + (unsigned)__ffs((int)hits) - 1u : 0u;
                unsigned sidx = want ? (unsigned)A.leaf_surfs[off + k] : 0u;
                coop_push_exact(W, ecount, want, (L << 16) | sidx, ray, recs, stride, extra, lane);
                hits &= hits - 1u;
            }
        }
    }
    WAVE_SYNC();
}

// fresh ray for a lane: from the source descriptor or from the given bundle
template <int KIND = -1>
__device__ __forceinline__ void fast_new_ray(const FastParams &P, const double *buie, long long id, double &px, double &py,
                                             double &pz, double &dx, double &dy, double &dz, double &e, double &ref, double &wl,
                                             unsigned long long &rid, const trc_buie_fast *bf = nullptr) {
    rid = P.rid ? P.rid[id] : (P.ray_offset + (unsigned long long)id);
    if (P.src) {
        trc_source_ray_t<KIND>(P.src, buie, buie ? buie + TRC_BUIE_TABLE : nullptr, P.seed, rid, &px, &py, &pz, &dx, &dy, &dz, bf);
        e = P.src->energy; ref = 1.0; wl = 0.0;
    } else {
        px = P.x[id]; py = P.y[id]; pz = P.z[id];
        dx = P.dx[id]; dy = P.dy[id]; dz = P.dz[id];
        e = P.e[id];
        ref = P.ref ? P.ref[id] : 1.0;
        wl = P.wl ? P.wl[id] : 0.0;
    }
}

// Generic fast kernel: float64 Kd traversal / brute force straight from trc_core.h; scene records (+ Kd arrays)
// staged in LDS when they fit in 64 KiB, read from global memory otherwise.  Used when the single-precision data
// cannot be built or do not fit (very large scenes), and as a cross-check of the cooperative kernel.
//   LDS (doubles): [recs S*stride][kd_split nodes][buie 639][tally 3S+2] then int32: [kd_a][kd_b][leaf][always]
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_trace_fast(FastParams P) {
    extern __shared__ double lds[];
    const DScene &sc = P.sc;
    const int S = sc.n_surf;
    const int tid = threadIdx.x;
    const unsigned lane = lane_id();

    const double *recs = sc.recs;
    const double *kd_split = sc.kd_split;
    const int32_t *kd_a = sc.kd_a, *kd_b = sc.kd_b, *kd_leaf = sc.kd_leaf, *kd_always = sc.kd_always;
    double *cursor = lds;
    if (P.lds_scene) {
        double *l_recs = cursor; cursor += (size_t)S * sc.stride;
        for (int i = tid; i < S * sc.stride; i += THREADS) l_recs[i] = sc.recs[i];
        recs = l_recs;
        if (sc.has_kd) {
            double *l_split = cursor; cursor += sc.kd_nodes;
            for (int i = tid; i < sc.kd_nodes; i += THREADS) l_split[i] = sc.kd_split[i];
            kd_split = l_split;
        }
    }
    const double *buie = nullptr;
    if (P.src) {
        const int NB = TRC_BUIE_TABLE;
        double *l_buie = cursor; cursor += TRC_BUIE_STAGED;
        if (P.src->kind == TRC_SRC_BUIE_DISK || P.src->kind == TRC_SRC_BUIE_RECT) {
            for (int i = tid; i < NB; i += THREADS) l_buie[i] = P.src->buie[i];
            if (tid == 0) trc_buie_aureole_consts(P.src->buie, l_buie + NB);
        }
        buie = l_buie;
    }
    double *l_tally = cursor;
    if (P.lds_tally) {
        cursor += 3 * S + 2;
        for (int i = tid; i < 3 * S + 2; i += THREADS) l_tally[i] = 0.0;
    }
    if (P.lds_scene && sc.has_kd) {
        int32_t *ic = (int32_t *)cursor;
        int32_t *l_a = ic; ic += sc.kd_nodes;
        int32_t *l_b = ic; ic += sc.kd_nodes;
        int32_t *l_leaf = ic; ic += sc.kd_nleaf;
        int32_t *l_alw = ic;
        for (int i = tid; i < sc.kd_nodes; i += THREADS) { l_a[i] = sc.kd_a[i]; l_b[i] = sc.kd_b[i]; }
        for (int i = tid; i < sc.kd_nleaf; i += THREADS) l_leaf[i] = sc.kd_leaf[i];
        for (int i = tid; i < sc.kd_nalways; i += THREADS) l_alw[i] = sc.kd_always[i];
        kd_a = l_a; kd_b = l_b; kd_leaf = l_leaf; kd_always = l_alw;
    }
    __syncthreads();

    const bool accel = sc.has_kd && (P.flags & TRC_TRACE_ACCEL);
    trc_kd_view kd = make_kd_view(sc, kd_a, kd_b, kd_split, kd_leaf, kd_always);

    // this wave's slice of the ray-id range
    const long long n_waves = (long long)gridDim.x * (THREADS >> 6);
    const long long wave = (long long)blockIdx.x * (THREADS >> 6) + (tid >> 6);
    const long long chunk = (P.n + n_waves - 1) / n_waves;
    long long next = wave * chunk;
    long long end = next + chunk;
    if (end > P.n) end = P.n;
    if (next > end) next = end;

    bool alive = false;
    double px = 0, py = 0, pz = 0, dx = 0, dy = 0, dz = 0, e = 0, ref = 1.0, wl = 0.0;
    unsigned long long rid = 0;
    int bounce = 0;
    int prev = S;              // surface the ray left (transfer matrix); S = the source
    double nseg = 0.0, nhit = 0.0;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    for (;;) {
        // ---- refill dead lanes with fresh rays (ballot + prefix rank) ----
        unsigned long long need = __ballot(!alive);
        if (next < end && need) {
            long long id = next + __popcll(need & lt_mask);
            if (!alive && id < end) {
                fast_new_ray(P, buie, id, px, py, pz, dx, dy, dz, e, ref, wl, rid);
                bounce = 0;
                prev = S;
                alive = true;
            }
            next += __popcll(need);
        }
        if (!__ballot(alive)) break;
        if (!alive) continue;

        // ---- one segment ----
        nseg += 1.0;
        double t;
        int s;
        if (accel) {
            LocalKdStack stk;
            trc_nearest_kd(kd, stk, recs, sc.stride, sc.extra, px, py, pz, dx, dy, dz, &t, &s);
        } else {
            trc_nearest_brute(recs, sc.stride, S, sc.extra, px, py, pz, dx, dy, dz, &t, &s);
        }
        if (s < 0) { alive = false; continue; }
        nhit += 1.0;
        if (P.lds_tally) alive = fast_shade<true>(P, recs, l_tally, t, s, px, py, pz, dx, dy, dz, e, ref, wl, rid, bounce, prev);
        else alive = fast_shade<false>(P, recs, l_tally, t, s, px, py, pz, dx, dy, dz, e, ref, wl, rid, bounce, prev);
    }

    // ---- flush ----
    nseg = wave_sum(nseg);
    nhit = wave_sum(nhit);
    if (P.lds_tally) {
        if (lane == 0) { atomicAdd(&l_tally[3 * S], nseg); atomicAdd(&l_tally[3 * S + 1], nhit); }
        __syncthreads();
        for (int i = tid; i < 3 * S + 2; i += THREADS) {
            double v = l_tally[i];
            if (v != 0.0) atomicAdd(&sc.tally[i], v);
        }
    } else if (lane == 0) {
        atomicAdd(&sc.tally[3 * S], nseg);
        atomicAdd(&sc.tally[3 * S + 1], nhit);
    }
}

// Cooperative fast kernel (see the block comment above).  Exact tests read the surface records from global
// memory (they are rare); everything the candidate search touches lives in LDS:
//   doubles [buie 639][tally 3S+2] | float [sbox 6S] | u32 [nodes 2n] | i32 [always][unbounded] | u16 [leaf] |
//   16-byte aligned per-wave regions of COOP_WAVE_BYTES(depth)
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_trace_coop(FastParams P) {
    extern __shared__ double lds[];
    const DScene &sc = P.sc;
    const int S = sc.n_surf;
    const int tid = threadIdx.x;
    const unsigned lane = lane_id();
    const double *recs = sc.recs;

    double *cursor = lds;
    const double *buie = nullptr;
    if (P.src) {
        const int NB = TRC_BUIE_TABLE;
        double *l_buie = cursor; cursor += TRC_BUIE_STAGED;
        if (P.src->kind == TRC_SRC_BUIE_DISK || P.src->kind == TRC_SRC_BUIE_RECT) {
            for (int i = tid; i < NB; i += THREADS) l_buie[i] = P.src->buie[i];
            if (tid == 0) trc_buie_aureole_consts(P.src->buie, l_buie + NB);
        }
        buie = l_buie;
    }
    double *l_tally = cursor;
    cursor += 3 * S + 2;
    for (int i = tid; i < 3 * S + 2; i += THREADS) l_tally[i] = 0.0;

    const bool kd32 = sc.a_kd_ok && (P.flags & TRC_TRACE_ACCEL);
    trc_accel_view A;
    CoopLds W;
    {
        float *l_sbox = (float *)cursor;
        for (int i = tid; i < 6 * S; i += THREADS) l_sbox[i] = sc.a_sbox[i];
        uint32_t *l_nodes = (uint32_t *)(l_sbox + 6 * S);
        const int nn = kd32 ? sc.kd_nodes : 1;          // without a Kd-tree: one leaf holding every bounded surface
        for (int i = tid; i < 2 * nn; i += THREADS) l_nodes[i] = kd32 ? sc.a_nodes[i] : sc.a_bnodes[i];
        int32_t *l_alw = (int32_t *)(l_nodes + 2 * nn);
        const int na = kd32 ? sc.kd_nalways : 0;
        for (int i = tid; i < na; i += THREADS) l_alw[i] = sc.kd_always[i];
        int32_t *l_unb = l_alw + na;
        for (int i = tid; i < sc.a_n_unbounded; i += THREADS) l_unb[i] = sc.a_unbounded[i];
        uint16_t *l_leaf = (uint16_t *)(l_unb + sc.a_n_unbounded);
        const int nl = kd32 ? sc.kd_nleaf : sc.a_n_bleaf;
        for (int i = tid; i < nl; i += THREADS) l_leaf[i] = kd32 ? sc.a_leaf[i] : sc.a_bleaf[i];
        size_t off = (size_t)((char *)(l_leaf + nl) - (char *)lds);
        off = (off + 15) & ~(size_t)15;
        const int depth = kd32 ? (sc.a_kd_depth > 0 ? sc.a_kd_depth : 1) : 1;
        W = coop_carve((char *)lds + off + (size_t)(tid >> 6) * COOP_WAVE_BYTES(depth));
        A.sbox = l_sbox; A.nodes = l_nodes; A.leaf_surfs = l_leaf; A.always = l_alw; A.unbounded = l_unb;
        A.n_always = na; A.n_unbounded = sc.a_n_unbounded; A.n_surf = S; A.has_kd = 1;
#pragma unroll
        for (int i = 0; i < 6; ++i) A.root[i] = kd32 ? sc.a_root[i] : sc.a_broot[i];
        A.delta = sc.a_delta;
#pragma unroll
        for (int i = 0; i < 3; ++i) { A.cen[i] = sc.a_cen[i]; A.slo[i] = sc.a_slo[i]; A.shi[i] = sc.a_shi[i]; }
    }
    __syncthreads();

    // this wave's slice of the ray-id range
    const long long n_waves = (long long)gridDim.x * (THREADS >> 6);
    const long long wave = (long long)blockIdx.x * (THREADS >> 6) + (tid >> 6);
    const long long chunk = (P.n + n_waves - 1) / n_waves;
    long long next = wave * chunk;
    long long end = next + chunk;
    if (end > P.n) end = P.n;
    if (next > end) next = end;

    bool alive = false;       // the lane holds a ray that still needs its segment
    bool prepared = false;    // ... and the ray has candidates: it takes part in the search below
    double px = 0, py = 0, pz = 0, dx = 0, dy = 0, dz = 0, e = 0, ref = 1.0, wl = 0.0;
    unsigned long long rid = 0;
    int bounce = 0;
    int prev = S;              // surface the ray left (transfer matrix); S = the source
    double nseg = 0.0, nhit = 0.0;
    double tb = TRC_INF;      // best hit among the surfaces tested inline (unbounded ones)
    int sb = -1;
    bool in = false, walking = false;
    float tmin = 0.0f, tmax = 0.0f;
    unsigned alw_mask = 0;    // always-relevant surfaces whose box the ray crosses (first 32 of them; the rest always pass)
    trc_ray32 r;
    r.ox = r.oy = r.oz = r.dx = r.dy = r.dz = r.ix = r.iy = r.iz = 0.0f;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const int stride = sc.stride;
    const double *extra = sc.extra;

    for (;;) {
        // ================= refill + prepare: up to 3 passes =================
        for (int pass = 0; pass < 3; ++pass) {
            unsigned long long need = __ballot(!alive);
            if (next < end && need) {
                long long id = next + __popcll(need & lt_mask);
                if (!alive && id < end) {
                    fast_new_ray(P, buie, id, px, py, pz, dx, dy, dz, e, ref, wl, rid);
                    bounce = 0;
                    prev = S;
                    alive = true;
                    prepared = false;
                }
                next += __popcll(need);
            }
            if (!__ballot(alive && !prepared)) break;
            if (alive && !prepared) {
                tb = TRC_INF;
                sb = -1;
                const double vx = px, vy = py, vz = pz;   // names used by TRC_TEST_EXACT
                for (int k = 0; k < A.n_unbounded; ++k) TRC_TEST_EXACT(A.unbounded[k]);
                double t0;
                in = trc_ray32_prepare(A.slo, A.shi, A.cen, px, py, pz, dx, dy, dz, &r, &t0);
                walking = false;
                alw_mask = 0;
                bool cand = false;
                if (in) {
                    walking = trc_kd32_root(A.root, r, &tmin, &tmax);
                    for (int k = 0; k < A.n_always; ++k) {
                        const float *b = A.sbox + 6 * (size_t)A.always[k];
                        bool bounded = !(b[3] == TRC_INF && b[0] == -TRC_INF);
                        if (bounded && (k >= 32 || trc_box_hit32(b, r))) { if (k < 32) alw_mask |= 1u << k; cand = true; }
                    }
                    cand = cand || walking;
                }
                if (cand || sb >= 0) {
                    prepared = true;
                } else {
                    nseg += 1.0;        // the segment hits nothing: finished, the lane can be refilled in the next pass
                    alive = false;
                }
            }
        }
        if (!__ballot(alive)) {
            if (next >= end) break;
            continue;
        }

        // ================= search (wave-cooperative; every lane takes part) =================
        W.best_t[lane] = (unsigned long long)__double_as_longlong(TRC_INF);
        W.best_s[lane] = 0x7FFFFFFF;
        W.dirty[lane] = 0;
        CoopRay ray;
        ray.px = px; ray.py = py; ray.pz = pz; ray.dx = dx; ray.dy = dy; ray.dz = dz;
        int ecount = 0;   // wave-uniform
        const uint32_t me = (uint32_t)lane << 16;
        const bool searching = alive && prepared && in;
        {
            for (int k = 0; k < A.n_always; ++k) {
                bool hit = searching && (k >= 32 ? true : ((alw_mask >> k) & 1u) != 0);
                if (k >= 32) {
                    const float *b = A.sbox + 6 * (size_t)A.always[k];
                    hit = hit && !(b[3] == TRC_INF && b[0] == -TRC_INF) && trc_box_hit32(b, r);
                }
                coop_push_exact(W, ecount, hit, me | (uint32_t)A.always[k], ray, recs, stride, extra, lane);
            }
            bool walk = searching && walking;
            uint32_t node = 0;
            int sp = 0;
            unsigned cnt = 0;     // leaves listed since the last drain
            while (__ballot(walk)) {
                // up to four steps of this lane's walk per wave-level iteration (amortises the loop's wave-level checks)
#pragma unroll 1
                for (int rep = 0; rep < 4 && walk && cnt < COOP_LEAFCAP; ++rep) {
                    uint32_t w0 = A.nodes[2 * node], w1 = A.nodes[2 * node + 1];
                    if ((w1 & 3u) != 3u) {
                        bool push;
                        uint32_t na;
                        float pt;
                        node = trc_kd32_step(w0, w1, r, A.delta, tmin, &tmax, &push, &na, &pt);
                        if (push) {
                            // stack entry: (node << 18) | (axis code << 16) | interval end as the upper half of a float32
                            // rounded UP (the interval only ever grows: conservative)
                            uint32_t bts = __float_as_uint(pt);
                            uint32_t hb = (bts >> 16) + ((bts & 0xFFFFu) ? 1u : 0u);
                            W.stack[sp * 64 + lane] = (na << 16) | (hb & 0xFFFFu);
                            ++sp;
                        }
                    } else {
                        if ((w1 >> 2) != 0u) { W.lst[cnt * 64 + lane] = (uint16_t)node; ++cnt; }
                        if (sp == 0) walk = false;
                        else {
                            --sp;
                            uint32_t en = W.stack[sp * 64 + lane];
                            uint32_t na = en >> 16;
                            node = na >> 2;
                            tmin = trc_kd32_pop_tmin(na & 3u, r, A.delta, tmax);
                            tmax = __uint_as_float(en << 16);
                        }
                    }
                }
                if (__ballot(cnt >= COOP_LEAFCAP)) {    // some lane's list is full: everybody drains
                    coop_drain_leaves(A, W, cnt, ecount, r, ray, recs, stride, extra, lane);
                    cnt = 0;
                }
            }
            coop_drain_leaves(A, W, cnt, ecount, r, ray, recs, stride, extra, lane);
        }
        coop_drain_exact(W, ecount, ray, recs, stride, extra, lane);

        // ================= per lane: result of the segment, shading =================
        if (alive) {
            double t = tb;
            int s = sb;
            double tq = __longlong_as_double((long long)W.best_t[lane]);
            int sq = W.best_s[lane];
            if (tq < TRC_INF && (tq < t || (tq == t && sq < s))) { t = tq; s = sq; }
            nseg += 1.0;
            prepared = false;
            if (s < 0) alive = false;
            else {
                nhit += 1.0;
                alive = fast_shade<true>(P, recs, l_tally, t, s, px, py, pz, dx, dy, dz, e, ref, wl, rid, bounce, prev);
            }
        }
        WAVE_SYNC();
    }

    // ---- flush ----
    nseg = wave_sum(nseg);
    nhit = wave_sum(nhit);
    if (lane == 0) { atomicAdd(&l_tally[3 * S], nseg); atomicAdd(&l_tally[3 * S + 1], nhit); }
    __syncthreads();
    for (int i = tid; i < 3 * S + 2; i += THREADS) {
        double v = l_tally[i];
        if (v != 0.0) atomicAdd(&sc.tally[i], v);
    }
}

#define TRC_STREAM_MIN_RAYS 1048576
#include "trc_stream.inc"

static void scene_free_stream_ws(trc_scene *sc) {
    if (sc->stream_eng) { stream_engine_free(sc->stream_eng); sc->stream_eng = nullptr; }
}

// ================================================================================================
// source generation as a bundle (sources.*_bundle)
// ================================================================================================
__global__ __launch_bounds__(256) void k_source_generate(const trc_source_desc *src, long long n,
                                                         unsigned long long seed, unsigned long long offset,
                                                         double *x, double *y, double *z, double *dx, double *dy,
                                                         double *dz, double *e, uint64_t *rid) {
    __shared__ double l_buie[TRC_BUIE_STAGED];
    for (int i = threadIdx.x; i < TRC_BUIE_TABLE; i += blockDim.x) l_buie[i] = src->buie[i];
    if (threadIdx.x == 0 && (src->kind == TRC_SRC_BUIE_DISK || src->kind == TRC_SRC_BUIE_RECT)) trc_buie_aureole_consts(src->buie, l_buie + TRC_BUIE_TABLE);
    __syncthreads();
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        unsigned long long r = offset + (unsigned long long)i;
        double px, py, pz, qx, qy, qz;
        trc_source_ray(src, l_buie, l_buie + TRC_BUIE_TABLE, seed, r, &px, &py, &pz, &qx, &qy, &qz);
        x[i] = px; y[i] = py; z[i] = pz;
        dx[i] = qx; dy[i] = qy; dz[i] = qz;
        e[i] = src->energy;
        if (rid) rid[i] = r;
    }
}

// float32 start points as k_s_cull evaluates them (trc_source_start32)
__global__ __launch_bounds__(256) void k_source_start32(trc_fp_params F, long long n, unsigned long long seed, unsigned long long offset,
                                                        float *lx, float *ly) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const unsigned long long rid = offset + (unsigned long long)i;
        uint32_t o[4];
        trc_philox4x32_10((uint32_t)rid, (uint32_t)(rid >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
        float x, y;
        trc_fp_position32(F, o, &x, &y);
        lx[i] = x; ly[i] = y;
    }
}

// ================================================================================================
// ordered engine
// ================================================================================================
struct OrdParams {
    DScene sc;
    // current bundle (live rays of the previous level)
    const double *x, *y, *z, *dx, *dy, *dz, *e, *ref, *wl;
    const uint64_t *rid;
    long long n;
    int event;  // index of this interaction (1-based), same for the whole bundle
    int flags;
    double min_energy;
    unsigned long long seed;
    // outputs, 2n slots: child 0 of ray i at slot i, child 1 at slot n+i
    double *ox, *oy, *oz, *odx, *ody, *odz, *oe, *oref, *owl;
    uint64_t *orid;
    uint32_t *key;  // (culled << 30) | (surface << 2) | block, 0xFFFFFFFF = empty slot
    const double *pay;      // PayLayout rows of the current bundle (null: none), rows pay_stride apart (the level's total ray count:
                            // its culled rays sit behind the n live ones)
    long long pay_stride;
    double *opay;           // the children's, rows 2n apart
    PayLayout lay;
};

// pending far children of the single-precision Kd walk (trc_nearest_accel32), in LDS: entry sp of thread t at [sp * 256 + t].
// (In private memory the two stacks of this kernel were 576 bytes of scratch per lane -- 300 MB per dispatch, more than the
// runtime keeps, so every launch allocated and freed it: 7.4 ms per bounce, whatever the number of rays.)
#define ORD_STACK_DEPTH 24
struct LdsStack32 {
    uint32_t *na;
    float *tmax;
    __device__ __forceinline__ void push(int sp, uint32_t n, float t) { na[sp * 256] = n; tmax[sp * 256] = t; }
    __device__ __forceinline__ void pop(int sp, uint32_t *n, float *t) { *n = na[sp * 256]; *t = tmax[sp * 256]; }
};

#define ORD_EMPTY 0xFFFFFFFFu
#define ORD_CULLED_BIT (1u << 30)

// FAST: the conservative single-precision search of the fast engines (boxes of the geometry, packed Kd nodes) in front of the exact
// float64 tests -- the same nearest hit, the same tie rule (trc_nearest_accel32; tests/hostcheck pins it against brute force) --
// with its stack in LDS; otherwise the float64 walk of the caller's tree / brute force, with a stack in private memory.
// MODE 2: the scene stands on the large grid (a mesh): trc_nearest_grid32.
template <int MODE>
__global__ __launch_bounds__(256) void k_ord_bounce(OrdParams P) {
    constexpr bool FAST = MODE == 1;
    __shared__ uint32_t s_na[FAST ? ORD_STACK_DEPTH * 256 : 1];
    __shared__ float s_tm[FAST ? ORD_STACK_DEPTH * 256 : 1];
    const DScene &sc = P.sc;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < P.n;              // (no lane leaves early: the tallies below are summed per wave)
    if (!live) i = 0;
    double px = P.x[i], py = P.y[i], pz = P.z[i], dx = P.dx[i], dy = P.dy[i], dz = P.dz[i], e = P.e[i];
    double ref = P.ref[i], wl = P.wl[i];
    unsigned long long rid = P.rid[i];
    double t;
    int s;
    const bool use_kd = sc.has_kd && (P.flags & TRC_TRACE_ACCEL);
    if (MODE == 2) {
        trc_nearest_grid32(sc, px, py, pz, dx, dy, dz, &t, &s);
    } else if (FAST) {
        const trc_accel_view A = stream_accel_global(sc, use_kd ? 1 : 0);
        LdsStack32 stk;
        stk.na = s_na + threadIdx.x; stk.tmax = s_tm + threadIdx.x;
        trc_nearest_accel32(A, stk, sc.recs, sc.stride, sc.extra, px, py, pz, dx, dy, dz, use_kd, &t, &s);
        if (s < 0) t = 0.0;
    } else if (use_kd) {
        trc_kd_view kd = make_kd_view(sc, sc.kd_a, sc.kd_b, sc.kd_split, sc.kd_leaf, sc.kd_always);
        LocalKdStack stk;
        trc_nearest_kd(kd, stk, sc.recs, sc.stride, sc.extra, px, py, pz, dx, dy, dz, &t, &s);
    } else {
        trc_nearest_brute(sc.recs, sc.stride, sc.n_surf, sc.extra, px, py, pz, dx, dy, dz, &t, &s);
    }
    if (!live) s = -1;
    int ts = -1;                // the surface this lane's hit is tallied on, with absorbed and incident energy
    double tea = 0.0, tei = 0.0;
    uint32_t k0 = ORD_EMPTY, k1 = ORD_EMPTY;
    if (s >= 0) {
        const double *rec = sc.recs + (size_t)s * sc.stride;
        double hx = px + t * dx, hy = py + t * dy, hz = pz + t * dz;
        double nx, ny, nz;
        trc_normal(rec, hx, hy, hz, dx, dy, dz, &nx, &ny, &nz);
        trc_ray_out out[2];
        const double path = sqrt((hx - px) * (hx - px) + (hy - py) * (hy - py) + (hz - pz) * (hz - pz));
        trc_ray_ext X;
        X.ref_im = 0.0; X.W = P.lay.W; X.n_mat = P.lay.n_mat; X.stride = P.pay_stride; X.mat = X.wl = X.spec = nullptr;
        if (P.pay) {
            if (P.lay.has_im) X.ref_im = P.pay[i];
            X.mat = P.pay + (size_t)P.lay.r_mat() * P.pay_stride + i;
            X.wl = P.pay + (size_t)P.lay.r_wl() * P.pay_stride + i;
            X.spec = P.pay + (size_t)P.lay.r_spec() * P.pay_stride + i;
        }
        double out_im[2], poly_th;
        int n_out = trc_shade_x(trc_rec_opt_kind(rec), sc.opt + (size_t)s * 8, sc.extra, trc_rec_extra_off(rec),
                                trc_rec_extra_len(rec), rec[2], rec[5], rec[8], dx, dy, dz, e, ref, wl, path, nx, ny,
                                nz, P.seed, rid, (uint32_t)P.event, X, out, out_im, &poly_th);
        double e_out = out[0].e + (n_out > 1 ? out[1].e : 0.0);
        const bool volume = out[0].back > 0.0;      // scattered in the medium before the surface: nothing recorded there
        if (volume) { hx -= out[0].back * dx; hy -= out[0].back * dy; hz -= out[0].back * dz; }
        if (!volume) { ts = s; tea = e - e_out; tei = e; }         // (the three sums of the surface: below, per wave)
        int fm = (!volume && sc.fm_of_surf) ? sc.fm_of_surf[s] : -1;
        if (fm >= 0) {
            const FluxMapDev &m = sc.fms[fm];
            double u = m.proj[0] * hx + m.proj[1] * hy + m.proj[2] * hz + m.proj[3];
            double v = m.proj[4] * hx + m.proj[5] * hy + m.proj[6] * hz + m.proj[7];
            int iu = trc_bin_index(sc.fm_edges + m.edges_u, m.nu, u);
            int iv = trc_bin_index(sc.fm_edges + m.edges_v, m.nv, v);
            if (iu >= 0 && iv >= 0) atomicAdd(&sc.tally[m.bins + (int64_t)iu * m.nv + iv], e - e_out);
        }
        for (int c = 0; c < n_out; ++c) {
            long long slot = (c == 0) ? i : (P.n + i);
            P.ox[slot] = hx + out[c].shift * nx; P.oy[slot] = hy + out[c].shift * ny; P.oz[slot] = hz + out[c].shift * nz;
            P.odx[slot] = out[c].dx; P.ody[slot] = out[c].dy; P.odz[slot] = out[c].dz;
            P.oe[slot] = out[c].e; P.oref[slot] = out[c].ref; P.owl[slot] = wl;
            P.orid[slot] = (c == 0) ? rid : trc_child_rid(rid, (uint32_t)P.event);
            if (P.opay) {       // children inherit what the ray carries (RayBundle.inherit, ray_bundle.py:117-143), the optics' changes applied
                const size_t S2 = 2 * (size_t)P.n;
                if (P.lay.has_im) P.opay[slot] = out_im[c];
                for (int k = 0; k < 2 * P.lay.n_mat; ++k) P.opay[(size_t)(P.lay.r_mat() + k) * S2 + slot] = X.mat[(size_t)k * P.pay_stride];
                const double *tab = sc.extra + trc_rec_extra_off(rec);
                for (int w = 0; w < P.lay.W; ++w) {
                    const double xw = X.wl[(size_t)w * P.pay_stride];
                    const double f = poly_th >= 0.0 ? 1.0 - trc_poly_absorptance(tab, poly_th, xw) : out[c].sf;
                    P.opay[(size_t)(P.lay.r_wl() + w) * S2 + slot] = xw;
                    P.opay[(size_t)(P.lay.r_spec() + w) * S2 + slot] = X.spec[(size_t)w * P.pay_stride] * f;
                }
            }
            uint32_t k = ((uint32_t)s << 2) | (uint32_t)out[c].blk;
            if (out[c].e <= P.min_energy) k |= ORD_CULLED_BIT;   // tracer_engine.py:242, :270-274
            if (c == 0) k0 = k; else k1 = k;
        }
    }
    {
        // The three sums per surface: global float64 atomics on one word are served one after the other, and a bounce whose rays
        // all land on the receiver of a field put 640 000 x 3 of them on three words -- 7.8 ms for a bounce of 6e5 rays.  The
        // lanes of a wave that share the surface of the first lane still to be served are summed in registers and added once.
        const int S = sc.n_surf;
        unsigned long long todo = __ballot(ts >= 0);
        for (int round = 0; round < 8 && todo; ++round) {
            const int s0 = __shfl(ts, __ffsll((long long)todo) - 1, 64);
            const bool in = ts == s0;
            const unsigned long long m = __ballot(in);
            if (__popcll(m) < 4) break;
            const double a = wave_sum(in ? tea : 0.0), b = wave_sum(in ? tei : 0.0);
            if (lane_id() == 0) { atomicAdd(&sc.tally[s0], a); atomicAdd(&sc.tally[S + s0], b); atomicAdd(&sc.tally[2 * S + s0], (double)__popcll(m)); }
            if (in) ts = -1;
            todo &= ~m;
        }
        if (ts >= 0) { atomicAdd(&sc.tally[ts], tea); atomicAdd(&sc.tally[S + ts], tei); atomicAdd(&sc.tally[2 * S + ts], 1.0); }
    }
    if (live) {
        P.key[i] = k0;
        P.key[P.n + i] = k1;
    }
}

// ---- order-preserving compaction of the occupied slots: ballot + prefix sum ----
// pass 1: per-block counts of occupied and of culled slots
__global__ __launch_bounds__(256) void k_compact_count(const uint32_t *key, long long n, unsigned *blk_cnt,
                                                       unsigned *blk_culled) {
    __shared__ unsigned s_cnt[4], s_cul[4];
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t k = (i < n) ? key[i] : ORD_EMPTY;
    bool occ = k != ORD_EMPTY;
    bool cul = occ && (k & ORD_CULLED_BIT);
    unsigned long long m = __ballot(occ), mc = __ballot(cul);
    int w = threadIdx.x >> 6;
    if (lane_id() == 0) { s_cnt[w] = __popcll(m); s_cul[w] = __popcll(mc); }
    __syncthreads();
    if (threadIdx.x == 0) {
        blk_cnt[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        blk_culled[blockIdx.x] = s_cul[0] + s_cul[1] + s_cul[2] + s_cul[3];
    }
}

// pass 2: exclusive scan of the block counts (one workgroup, wave-level scan over 256-wide chunks)
__global__ __launch_bounds__(256) void k_scan_blocks(const unsigned *blk_cnt, const unsigned *blk_culled,
                                                     long long n_blocks, unsigned long long *blk_off,
                                                     unsigned long long *totals) {
    __shared__ unsigned long long s_wave[4];
    __shared__ unsigned long long s_carry;
    unsigned long long culled = 0;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (long long base = 0; base < n_blocks; base += blockDim.x) {
        long long i = base + threadIdx.x;
        unsigned long long v = (i < n_blocks) ? blk_cnt[i] : 0;
        culled += (i < n_blocks) ? blk_culled[i] : 0;
        // inclusive scan inside the wave
        unsigned long long incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            unsigned long long o = __shfl_up(incl, off, 64);
            if ((int)lane_id() >= off) incl += o;
        }
        int w = threadIdx.x >> 6;
        if (lane_id() == 63) s_wave[w] = incl;
        __syncthreads();
        unsigned long long wave_off = 0;
        for (int k = 0; k < w; ++k) wave_off += s_wave[k];
        unsigned long long carry = s_carry;
        if (i < n_blocks) blk_off[i] = carry + wave_off + incl - v;
        __syncthreads();
        if (threadIdx.x == blockDim.x - 1) s_carry = carry + wave_off + incl;
        __syncthreads();
    }
    // total culled: block reduction
    __shared__ unsigned long long s_cul[4];
    unsigned long long c = culled;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if (lane_id() == 0) s_cul[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        totals[0] = s_carry;
        totals[1] = s_cul[0] + s_cul[1] + s_cul[2] + s_cul[3];
    }
}

// pass 3: scatter (key, slot) of the occupied slots, in slot order
__global__ __launch_bounds__(256) void k_compact_scatter(const uint32_t *key, long long n,
                                                         const unsigned long long *blk_off, uint32_t *ckey,
                                                         uint32_t *cslot) {
    __shared__ unsigned s_cnt[4];
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t k = (i < n) ? key[i] : ORD_EMPTY;
    bool occ = k != ORD_EMPTY;
    unsigned long long m = __ballot(occ);
    int w = threadIdx.x >> 6;
    if (lane_id() == 0) s_cnt[w] = __popcll(m);
    __syncthreads();
    unsigned wave_off = 0;
    for (int q = 0; q < w; ++q) wave_off += s_cnt[q];
    if (occ) {
        unsigned long long pos = blk_off[blockIdx.x] + wave_off + __popcll(m & ((1ull << lane_id()) - 1ull));
        ckey[pos] = k;
        cslot[pos] = (uint32_t)i;
    }
}

struct GatherParams {
    const double *ox, *oy, *oz, *odx, *ody, *odz, *oe, *oref, *owl;
    const uint64_t *orid;
    const uint32_t *skey, *sslot;
    long long m, n_parent;
    double *x, *y, *z, *dx, *dy, *dz, *e, *ref, *wl;
    uint64_t *rid;
    int64_t *parent;
    int32_t *surf;
    const double *opay;     // n_pay rows, 2 n_parent apart
    double *pay;            // n_pay rows, m apart
    int n_pay;
    const double *recs;     // surface records (optics kind): a ray of the scattered block of a scattering medium never reached the
    int stride;             // surface it is filed under -- the level says so (TRC_LEVEL_VOLUME)
};

__global__ __launch_bounds__(256) void k_ord_gather(GatherParams G) {
    long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= G.m) return;
    uint32_t slot = G.sslot[j], k = G.skey[j];
    G.x[j] = G.ox[slot]; G.y[j] = G.oy[slot]; G.z[j] = G.oz[slot];
    G.dx[j] = G.odx[slot]; G.dy[j] = G.ody[slot]; G.dz[j] = G.odz[slot];
    G.e[j] = G.oe[slot]; G.ref[j] = G.oref[slot]; G.wl[j] = G.owl[slot];
    G.rid[j] = G.orid[slot];
    G.parent[j] = (int64_t)(slot >= G.n_parent ? slot - G.n_parent : slot);   // tracer_engine.py:235-236
    {
        const int32_t sj = (int32_t)((k & ~ORD_CULLED_BIT) >> 2);
        // block 0 of a scattering medium is the scattered block (trc_shade: the blocks of the surface interaction come behind it)
        const bool vol = (k & 3u) == 0u && trc_rec_opt_kind(G.recs + (size_t)sj * G.stride) == TRC_OPT_REFRACTIVE_SCATTERING;
        G.surf[j] = vol ? (sj | TRC_LEVEL_VOLUME) : sj;
    }
    for (int r = 0; r < G.n_pay; ++r) G.pay[(size_t)r * G.m + j] = G.opay[(size_t)r * 2 * G.n_parent + slot];
}

__global__ void k_fill_f64(double *p, long long n, double v) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        p[i] = v;
}
__global__ void k_fill_rid(uint64_t *p, long long n, unsigned long long offset) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        p[i] = offset + (unsigned long long)i;
}

// ================================================================================================
// per-surface protocol kernels
// ================================================================================================
__global__ __launch_bounds__(256) void k_gm_intersect(const double *rec, const double *extra, long long n,
                                                      const double *x, const double *y, const double *z,
                                                      const double *dx, const double *dy, const double *dz,
                                                      double *t_out, double *hx, double *hy, double *hz) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double t = trc_intersect(rec, extra, x[i], y[i], z[i], dx[i], dy[i], dz[i]);
    t_out[i] = t;
    if (hx) {
        // FiniteFlatGM / QuadricGM keep v + t*d for every ray (flat_surface.py:156, quadric.py:101)
        hx[i] = x[i] + t * dx[i]; hy[i] = y[i] + t * dy[i]; hz[i] = z[i] + t * dz[i];
    }
}

__global__ __launch_bounds__(256) void k_gm_normals(const double *rec, long long n, const double *hx,
                                                    const double *hy, const double *hz, const double *dx,
                                                    const double *dy, const double *dz, double *nx, double *ny,
                                                    double *nz) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a, b, c;
    trc_normal(rec, hx[i], hy[i], hz[i], dx[i], dy[i], dz[i], &a, &b, &c);
    nx[i] = a; ny[i] = b; nz[i] = c;
}

struct OpticsParams {
    const double *rec, *opt, *extra;
    long long n;
    const double *dx, *dy, *dz, *e, *ref, *wl;
    const uint64_t *rid;
    unsigned long long ray_offset;
    const double *path;             // distance from the ray origin to the hit (null: 0)
    const double *nx, *ny, *nz;
    unsigned long long seed;
    int event;
    double *odx, *ody, *odz, *oe, *oref;
    int32_t *oblk;  // 2n: -1 empty, else block id
    // complex indices, material rows, spectra (trc_shade_x): inputs with rows n apart, outputs 2n apart
    const double *ref_im, *mat, *spec_wl, *spec;
    int n_mat, W;
    double *o_im, *o_spec;
    double *oshift;     // 2n: how far along the oriented normal the outgoing ray starts from the hit point (PeriodicBoundary)
};

__global__ __launch_bounds__(256) void k_optics_apply(OpticsParams P) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n) return;
    trc_ray_out out[2];
    unsigned long long rid = P.rid ? P.rid[i] : (P.ray_offset + (unsigned long long)i);
    const double path = P.path ? P.path[i] : 0.0;
    trc_ray_ext X;
    X.ref_im = P.ref_im ? P.ref_im[i] : 0.0;
    X.W = P.W; X.n_mat = P.n_mat; X.stride = P.n;
    X.mat = P.mat ? P.mat + i : nullptr;
    X.wl = P.spec_wl ? P.spec_wl + i : nullptr;
    X.spec = P.spec ? P.spec + i : nullptr;
    double out_im[2], poly_th;
    int n_out = trc_shade_x(trc_rec_opt_kind(P.rec), P.opt, P.extra, trc_rec_extra_off(P.rec),
                            trc_rec_extra_len(P.rec), P.rec[2], P.rec[5], P.rec[8], P.dx[i], P.dy[i], P.dz[i],
                            P.e[i], P.ref ? P.ref[i] : 1.0, P.wl ? P.wl[i] : 0.0, path, P.nx[i], P.ny[i], P.nz[i], P.seed,
                            rid, (uint32_t)P.event, X, out, out_im, &poly_th);
    for (int c = 0; c < 2; ++c) {
        long long slot = c == 0 ? i : P.n + i;
        if (c < n_out) {
            P.odx[slot] = out[c].dx; P.ody[slot] = out[c].dy; P.odz[slot] = out[c].dz;
            P.oe[slot] = out[c].e; P.oref[slot] = out[c].ref; P.oblk[slot] = out[c].blk;
            P.oshift[slot] = out[c].shift;
            if (P.o_im) P.o_im[slot] = out_im[c];
            if (P.o_spec) {
                const double *tab = P.extra + trc_rec_extra_off(P.rec);
                for (int w = 0; w < P.W; ++w) {
                    const double f = poly_th >= 0.0 ? 1.0 - trc_poly_absorptance(tab, poly_th, X.wl[(size_t)w * P.n]) : out[c].sf;
                    P.o_spec[(size_t)w * 2 * P.n + slot] = X.spec[(size_t)w * P.n] * f;
                }
            }
        } else {
            P.oblk[slot] = -1;
        }
    }
}

// ================================================================================================
// C-ABI: context
// ================================================================================================
extern "C" int trc_ctx_create(int device_id, trc_ctx **out) {
    if (!out) return trc_fail(TRC_ERR_INVALID, "trc_ctx_create: out is NULL");
    *out = nullptr;
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev == 0)
        return trc_fail(TRC_ERR_DEVICE, "no HIP device available (%s): this library has no CPU path",
                        e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= n_dev)
        return trc_fail(TRC_ERR_INVALID, "device %d out of range (%d devices)", device_id, n_dev);
    HIP_TRY(hipSetDevice(device_id));
    trc_ctx *c = new (std::nothrow) trc_ctx();
    if (!c) return trc_fail(TRC_ERR_NOMEM, "out of host memory");
    c->device = device_id;
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&c->ev0));
    HIP_TRY(hipEventCreate(&c->ev1));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    c->n_cu = prop.multiProcessorCount;
    *out = c;
    return TRC_OK;
}

extern "C" int trc_ctx_destroy(trc_ctx *ctx) {
    if (!ctx) return TRC_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipEventDestroy(ctx->ev0);
    (void)hipEventDestroy(ctx->ev1);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    pool_trim();          // blocks kept for reuse go back to the driver with the context
    return TRC_OK;
}

extern "C" int trc_ctx_synchronize(trc_ctx *ctx) {
    if (!ctx) return trc_fail(TRC_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return TRC_OK;
}

extern "C" int trc_ctx_device_name(trc_ctx *ctx, char *buf, int buflen) {
    if (!ctx || !buf || buflen <= 0) return trc_fail(TRC_ERR_INVALID, "bad arguments");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, ctx->device));
    snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return TRC_OK;
}

// ================================================================================================
// C-ABI: scene
// ================================================================================================
static int validate_surface(const trc_surface_desc &s, int idx, int n_extra) {
    if (s.gm_kind < 0 || s.gm_kind >= TRC_GM_KIND_COUNT)
        return trc_fail(TRC_ERR_UNSUPPORTED, "surface %d: geometry kind %d is not in the native table", idx, s.gm_kind);
    if (s.optics_kind < 0 || s.optics_kind >= TRC_OPT_KIND_COUNT)
        return trc_fail(TRC_ERR_UNSUPPORTED, "surface %d: optics kind %d is not in the native table", idx, s.optics_kind);
    if (s.optics_kind == TRC_OPT_REFRACTIVE_SCATTERING) {
        if (s.opt[2] == 0.0) return trc_fail(TRC_ERR_UNSUPPORTED, "surface %d: scattering optics emit one ray per interaction (single_ray)", idx);
        if (s.extra_off < 0 || s.extra_len < 4 || s.extra_off + s.extra_len > n_extra)
            return trc_fail(TRC_ERR_INVALID, "surface %d: scattering optics need s_c1, s_c2, g1, g2 in the extra values", idx);
        if (s.gm_kind == TRC_GM_RECT_PERFORATED || s.gm_kind == TRC_GM_POLYGON)
            return trc_fail(TRC_ERR_UNSUPPORTED, "surface %d: scattering optics share the extra range with the geometry", idx);
    }
    bool opt_table = s.optics_kind == TRC_OPT_REFLECTIVE_SPECTRAL || s.optics_kind == TRC_OPT_LAMBERTIAN_DIRECTIONAL ||
                     s.optics_kind == TRC_OPT_LAMBERTIAN_DIRECTIONAL_SPECTRAL || s.optics_kind == TRC_OPT_FRESNEL_CONDUCTOR ||
                     s.optics_kind == TRC_OPT_LAMBERTIAN_POLYCHROMATIC;
    if (s.optics_kind == TRC_OPT_REFRACTIVE_MATERIAL && (s.opt[4] < 0 || s.opt[5] < 0 || s.opt[4] >= 64 || s.opt[5] >= 64))
        return trc_fail(TRC_ERR_INVALID, "surface %d: material rows out of range", idx);
    bool needs_extra = s.gm_kind == TRC_GM_RECT_PERFORATED || opt_table;
    if (needs_extra && (s.extra_off < 0 || s.extra_len <= 0 || s.extra_off + s.extra_len > n_extra))
        return trc_fail(TRC_ERR_INVALID, "surface %d: extra range [%d,+%d) outside the %d extra values", idx,
                        s.extra_off, s.extra_len, n_extra);
    if (s.gm_kind == TRC_GM_RECT_PERFORATED && opt_table)
        return trc_fail(TRC_ERR_UNSUPPORTED, "surface %d: perforated plate with spectral optics shares one extra range", idx);
    // reference argument checks (flat_surface.py:192-195, :470-480; sphere_surface.py:31-32; cone.py:82-83)
    const double *g = s.gm;
    switch (s.gm_kind) {
    case TRC_GM_RECT: case TRC_GM_RECT_EXTRUDED: case TRC_GM_RECT_PERFORATED:
        if (!(g[0] > 0) || !(g[1] > 0)) return trc_fail(TRC_ERR_INVALID, "surface %d: width and height must be positive", idx);
        break;
    case TRC_GM_ROUND: case TRC_GM_ROUND_CUT:
        if (!(g[0] > 0)) return trc_fail(TRC_ERR_INVALID, "surface %d: radius must be positive", idx);
        if (g[1] >= 0 && !(g[1] < g[0])) return trc_fail(TRC_ERR_INVALID, "surface %d: inner radius must be lower than the outer one", idx);
        break;
    case TRC_GM_SPHERE: case TRC_GM_HEMISPHERE: case TRC_GM_SPHERE_RECT: case TRC_GM_SPHERE_CUT:
        if (!(g[0] > 0)) return trc_fail(TRC_ERR_INVALID, "surface %d: radius must be positive", idx);
        break;
    default: break;
    }
    return TRC_OK;
}

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

static int scene_upload_surfaces(trc_scene *sc) {
    std::vector<double> recs((size_t)sc->n_surf * sc->stride), opt((size_t)sc->n_surf * 8);
    std::vector<int32_t> flags(sc->n_surf);
    for (int i = 0; i < sc->n_surf; ++i) {
        pack_record(sc->surfs[i], recs.data() + (size_t)i * sc->stride, sc->stride);
        for (int k = 0; k < 8; ++k) opt[(size_t)i * 8 + k] = sc->surfs[i].opt[k];
        flags[i] = sc->surfs[i].flags & 0xFFFF;
        // e_out = e (1 - absorptivity) = 0 exactly, and 0 <= min_energy for every min_energy the API accepts
        const trc_surface_desc &sd = sc->surfs[i];
        const int ok = sd.optics_kind;
        const bool plain = ((ok == TRC_OPT_REFLECTIVE || ok == TRC_OPT_ONE_SIDED_REFLECTIVE) && sd.opt[1] == 0.0) ||
                           ((ok == TRC_OPT_REAL_REFLECTIVE || ok == TRC_OPT_ONE_SIDED_REAL_REFLECTIVE) && sd.opt[3] == 0.0) ||
                           (ok == TRC_OPT_LAMBERTIAN && sd.opt[2] == 0.0 && sd.opt[4] == 0.0) || ok == TRC_OPT_LAMBERTIAN_SPECULAR;
        if (plain && sd.opt[0] == 1.0) flags[i] |= TRC_SURF_TERMINAL;
        flags[i] |= trc_shade_class_of(sd) << TRC_SURF_CLS_SHIFT;      // which shading kernel of the streaming engine serves the surface
    }
    HIP_TRY(hipMemcpy(sc->d_recs, recs.data(), recs.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(sc->d_opt, opt.data(), opt.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(sc->d_sflags, flags.data(), flags.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    // conservative single-precision boxes of the bounded surfaces (fast engine)
    trc_accel_build_surfaces(sc->surfs.data(), sc->n_surf, sc->accel);
    sc->geom_version += 1;
    dev_free(sc->d_a_sbox); dev_free(sc->d_a_obb); dev_free(sc->d_a_unbounded); dev_free(sc->d_a_bleaf);
    TRC_TRY(dev_alloc(&sc->d_a_obb, sc->accel.obb.size()));
    HIP_TRY(hipMemcpy(sc->d_a_obb, sc->accel.obb.data(), sc->accel.obb.size() * sizeof(float), hipMemcpyHostToDevice));
    TRC_TRY(dev_alloc(&sc->d_a_bleaf, sc->accel.brute_leaf.size()));
    if (!sc->accel.brute_leaf.empty())
        HIP_TRY(hipMemcpy(sc->d_a_bleaf, sc->accel.brute_leaf.data(), sc->accel.brute_leaf.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    TRC_TRY(dev_alloc(&sc->d_a_sbox, sc->accel.sbox.size()));
    TRC_TRY(dev_alloc(&sc->d_a_unbounded, sc->accel.unbounded.size()));
    HIP_TRY(hipMemcpy(sc->d_a_sbox, sc->accel.sbox.data(), sc->accel.sbox.size() * sizeof(float), hipMemcpyHostToDevice));
    if (!sc->accel.unbounded.empty())
        HIP_TRY(hipMemcpy(sc->d_a_unbounded, sc->accel.unbounded.data(), sc->accel.unbounded.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    trc_accel_build_grid(sc->accel, sc->n_surf);
    { const char *ev = getenv("TRC_GRID_FORCE32"); if (ev && atoi(ev)) sc->accel.grid_ok = false; }      // (measurements: the large grid for a scene the LDS-sized one holds)
    dev_free(sc->d_a_goff); dev_free(sc->d_a_glist); dev_free(sc->d_a_gapart);
    sc->d_a_goff = nullptr; sc->d_a_glist = nullptr; sc->d_a_gapart = nullptr;
    if (sc->accel.grid_ok) {
        TRC_TRY(dev_alloc(&sc->d_a_gapart, sc->accel.grid_apart.size()));
        if (!sc->accel.grid_apart.empty())
            HIP_TRY(hipMemcpy(sc->d_a_gapart, sc->accel.grid_apart.data(), sc->accel.grid_apart.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        TRC_TRY(dev_alloc(&sc->d_a_goff, sc->accel.grid_off.size()));
        TRC_TRY(dev_alloc(&sc->d_a_glist, sc->accel.grid_list.size()));
        HIP_TRY(hipMemcpy(sc->d_a_goff, sc->accel.grid_off.data(), sc->accel.grid_off.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(sc->d_a_glist, sc->accel.grid_list.data(), sc->accel.grid_list.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    dev_free(sc->d_a_bg_off); dev_free(sc->d_a_bg_occ); dev_free(sc->d_a_bg_ent); dev_free(sc->d_a_bg_apart);
    sc->d_a_bg_off = nullptr; sc->d_a_bg_occ = nullptr; sc->d_a_bg_ent = nullptr; sc->d_a_bg_apart = nullptr;
    if (!sc->accel.grid_ok) {      // a scene the LDS-sized grid cannot hold: the 32-bit grid in global memory
        trc_accel_build_grid32(sc->surfs.data(), sc->n_surf, sc->accel);
        if (sc->accel.big_ok) {
            TRC_TRY(dev_alloc(&sc->d_a_bg_off, sc->accel.big_off.size()));
            TRC_TRY(dev_alloc(&sc->d_a_bg_ent, sc->accel.big_ent.size()));
            TRC_TRY(dev_alloc(&sc->d_a_bg_occ, sc->accel.big_occ.size()));
            HIP_TRY(hipMemcpy(sc->d_a_bg_off, sc->accel.big_off.data(), sc->accel.big_off.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(sc->d_a_bg_ent, sc->accel.big_ent.data(), sc->accel.big_ent.size() * sizeof(float), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(sc->d_a_bg_occ, sc->accel.big_occ.data(), sc->accel.big_occ.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            std::vector<float>().swap(sc->accel.big_ent);       // (the host keeps the list itself, not its 48-byte entries)
            TRC_TRY(dev_alloc(&sc->d_a_bg_apart, sc->accel.big_apart.size() + 1));
            if (!sc->accel.big_apart.empty())
                HIP_TRY(hipMemcpy(sc->d_a_bg_apart, sc->accel.big_apart.data(), sc->accel.big_apart.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        }
    }
    sc->accel_ok = true;
    sc->accel_kd_ok = false;   // packed Kd nodes are relative to the scene centre: rebuilt by trc_scene_set_kdtree
    return TRC_OK;
}

static int scene_alloc_tally(trc_scene *sc) {
    dev_free(sc->d_tally);
    int64_t n = 3 * (int64_t)sc->n_surf + 2;
    for (auto &m : sc->fms_h) { m.bins = n; n += (int64_t)m.nu * m.nv; }
    sc->tr_off = -1;
    if (sc->transfer_on) { sc->tr_off = n; n += ((int64_t)sc->n_surf + 1) * sc->n_surf; }
    sc->tally_n = n;
    TRC_TRY(dev_alloc(&sc->d_tally, (size_t)n));
    HIP_TRY(hipMemset(sc->d_tally, 0, (size_t)n * sizeof(double)));
    return TRC_OK;
}

extern "C" int trc_scene_create(trc_ctx *ctx, int32_t n_surf, const trc_surface_desc *surfs, int32_t n_extra,
                                const double *extra, trc_scene **out) {
    if (!ctx || !out || n_surf <= 0 || !surfs) return trc_fail(TRC_ERR_INVALID, "trc_scene_create: bad arguments");
    if (n_surf >= (1 << 28)) return trc_fail(TRC_ERR_INVALID, "too many surfaces");
    *out = nullptr;
    HIP_TRY(hipSetDevice(ctx->device));
    int max_np = 0;
    bool splits = false, carries = false;
    for (int i = 0; i < n_surf; ++i) {
        TRC_TRY(validate_surface(surfs[i], i, n_extra));
        int np = trc_gm_nparams(surfs[i].gm_kind);
        if (np > max_np) max_np = np;
        if (surfs[i].optics_kind == TRC_OPT_REFRACTIVE_HOMOGENOUS && surfs[i].opt[2] == 0.0) splits = true;
        if (surfs[i].optics_kind == TRC_OPT_REFRACTIVE_MATERIAL && surfs[i].opt[0] == 0.0) splits = true;
        if (surfs[i].optics_kind == TRC_OPT_REFRACTIVE_MATERIAL || surfs[i].optics_kind == TRC_OPT_LAMBERTIAN_POLYCHROMATIC) carries = true;
    }
    trc_scene *sc = new (std::nothrow) trc_scene();
    if (!sc) return trc_fail(TRC_ERR_NOMEM, "out of host memory");
    sc->ctx = ctx;
    sc->src_host_ok = false;
    sc->cnt_host_ok = false;
    sc->hit_dirty_to = 0;
    for (int i = 0; i < 7; ++i) sc->d_last[i] = nullptr;
    sc->d_last_cap = 0;
    memset(sc->cnt_host, 0, sizeof(sc->cnt_host));
    sc->n_surf = n_surf;
    sc->stride = TRC_REC_HDR + max_np;
    if ((sc->stride & 1) == 0) sc->stride += 1;  // odd number of doubles: spreads records over LDS banks
    sc->n_extra = n_extra;
    sc->surfs.assign(surfs, surfs + n_surf);
    if (n_extra > 0 && extra) sc->extra_h.assign(extra, extra + n_extra);
    sc->splits = splits;
    sc->carries = carries;
    sc->has_kd = false;
    sc->hit_cap = 0;
    sc->fm_of_surf_h.assign(n_surf, -1);
    int st = TRC_OK;
    do {
        if ((st = dev_alloc(&sc->d_recs, (size_t)n_surf * sc->stride))) break;
        if ((st = dev_alloc(&sc->d_opt, (size_t)n_surf * 8))) break;
        if ((st = dev_alloc(&sc->d_sflags, (size_t)n_surf))) break;
        if ((st = dev_alloc(&sc->d_extra, (size_t)n_extra))) break;
        if ((st = dev_alloc(&sc->d_fm_of_surf, (size_t)n_surf))) break;
        if ((st = dev_alloc(&sc->d_counters, 8))) break;
        sc->d_energy_left = (double *)(sc->d_counters + 5);      // same 64-byte block as the counters: one read-back gets both
        if ((st = scene_upload_surfaces(sc))) break;
        if (n_extra > 0 && hipMemcpy(sc->d_extra, sc->extra_h.data(), (size_t)n_extra * sizeof(double),
                                     hipMemcpyHostToDevice) != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "extra upload failed"); break; }
        if (hipMemcpy(sc->d_fm_of_surf, sc->fm_of_surf_h.data(), (size_t)n_surf * sizeof(int32_t),
                      hipMemcpyHostToDevice) != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "upload failed"); break; }
        if (hipMemset(sc->d_counters, 0, 8 * sizeof(unsigned long long)) != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "memset failed"); break; }
        if ((st = scene_alloc_tally(sc))) break;
    } while (0);
    if (st != TRC_OK) { trc_scene_destroy(sc); return st; }
    *out = sc;
    return TRC_OK;
}

static void scene_free_ord_scratch(trc_scene *sc);
extern "C" int trc_scene_destroy(trc_scene *sc) {
    if (!sc) return TRC_OK;
    (void)hipSetDevice(sc->ctx->device);
    (void)hipStreamSynchronize(sc->ctx->stream);
    scene_free_stream_ws(sc);
    scene_free_ord_scratch(sc);
    dev_free(sc->d_recs); dev_free(sc->d_opt); dev_free(sc->d_extra); dev_free(sc->d_sflags);
    dev_free(sc->d_kd_a); dev_free(sc->d_kd_b); dev_free(sc->d_kd_leaf); dev_free(sc->d_kd_always);
    dev_free(sc->d_a_sbox); dev_free(sc->d_a_obb); dev_free(sc->d_a_nodes); dev_free(sc->d_a_leaf); dev_free(sc->d_a_unbounded); dev_free(sc->d_a_bleaf);
    dev_free(sc->d_a_goff); dev_free(sc->d_a_glist); dev_free(sc->d_a_gapart); dev_free(sc->d_a_bg_off); dev_free(sc->d_a_bg_occ); dev_free(sc->d_a_bg_ent); dev_free(sc->d_a_bg_apart);
    dev_free(sc->d_kd_split); dev_free(sc->d_tally); dev_free(sc->d_fm_of_surf); dev_free(sc->d_fms);
    dev_free(sc->d_fm_edges); dev_free(sc->d_counters); dev_free(sc->d_src_buf); dev_free(sc->d_h_surf); dev_free(sc->d_hx);
    for (int i = 0; i < 8; ++i) dev_free(sc->d_h[i]);
    for (int i = 0; i < 7; ++i) dev_free(sc->d_last[i]);
    delete sc;
    return TRC_OK;
}

extern "C" int trc_scene_update_frames(trc_scene *sc, int32_t n_surf, const double *frames12) {
    if (!sc || !frames12 || n_surf != sc->n_surf) return trc_fail(TRC_ERR_INVALID, "trc_scene_update_frames: bad arguments");
    HIP_TRY(hipSetDevice(sc->ctx->device));
    HIP_TRY(hipStreamSynchronize(sc->ctx->stream));
    for (int i = 0; i < n_surf; ++i) memcpy(sc->surfs[i].frame, frames12 + 12 * (size_t)i, 12 * sizeof(double));
    sc->has_kd = false;        // a Kd-tree set before described the old poses: the caller sets a new one (or does without)
    return scene_upload_surfaces(sc);
}

extern "C" int trc_scene_set_kdtree(trc_scene *sc, const trc_kdtree_desc *kd) {
    if (!sc) return trc_fail(TRC_ERR_INVALID, "scene is NULL");
    HIP_TRY(hipSetDevice(sc->ctx->device));
    HIP_TRY(hipStreamSynchronize(sc->ctx->stream));
    dev_free(sc->d_kd_a); dev_free(sc->d_kd_b); dev_free(sc->d_kd_leaf); dev_free(sc->d_kd_always); dev_free(sc->d_kd_split);
    sc->has_kd = false;
    sc->accel_kd_ok = false;
    if (!kd) return TRC_OK;
    if (kd->n_nodes <= 0 || !kd->flag || !kd->split || !kd->child || !kd->leaf_off || !kd->leaf_cnt)
        return trc_fail(TRC_ERR_INVALID, "trc_scene_set_kdtree: incomplete tree");
    if (kd->n_nodes >= (1 << 29) || kd->n_leaf_surfs >= (1 << 29)) return trc_fail(TRC_ERR_INVALID, "tree too large");
    std::vector<int32_t> a(kd->n_nodes), b(kd->n_nodes);
    // validate and measure depth (the traversal stack is TRC_KD_STACK deep)
    std::vector<int32_t> depth(kd->n_nodes, 0);
    int max_depth = 0;
    for (int i = 0; i < kd->n_nodes; ++i) {
        int f = kd->flag[i];
        if (f < 0 || f > 3) return trc_fail(TRC_ERR_INVALID, "node %d: flag %d", i, f);
        if (f == 3) {
            int off = kd->leaf_off[i], cnt = kd->leaf_cnt[i];
            if (off < 0 || cnt < 0 || off + cnt > kd->n_leaf_surfs) return trc_fail(TRC_ERR_INVALID, "node %d: leaf range", i);
            for (int k = 0; k < cnt; ++k) {
                int s = kd->leaf_surfs[off + k];
                if (s < 0 || s >= sc->n_surf) return trc_fail(TRC_ERR_INVALID, "node %d: surface %d out of range", i, s);
            }
            a[i] = (off << 2) | 3; b[i] = cnt;
        } else {
            int c = kd->child[i];
            if (c <= i || c + 1 >= kd->n_nodes) return trc_fail(TRC_ERR_INVALID, "node %d: child %d out of range", i, c);
            a[i] = (c << 2) | f; b[i] = 0;
            depth[c] = depth[c + 1] = depth[i] + 1;
            if (depth[c] > max_depth) max_depth = depth[c];
        }
    }
    if (max_depth > TRC_KD_STACK) return trc_fail(TRC_ERR_UNSUPPORTED, "Kd-tree depth %d exceeds the traversal stack (%d)", max_depth, TRC_KD_STACK);
    for (int k = 0; k < kd->n_always; ++k)
        if (kd->always_relevant[k] < 0 || kd->always_relevant[k] >= sc->n_surf) return trc_fail(TRC_ERR_INVALID, "always_relevant out of range");
    TRC_TRY(dev_alloc(&sc->d_kd_a, (size_t)kd->n_nodes));
    TRC_TRY(dev_alloc(&sc->d_kd_b, (size_t)kd->n_nodes));
    TRC_TRY(dev_alloc(&sc->d_kd_split, (size_t)kd->n_nodes));
    TRC_TRY(dev_alloc(&sc->d_kd_leaf, (size_t)kd->n_leaf_surfs));
    TRC_TRY(dev_alloc(&sc->d_kd_always, (size_t)kd->n_always));
    HIP_TRY(hipMemcpy(sc->d_kd_a, a.data(), a.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(sc->d_kd_b, b.data(), b.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(sc->d_kd_split, kd->split, (size_t)kd->n_nodes * 8, hipMemcpyHostToDevice));
    if (kd->n_leaf_surfs) HIP_TRY(hipMemcpy(sc->d_kd_leaf, kd->leaf_surfs, (size_t)kd->n_leaf_surfs * 4, hipMemcpyHostToDevice));
    if (kd->n_always) HIP_TRY(hipMemcpy(sc->d_kd_always, kd->always_relevant, (size_t)kd->n_always * 4, hipMemcpyHostToDevice));
    sc->kd_nodes = kd->n_nodes; sc->kd_nleaf = kd->n_leaf_surfs; sc->kd_nalways = kd->n_always;
    memcpy(sc->kd_bounds, kd->bounds, sizeof(sc->kd_bounds));
    sc->has_kd = true;
    dev_free(sc->d_a_nodes); dev_free(sc->d_a_leaf);
    sc->accel_kd_ok = false;
    if (sc->accel_ok && trc_accel_build_kd(kd, sc->accel)) {
        TRC_TRY(dev_alloc(&sc->d_a_nodes, sc->accel.nodes.size()));
        TRC_TRY(dev_alloc(&sc->d_a_leaf, sc->accel.leaf_surfs.size()));
        HIP_TRY(hipMemcpy(sc->d_a_nodes, sc->accel.nodes.data(), sc->accel.nodes.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        if (!sc->accel.leaf_surfs.empty())
            HIP_TRY(hipMemcpy(sc->d_a_leaf, sc->accel.leaf_surfs.data(), sc->accel.leaf_surfs.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        sc->accel_kd_ok = true;
    }
    return TRC_OK;
}

extern "C" int trc_scene_set_fluxmap(trc_scene *sc, int32_t surf, int32_t nu, int32_t nv, const double *u_edges,
                                           const double *v_edges, const double *proj12) {
    if (!sc || surf < 0 || surf >= sc->n_surf || nu <= 0 || nv <= 0 || !u_edges || !v_edges || !proj12)
        return trc_fail(TRC_ERR_INVALID, "trc_scene_set_fluxmap: bad arguments");
    HIP_TRY(hipSetDevice(sc->ctx->device));
    HIP_TRY(hipStreamSynchronize(sc->ctx->stream));
    for (int i = 0; i < nu; ++i) if (!(u_edges[i + 1] > u_edges[i])) return trc_fail(TRC_ERR_INVALID, "u edges must increase");
    for (int i = 0; i < nv; ++i) if (!(v_edges[i + 1] > v_edges[i])) return trc_fail(TRC_ERR_INVALID, "v edges must increase");
    if (sc->fm_of_surf_h[surf] >= 0) return trc_fail(TRC_ERR_INVALID, "surface %d already has a flux map", surf);
    FluxMapDev m;
    m.surf = surf; m.nu = nu; m.nv = nv; m.pad = 0;
    m.edges_u = (int64_t)sc->fm_edges_h.size();
    sc->fm_edges_h.insert(sc->fm_edges_h.end(), u_edges, u_edges + nu + 1);
    m.edges_v = (int64_t)sc->fm_edges_h.size();
    sc->fm_edges_h.insert(sc->fm_edges_h.end(), v_edges, v_edges + nv + 1);
    memcpy(m.proj, proj12, sizeof(m.proj));
    m.bins = 0;
    sc->fm_of_surf_h[surf] = (int32_t)sc->fms_h.size();
    sc->fms_h.push_back(m);
    TRC_TRY(scene_alloc_tally(sc));  // resets the tallies
    dev_free(sc->d_fms); dev_free(sc->d_fm_edges);
    TRC_TRY(dev_alloc(&sc->d_fms, sc->fms_h.size()));
    TRC_TRY(dev_alloc(&sc->d_fm_edges, sc->fm_edges_h.size()));
    HIP_TRY(hipMemcpy(sc->d_fms, sc->fms_h.data(), sc->fms_h.size() * sizeof(FluxMapDev), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(sc->d_fm_edges, sc->fm_edges_h.data(), sc->fm_edges_h.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(sc->d_fm_of_surf, sc->fm_of_surf_h.data(), (size_t)sc->n_surf * sizeof(int32_t), hipMemcpyHostToDevice));
    return TRC_OK;
}

static int scene_reset_hit_buffer(trc_scene *sc);
extern "C" int trc_scene_set_hit_capacity(trc_scene *sc, int64_t capacity) {
    if (!sc || capacity < 0) return trc_fail(TRC_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(sc->ctx->device));
    HIP_TRY(hipStreamSynchronize(sc->ctx->stream));
    // the same capacity again (an engine sizes the buffer before every trace): the buffer is kept and emptied -- freeing and
    // allocating 15 GB per call was a tenth of a second at 1e8 rays
    if (capacity > 0 && capacity == sc->hit_cap_user && sc->d_h_surf) {
        HIP_TRY(hipMemset(sc->d_counters, 0, 2 * sizeof(unsigned long long)));      // cursor and dropped count start over
        sc->cnt_host[0] = sc->cnt_host[1] = 0ull;
        return scene_reset_hit_buffer(sc);
    }
    dev_free(sc->d_h_surf);
    for (int i = 0; i < 8; ++i) dev_free(sc->d_h[i]);
    dev_free(sc->d_hx); sc->hx_cols = 0; sc->hx_cap = 0;      // (made again by the next call that brings spectra)
    sc->hit_cap = 0;
    sc->hit_cap_user = 0;
    sc->hit_epoch += 1;
    HIP_TRY(hipMemset(sc->d_counters, 0, 2 * sizeof(unsigned long long)));
    sc->cnt_host[0] = sc->cnt_host[1] = 0ull;
    if (capacity == 0) return TRC_OK;
    // The streaming engine appends in chunks that stay open between launches: room for what they can leave unused.  The chunk
    // follows the buffer -- 1024 entries per atomic for the buffers of full-size runs (the cursor is one word), less for modest
    // ones -- and the slack is what every wave that can hold an open chunk may leave unused of one: a call whose hits fit the
    // capacity asked for never drops one, whatever its size.
    int64_t chunk = 64;          // (never below a wave's worth: one append of a wave must fit a fresh chunk)
    while (chunk < SQ_HIT_CHUNK && 2048 * (chunk * 2) <= capacity) chunk *= 2;
    sc->hit_chunk = (uint32_t)chunk;
    const int64_t slack = TRC_HIT_HOLDERS * chunk + 64;
    const int64_t alloc = capacity + slack;
    TRC_TRY(dev_alloc(&sc->d_h_surf, (size_t)alloc));
    for (int i = 0; i < 8; ++i) TRC_TRY(dev_alloc(&sc->d_h[i], (size_t)alloc));
    HIP_TRY(hipMemset(sc->d_h_surf, 0xFF, (size_t)alloc * sizeof(int32_t)));     // surface -1: entry not written
    sc->hit_dirty_to = 0;
    sc->hit_cap = alloc;
    sc->hit_cap_user = capacity;
    return TRC_OK;
}

// A buffer of at least `capacity` hits that keeps what it holds: the accountants of a script that traces again before it has
// read the hits of its last call (optics_callables.py:1577-1643 accumulate over calls) are served from the device when they are
// read at last.  Grows by half at least; the chunk size stays (chunks left open by earlier launches go on being filled).
extern "C" int trc_scene_reserve_hits(trc_scene *sc, int64_t capacity) {
    if (!sc || capacity < 0) return trc_fail(TRC_ERR_INVALID, "bad arguments");
    if (!sc->d_h_surf || sc->hit_cap_user == 0) return trc_scene_set_hit_capacity(sc, capacity);
    if (capacity <= sc->hit_cap_user) return TRC_OK;
    HIP_TRY(hipSetDevice(sc->ctx->device));
    HIP_TRY(hipStreamSynchronize(sc->ctx->stream));
    if (capacity < sc->hit_cap_user + sc->hit_cap_user / 2) capacity = sc->hit_cap_user + sc->hit_cap_user / 2;
    unsigned long long c[2];
    HIP_TRY(hipMemcpy(c, sc->d_counters, sizeof(c), hipMemcpyDeviceToHost));
    const int64_t used = (int64_t)c[0] < sc->hit_cap ? (int64_t)c[0] : sc->hit_cap;
    const int64_t alloc = capacity + (sc->hit_cap - sc->hit_cap_user);
    int32_t *n_surf = nullptr;
    double *n_col[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int st = dev_alloc(&n_surf, (size_t)alloc);
    for (int i = 0; i < 8 && st == TRC_OK; ++i) st = dev_alloc(&n_col[i], (size_t)alloc);
    if (st != TRC_OK) { dev_free(n_surf); for (int i = 0; i < 8; ++i) dev_free(n_col[i]); return st; }
    HIP_TRY(hipMemset(n_surf, 0xFF, (size_t)alloc * sizeof(int32_t)));
    if (used > 0) {
        HIP_TRY(hipMemcpy(n_surf, sc->d_h_surf, (size_t)used * sizeof(int32_t), hipMemcpyDeviceToDevice));
        for (int i = 0; i < 8; ++i) HIP_TRY(hipMemcpy(n_col[i], sc->d_h[i], (size_t)used * sizeof(double), hipMemcpyDeviceToDevice));
    }
    dev_free(sc->d_h_surf);
    for (int i = 0; i < 8; ++i) dev_free(sc->d_h[i]);
    dev_free(sc->d_hx); sc->hx_cols = 0; sc->hx_cap = 0;      // (a buffer that grows keeps its hits, not their spectra: the host reads those before)
    sc->d_h_surf = n_surf;
    for (int i = 0; i < 8; ++i) sc->d_h[i] = n_col[i];
    sc->hit_dirty_to = used;
    sc->hit_cap = alloc;
    sc->hit_cap_user = capacity;
    return TRC_OK;
}

// entries of the hit buffer reserved so far (written ones and the unused parts of chunks still open) and its capacity
extern "C" int trc_scene_hits_reserved(trc_scene *sc, int64_t *reserved, int64_t *capacity) {
    if (!sc) return trc_fail(TRC_ERR_INVALID, "scene is NULL");
    if (reserved) {
        if (sc->cnt_host_ok) *reserved = (int64_t)sc->cnt_host[0];
        else {
            HIP_TRY(hipSetDevice(sc->ctx->device));
            HIP_TRY(hipStreamSynchronize(sc->ctx->stream));
            unsigned long long c = 0;
            HIP_TRY(hipMemcpy(&c, sc->d_counters, sizeof(c), hipMemcpyDeviceToHost));
            *reserved = (int64_t)c;
        }
    }
    if (capacity) *capacity = sc->hit_cap_user;
    return TRC_OK;
}

// Page-locked host memory for large results (hit lists, levels of the ray tree): a device-to-host copy into pageable memory is
// staged through the driver's own bounce buffers at a third of the rate.  Blocks are kept by size class when they are freed
// (page-locking 250 MB takes longer than copying them): at most HOST_KEEP bytes wait idle.
struct HostPool {
    std::mutex mu;
    std::unordered_map<void *, size_t> live;
    std::multimap<size_t, void *> idle;
    size_t idle_bytes = 0;
};
static HostPool g_host_pool;
static const size_t HOST_KEEP = (size_t)4 << 30;

extern "C" int trc_host_alloc(int64_t bytes, void **out) {
    if (!out || bytes < 0) return trc_fail(TRC_ERR_INVALID, "bad arguments");
    *out = nullptr;
    const size_t cls = pool_class((size_t)(bytes > 0 ? bytes : 1));
    {
        std::lock_guard<std::mutex> g(g_host_pool.mu);
        auto it = g_host_pool.idle.find(cls);
        if (it != g_host_pool.idle.end()) {
            *out = it->second;
            g_host_pool.idle.erase(it);
            g_host_pool.idle_bytes -= cls;
            g_host_pool.live[*out] = cls;
            return TRC_OK;
        }
    }
    hipError_t e = hipHostMalloc(out, cls, hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        std::vector<void *> gone;
        {
            std::lock_guard<std::mutex> g(g_host_pool.mu);
            for (auto &kv : g_host_pool.idle) gone.push_back(kv.second);
            g_host_pool.idle.clear();
            g_host_pool.idle_bytes = 0;
        }
        for (void *p : gone) (void)hipHostFree(p);
        e = hipHostMalloc(out, cls, hipHostMallocDefault);
        if (e != hipSuccess) return trc_fail(TRC_ERR_NOMEM, "hipHostMalloc(%zu bytes) failed: %s", cls, hipGetErrorString(e));
    }
    std::lock_guard<std::mutex> g(g_host_pool.mu);
    g_host_pool.live[*out] = cls;
    return TRC_OK;
}

extern "C" int trc_host_free(void *p) {
    if (!p) return TRC_OK;
    size_t cls = 0;
    bool keep = false;
    {
        std::lock_guard<std::mutex> g(g_host_pool.mu);
        auto it = g_host_pool.live.find(p);
        if (it == g_host_pool.live.end()) return trc_fail(TRC_ERR_INVALID, "trc_host_free: not a block of trc_host_alloc");
        cls = it->second;
        g_host_pool.live.erase(it);
        keep = g_host_pool.idle_bytes + cls <= HOST_KEEP;
        if (keep) { g_host_pool.idle.insert(std::make_pair(cls, p)); g_host_pool.idle_bytes += cls; }
    }
    if (!keep) (void)hipHostFree(p);
    return TRC_OK;
}

// forget the captured hits: cursor to zero, every entry unwritten, open chunks of the streaming engine stale
static int scene_reset_hit_buffer(trc_scene *sc) {
    sc->hit_epoch += 1;
    const int64_t upto = sc->hit_dirty_to < sc->hit_cap ? sc->hit_dirty_to : sc->hit_cap;
    if (upto > 0) HIP_TRY(hipMemset(sc->d_h_surf, 0xFF, (size_t)upto * sizeof(int32_t)));
    sc->hit_dirty_to = 0;
    return TRC_OK;
}

extern "C" int trc_scene_clear_hits(trc_scene *sc) {
    if (!sc) return trc_fail(TRC_ERR_INVALID, "scene is NULL");
    HIP_TRY(hipSetDevice(sc->ctx->device));
    HIP_TRY(hipStreamSynchronize(sc->ctx->stream));
    HIP_TRY(hipMemset(sc->d_counters, 0, 2 * sizeof(unsigned long long)));
    sc->cnt_host[0] = sc->cnt_host[1] = 0ull;
    return scene_reset_hit_buffer(sc);
}

extern "C" int trc_scene_reset_tallies(trc_scene *sc) {
    if (!sc) return trc_fail(TRC_ERR_INVALID, "scene is NULL");
    HIP_TRY(hipSetDevice(sc->ctx->device));
    HIP_TRY(hipStreamSynchronize(sc->ctx->stream));
    HIP_TRY(hipMemset(sc->d_tally, 0, (size_t)sc->tally_n * sizeof(double)));
    HIP_TRY(hipMemset(sc->d_counters, 0, 8 * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(sc->d_energy_left, 0, sizeof(double)));
    memset(sc->cnt_host, 0, sizeof(sc->cnt_host));
    return scene_reset_hit_buffer(sc);
}

extern "C" int trc_scene_get_tallies(trc_scene *sc, double *absorbed, double *received, int64_t *hits) {
    if (!sc) return trc_fail(TRC_ERR_INVALID, "scene is NULL");
    HIP_TRY(hipSetDevice(sc->ctx->device));
    HIP_TRY(hipStreamSynchronize(sc->ctx->stream));
    const int S = sc->n_surf;
    std::vector<double> t((size_t)3 * S);
    HIP_TRY(hipMemcpy(t.data(), sc->d_tally, t.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int i = 0; i < S; ++i) {
        if (absorbed) absorbed[i] = t[i];
        if (received) received[i] = t[S + i];
        if (hits) hits[i] = (int64_t)(t[2 * S + i] + 0.5);
    }
    return TRC_OK;
}

extern "C" int trc_scene_get_fluxmap(trc_scene *sc, int32_t surf, double *out) {
    if (!sc || surf < 0 || surf >= sc->n_surf || !out) return trc_fail(TRC_ERR_INVALID, "bad arguments");
    int fm = sc->fm_of_surf_h[surf];
    if (fm < 0) return trc_fail(TRC_ERR_INVALID, "surface %d has no flux map", surf);
    HIP_TRY(hipSetDevice(sc->ctx->device));
    HIP_TRY(hipStreamSynchronize(sc->ctx->stream));
    const FluxMapDev &m = sc->fms_h[fm];
    HIP_TRY(hipMemcpy(out, sc->d_tally + m.bins, (size_t)m.nu * m.nv * sizeof(double), hipMemcpyDeviceToHost));
    return TRC_OK;
}

// written entries of the hit buffer (surface >= 0) packed to the front, in buffer order: flags -> exclusive scan -> scatter
__global__ __launch_bounds__(256) void k_hits_flag(const int32_t *surf, long long n, uint32_t *flag) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = surf[i] >= 0 ? 1u : 0u;
}
struct HitPack {
    const int32_t *surf;
    const double *col[8];
    int32_t *o_surf;
    double *o_col[8];
    int want[8];
    const int32_t *sflags;      // TRC_SURF_CAPTURE_LEAN: columns 1 (incident energy) and 5-7 (direction) of the hit were not written
    const double *x;            // n_x more columns (spectra), column k at x + k * x_cap; packed to o_x + k * o_cnt
    double *o_x;
    int n_x;
    long long x_cap, o_cnt;
};
__global__ __launch_bounds__(256) void k_hits_pack(HitPack H, const uint32_t *offs, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t s = H.surf[i];
    if (s < 0) return;
    const uint32_t o = offs[i];
    H.o_surf[o] = s;
#pragma unroll
    for (int k = 0; k < 8; ++k) if (H.want[k]) H.o_col[k][o] = H.col[k][i];
    if (H.sflags[s] & TRC_SURF_CAPTURE_LEAN) {
        if (H.want[1]) H.o_col[1][o] = H.col[0][i];
#pragma unroll
        for (int k = 5; k < 8; ++k) if (H.want[k]) H.o_col[k][o] = 0.0;
    }
    for (int k = 0; k < H.n_x; ++k) H.o_x[(long long)k * H.o_cnt + o] = H.x[(long long)k * H.x_cap + i];
}

// The same, surface by surface: entry src[o] of the buffer goes to place o (src = the entries sorted by surface, stably).
__global__ __launch_bounds__(256) void k_hits_keys(const int32_t *surf, long long n, uint32_t none, uint32_t *key, uint32_t *entry) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t s = surf[i];
    key[i] = s < 0 ? none : (uint32_t)s;          // unwritten entries of open chunks sort behind every surface
    entry[i] = (uint32_t)i;
}
__global__ __launch_bounds__(256) void k_hits_gather(HitPack H, const uint32_t *src, long long cnt) {
    const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= cnt) return;
    const uint32_t i = src[o];
    const int32_t s = H.surf[i];
    H.o_surf[o] = s;
#pragma unroll
    for (int k = 0; k < 8; ++k) if (H.want[k]) H.o_col[k][o] = H.col[k][i];
    if (H.sflags[s] & TRC_SURF_CAPTURE_LEAN) {
        if (H.want[1]) H.o_col[1][o] = H.col[0][i];
#pragma unroll
        for (int k = 5; k < 8; ++k) if (H.want[k]) H.o_col[k][o] = 0.0;
    }
    for (int k = 0; k < H.n_x; ++k) H.o_x[(long long)k * H.o_cnt + o] = H.x[(long long)k * H.x_cap + i];
}

static int scene_get_hits(trc_scene *sc, int64_t *n, int32_t *surf, double *e_abs, double *e_in, double *px,
                          double *py, double *pz, double *dx, double *dy, double *dz, int32_t n_x, double *x_out) {
    if (!sc || !n) return trc_fail(TRC_ERR_INVALID, "bad arguments");
    if (n_x < 0 || (n_x > 0 && (!x_out || n_x != sc->hx_cols || !sc->d_hx)))
        return trc_fail(TRC_ERR_INVALID, "trc_scene_get_hits_x: the hit buffer holds %d spectral columns, %d asked for", sc ? sc->hx_cols : 0, n_x);
    HIP_TRY(hipSetDevice(sc->ctx->device));
    HIP_TRY(hipStreamSynchronize(sc->ctx->stream));
    unsigned long long c[2];
    HIP_TRY(hipMemcpy(c, sc->d_counters, sizeof(c), hipMemcpyDeviceToHost));
    int64_t reserved = (int64_t)c[0];
    if (reserved > sc->hit_cap) reserved = sc->hit_cap;
    *n = 0;
    if (reserved == 0) return TRC_OK;
    if (reserved >= (1ll << 32)) return trc_fail(TRC_ERR_CAPACITY, "more than 2^32 entries in the hit buffer");
    // The reserved range holds unwritten entries (surface -1) where the streaming engine's chunks are still open: the caller
    // gets the written ones, in buffer order (one capturing surface) or surface by surface (several).  They are packed on the device (copying the whole range and picking on the
    // host was 0.11 s for the 6.5e6 receiver hits of an NSTTF step).
    double *dst[8] = {e_abs, e_in, px, py, pz, dx, dy, dz};
    uint32_t *d_flag = nullptr, *d_off = nullptr;
    uint32_t *d_key[2] = {nullptr, nullptr}, *d_ent[2] = {nullptr, nullptr};
    char *d_tmp = nullptr;
    HitPack H;
    memset(&H, 0, sizeof(H));
    int st = TRC_OK;
    do {
        if ((st = dev_alloc(&d_flag, (size_t)reserved)) || (st = dev_alloc(&d_off, (size_t)reserved))) break;
        const unsigned nblk = (unsigned)((reserved + 255) / 256);
        hipLaunchKernelGGL(k_hits_flag, dim3(nblk), dim3(256), 0, sc->ctx->stream, sc->d_h_surf, (long long)reserved, d_flag);
        size_t tmp_bytes = 0;
        if (rocprim::exclusive_scan(nullptr, tmp_bytes, d_flag, d_off, 0u, (size_t)reserved, rocprim::plus<uint32_t>(), sc->ctx->stream) != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "exclusive_scan (size query) failed"); break; }
        if ((st = dev_alloc(&d_tmp, tmp_bytes ? tmp_bytes : 1))) break;
        if (rocprim::exclusive_scan(d_tmp, tmp_bytes, d_flag, d_off, 0u, (size_t)reserved, rocprim::plus<uint32_t>(), sc->ctx->stream) != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "exclusive_scan failed"); break; }
        uint32_t last_off = 0, last_flag = 0;
        if (hipMemcpyAsync(&last_off, d_off + (reserved - 1), 4, hipMemcpyDeviceToHost, sc->ctx->stream) != hipSuccess ||
            hipMemcpyAsync(&last_flag, d_flag + (reserved - 1), 4, hipMemcpyDeviceToHost, sc->ctx->stream) != hipSuccess ||
            hipStreamSynchronize(sc->ctx->stream) != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "hit count readback failed"); break; }
        const int64_t cnt = (int64_t)last_off + (int64_t)last_flag;
        *n = cnt;
        if (cnt == 0 || (!surf && !e_abs && !e_in && !px && !py && !pz && !dx && !dy && !dz)) break;
        H.surf = sc->d_h_surf;
        H.sflags = sc->d_sflags;
        if ((st = dev_alloc(&H.o_surf, (size_t)cnt))) break;
        for (int k = 0; k < 8 && st == TRC_OK; ++k) {
            H.col[k] = sc->d_h[k];
            H.want[k] = dst[k] ? 1 : 0;
            if (dst[k]) st = dev_alloc(&H.o_col[k], (size_t)cnt);
        }
        if (st) break;
        if (n_x > 0) {
            H.x = sc->d_hx; H.n_x = n_x; H.x_cap = sc->hx_cap; H.o_cnt = cnt;
            if ((st = dev_alloc(&H.o_x, (size_t)cnt * (size_t)n_x))) break;
        }
        int n_capture = 0;
        for (int i = 0; i < sc->n_surf; ++i) if (sc->surfs[i].flags & TRC_SURF_CAPTURE_HITS) ++n_capture;
        if (n_capture <= 1) {
            hipLaunchKernelGGL(k_hits_pack, dim3(nblk), dim3(256), 0, sc->ctx->stream, H, (const uint32_t *)d_off, (long long)reserved);
        } else {
            // Several capturing surfaces: the caller wants each one's hits together (accountants), and regrouping eight columns
            // on the host cost 60 ms per 1e6 hits.  A stable radix sort of (surface, entry) pairs over the bits a surface index
            // needs, then one gather: surface by surface, buffer order inside a surface, unwritten entries behind all of them.
            if ((st = dev_alloc(&d_key[0], (size_t)reserved)) || (st = dev_alloc(&d_key[1], (size_t)reserved)) ||
                (st = dev_alloc(&d_ent[0], (size_t)reserved)) || (st = dev_alloc(&d_ent[1], (size_t)reserved))) break;
            hipLaunchKernelGGL(k_hits_keys, dim3(nblk), dim3(256), 0, sc->ctx->stream, sc->d_h_surf, (long long)reserved, (uint32_t)sc->n_surf,
                               d_key[0], d_ent[0]);
            unsigned bits = 1;
            while ((1u << bits) <= (unsigned)sc->n_surf) ++bits;
            size_t sort_bytes = 0;
            if (rocprim::radix_sort_pairs(nullptr, sort_bytes, d_key[0], d_key[1], d_ent[0], d_ent[1], (size_t)reserved, 0, bits, sc->ctx->stream) != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "radix_sort_pairs (size query) failed"); break; }
            dev_free(d_tmp);
            if ((st = dev_alloc(&d_tmp, sort_bytes ? sort_bytes : 1))) break;
            if (rocprim::radix_sort_pairs(d_tmp, sort_bytes, d_key[0], d_key[1], d_ent[0], d_ent[1], (size_t)reserved, 0, bits, sc->ctx->stream) != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "radix_sort_pairs failed"); break; }
            hipLaunchKernelGGL(k_hits_gather, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, sc->ctx->stream, H, (const uint32_t *)d_ent[1], (long long)cnt);
        }
        // one copy per column, all behind the packing on the context's stream (page-locked destinations -- trc_host_alloc -- take
        // them at the rate of the link), one wait
        if (surf && hipMemcpyAsync(surf, H.o_surf, (size_t)cnt * 4, hipMemcpyDeviceToHost, sc->ctx->stream) != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "memcpy failed"); break; }
        for (int k = 0; k < 8; ++k)
            if (dst[k] && hipMemcpyAsync(dst[k], H.o_col[k], (size_t)cnt * 8, hipMemcpyDeviceToHost, sc->ctx->stream) != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "memcpy failed"); break; }
        if (n_x > 0 && hipMemcpyAsync(x_out, H.o_x, (size_t)cnt * (size_t)n_x * 8, hipMemcpyDeviceToHost, sc->ctx->stream) != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "memcpy failed"); break; }
        if (st == TRC_OK && hipStreamSynchronize(sc->ctx->stream) != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "fetching the hits failed"); break; }
    } while (0);
    (void)hipStreamSynchronize(sc->ctx->stream);
    dev_free(d_flag); dev_free(d_off);
    dev_free(d_key[0]); dev_free(d_key[1]); dev_free(d_ent[0]); dev_free(d_ent[1]);
    dev_free(d_tmp);
    dev_free(H.o_surf);
    dev_free(H.o_x);
    for (int k = 0; k < 8; ++k) dev_free(H.o_col[k]);
    return st;
}

extern "C" int trc_scene_get_hits(trc_scene *sc, int64_t *n, int32_t *surf, double *e_abs, double *e_in, double *px,
                                  double *py, double *pz, double *dx, double *dy, double *dz) {
    return scene_get_hits(sc, n, surf, e_abs, e_in, px, py, pz, dx, dy, dz, 0, nullptr);
}

extern "C" int trc_scene_get_hits_x(trc_scene *sc, int64_t *n, int32_t *surf, double *e_abs, double *e_in, double *px,
                                    double *py, double *pz, double *dx, double *dy, double *dz, int32_t n_x, double *x) {
    return scene_get_hits(sc, n, surf, e_abs, e_in, px, py, pz, dx, dy, dz, n_x, x);
}

extern "C" int trc_scene_hit_spectral_columns(trc_scene *sc, int32_t *n_x) {
    if (!sc || !n_x) return trc_fail(TRC_ERR_INVALID, "bad arguments");
    *n_x = sc->d_hx ? sc->hx_cols : 0;
    return TRC_OK;
}

#define BIN_TILE 256
__global__ __launch_bounds__(256) void k_bin_hits(long long n_hits, const int32_t *h_surf, const double *h_e, const double *hx,
                                                  const double *hy, const double *hz, int n_bins, const int32_t *surf_lo,
                                                  const int32_t *surf_hi, const double *ranges6, const int32_t *mode, double *out) {
    __shared__ double l_rng[BIN_TILE * 6];
    __shared__ double l_sum[BIN_TILE];
    __shared__ int32_t l_lo[BIN_TILE], l_hi[BIN_TILE], l_mode[BIN_TILE];
    for (int i = threadIdx.x; i < n_bins; i += blockDim.x) {
        l_lo[i] = surf_lo[i]; l_hi[i] = surf_hi[i]; l_mode[i] = mode[i]; l_sum[i] = 0.0;
        for (int k = 0; k < 6; ++k) l_rng[6 * i + k] = ranges6[6 * i + k];
    }
    __syncthreads();
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_hits; i += (long long)gridDim.x * blockDim.x) {
        const int s = h_surf[i];
        if (s < 0) continue;                       // reserved but unwritten entry of an open chunk
        const double x = hx[i], y = hy[i], z = hz[i], e = h_e[i];
        double ang = atan2(y, x);
        if (ang < 0.0) ang += TRC_TWO_PI;
        const double rad = sqrt(x * x + y * y);
        const double rad9 = rint(rad * 1e9) / 1e9, z9 = rint(z * 1e9) / 1e9;
        for (int j = 0; j < n_bins; ++j) {
            if (s < l_lo[j] || s > l_hi[j]) continue;
            const int m = l_mode[j];
            const double *g = l_rng + 6 * j;
            const double hh = (m & TRC_BIN_ROUND9) ? z9 : z, rr = (m & TRC_BIN_ROUND9) ? rad9 : rad;
            bool in = true;
            if (m & TRC_BIN_ANGLE) in = in && ang >= g[0] && ang <= g[1];
            if (m & TRC_BIN_HEIGHT) in = in && hh >= g[2] && hh <= g[3];
            if (m & TRC_BIN_RADIUS) in = in && rr >= g[4] && ((m & TRC_BIN_RADIUS_HALF_OPEN) ? rr < g[5] : rr <= g[5]);
            if (in) atomicAdd(&l_sum[j], e);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_bins; i += blockDim.x)
        if (l_sum[i] != 0.0) atomicAdd(&out[i], l_sum[i]);
}

extern "C" int trc_scene_bin_hits(trc_scene *sc, int32_t n_bins, const int32_t *surf_lo, const int32_t *surf_hi,
                                  const double *ranges6, const int32_t *mode, double *out) {
    if (!sc || n_bins < 0 || (n_bins > 0 && (!surf_lo || !surf_hi || !ranges6 || !mode || !out))) return trc_fail(TRC_ERR_INVALID, "bad arguments");
    if (n_bins == 0) return TRC_OK;
    trc_ctx *ctx = sc->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    unsigned long long c[2];
    HIP_TRY(hipMemcpy(c, sc->d_counters, sizeof(c), hipMemcpyDeviceToHost));
    long long reserved = (long long)c[0];
    if (reserved > sc->hit_cap) reserved = sc->hit_cap;
    for (int i = 0; i < n_bins; ++i) out[i] = 0.0;
    if (reserved == 0) return TRC_OK;
    const size_t per_bin = 2 * sizeof(int32_t) + 6 * sizeof(double) + sizeof(int32_t) + sizeof(double);
    char *d_buf = nullptr;
    HIP_TRY(hipMalloc((void **)&d_buf, (size_t)BIN_TILE * per_bin));
    double *d_rng = (double *)d_buf, *d_out = d_rng + 6 * BIN_TILE;
    int32_t *d_lo = (int32_t *)(d_out + BIN_TILE), *d_hi = d_lo + BIN_TILE, *d_mode = d_hi + BIN_TILE;
    int st = TRC_OK;
    unsigned grid = (unsigned)((reserved + 255) / 256);
    if (grid > (unsigned)(ctx->n_cu * 8)) grid = (unsigned)(ctx->n_cu * 8);
    for (int b0 = 0; b0 < n_bins && st == TRC_OK; b0 += BIN_TILE) {
        const int nb = n_bins - b0 < BIN_TILE ? n_bins - b0 : BIN_TILE;
        hipError_t e = hipMemcpyAsync(d_rng, ranges6 + 6 * (size_t)b0, (size_t)nb * 48, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_lo, surf_lo + b0, (size_t)nb * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_hi, surf_hi + b0, (size_t)nb * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_mode, mode + b0, (size_t)nb * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_out, 0, (size_t)nb * 8, ctx->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_bin_hits, dim3(grid), dim3(256), 0, ctx->stream, reserved, sc->d_h_surf, sc->d_h[0], sc->d_h[2], sc->d_h[3],
                               sc->d_h[4], nb, d_lo, d_hi, d_rng, d_mode, d_out);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(out + b0, d_out, (size_t)nb * 8, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) st = trc_fail(TRC_ERR_DEVICE, "trc_scene_bin_hits: %s", hipGetErrorString(e));
    }
    (void)hipFree(d_buf);
    return st;
}

#define TRC_TRANSFER_MAX_SURF 1024      /* (S+1) x S doubles, kept in every private copy of the tally buffer too */
extern "C" int trc_scene_enable_transfer(trc_scene *sc, int32_t on) {
    if (!sc) return trc_fail(TRC_ERR_INVALID, "scene is NULL");
    if (on && sc->n_surf > TRC_TRANSFER_MAX_SURF)
        return trc_fail(TRC_ERR_CAPACITY, "the transfer matrix is offered up to %d surfaces (scene has %d)", TRC_TRANSFER_MAX_SURF, sc->n_surf);
    HIP_TRY(hipSetDevice(sc->ctx->device));
    HIP_TRY(hipStreamSynchronize(sc->ctx->stream));
    if ((on != 0) == sc->transfer_on) return TRC_OK;
    sc->transfer_on = on != 0;
    TRC_TRY(scene_alloc_tally(sc));      // resets the tallies; flux-map bins keep their offsets (the matrix comes last)
    if (!sc->fms_h.empty()) HIP_TRY(hipMemcpy(sc->d_fms, sc->fms_h.data(), sc->fms_h.size() * sizeof(FluxMapDev), hipMemcpyHostToDevice));
    return TRC_OK;
}

extern "C" int trc_scene_get_transfer(trc_scene *sc, double *out) {
    if (!sc || !out) return trc_fail(TRC_ERR_INVALID, "bad arguments");
    if (!sc->transfer_on) return trc_fail(TRC_ERR_INVALID, "the transfer matrix is not enabled (trc_scene_enable_transfer)");
    HIP_TRY(hipSetDevice(sc->ctx->device));
    HIP_TRY(hipStreamSynchronize(sc->ctx->stream));
    HIP_TRY(hipMemcpy(out, sc->d_tally + sc->tr_off, (size_t)(sc->n_surf + 1) * sc->n_surf * sizeof(double), hipMemcpyDeviceToHost));
    return TRC_OK;
}

extern "C" int trc_scene_tally_size(trc_scene *sc, int64_t *n_doubles) {
    if (!sc || !n_doubles) return trc_fail(TRC_ERR_INVALID, "bad arguments");
    *n_doubles = sc->tally_n;
    return TRC_OK;
}

extern "C" int trc_scene_export_tallies(trc_scene *sc, double *dst, int32_t on_device) {
    if (!sc || !dst) return trc_fail(TRC_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(sc->ctx->device));
    HIP_TRY(hipStreamSynchronize(sc->ctx->stream));
    HIP_TRY(hipMemcpy(dst, sc->d_tally, (size_t)sc->tally_n * sizeof(double),
                      on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost));
    return TRC_OK;
}

extern "C" int trc_scene_import_tallies(trc_scene *sc, const double *src, int32_t on_device) {
    if (!sc || !src) return trc_fail(TRC_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(sc->ctx->device));
    HIP_TRY(hipStreamSynchronize(sc->ctx->stream));
    HIP_TRY(hipMemcpy(sc->d_tally, src, (size_t)sc->tally_n * sizeof(double),
                      on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    return TRC_OK;
}

static DScene make_dscene(trc_scene *sc) {
    DScene d;
    memset(&d, 0, sizeof(d));
    d.recs = sc->d_recs; d.opt = sc->d_opt; d.sflags = sc->d_sflags; d.extra = sc->d_extra;
    d.stride = sc->stride; d.n_surf = sc->n_surf; d.n_extra = sc->n_extra; d.has_kd = sc->has_kd ? 1 : 0;
    d.kd_a = sc->d_kd_a; d.kd_b = sc->d_kd_b; d.kd_leaf = sc->d_kd_leaf; d.kd_always = sc->d_kd_always;
    d.kd_split = sc->d_kd_split;
    d.kd_nodes = sc->kd_nodes; d.kd_nleaf = sc->kd_nleaf; d.kd_nalways = sc->kd_nalways;
    for (int i = 0; i < 3; ++i) { d.kd_bmin[i] = sc->kd_bounds[i]; d.kd_bmax[i] = sc->kd_bounds[3 + i]; }
    d.a_sbox = sc->d_a_sbox; d.a_obb = sc->d_a_obb; d.a_nodes = sc->d_a_nodes; d.a_leaf = sc->d_a_leaf; d.a_unbounded = sc->d_a_unbounded;
    d.a_n_unbounded = (int32_t)sc->accel.unbounded.size(); d.a_kd_depth = sc->accel.kd_depth;
    d.a_bleaf = sc->d_a_bleaf; d.a_n_bleaf = (int32_t)sc->accel.brute_leaf.size();
    d.a_bnodes[0] = sc->accel.brute_nodes[0]; d.a_bnodes[1] = sc->accel.brute_nodes[1];
    for (int i = 0; i < 6; ++i) d.a_broot[i] = sc->accel.brute_root[i];
    d.a_ok = sc->accel_ok ? 1 : 0; d.a_kd_ok = sc->accel_kd_ok ? 1 : 0;
    for (int i = 0; i < 6; ++i) d.a_root[i] = sc->accel.root[i];
    d.a_delta = sc->accel.delta;
    d.a_goff = sc->d_a_goff; d.a_glist = sc->d_a_glist; d.a_gapart = sc->d_a_gapart;
    d.a_bg_off = sc->d_a_bg_off; d.a_bg_occ = sc->d_a_bg_occ; d.a_bg_ent = sc->d_a_bg_ent; d.a_bg_ok = sc->accel.big_ok ? 1 : 0;
    d.a_bg_apart = sc->d_a_bg_apart; d.a_bg_napart = sc->accel.big_ok ? (int32_t)sc->accel.big_apart.size() : 0;
    for (int i = 0; i < 3; ++i) {
        d.a_bg_dim[i] = sc->accel.big_ok ? sc->accel.big_dim[i] : 1;
        d.a_bg_lo[i] = sc->accel.big_lo[i]; d.a_bg_cs[i] = sc->accel.big_cs[i]; d.a_bg_inv[i] = sc->accel.big_inv[i];
    }
    for (int i = 0; i < 6; ++i) d.a_bg_root[i] = sc->accel.big_ok ? sc->accel.big_root[i] : 0.0f;
    d.a_g_napart = sc->accel.grid_ok ? (int32_t)sc->accel.grid_apart.size() : 0;
    for (int i = 0; i < 6; ++i) d.a_groot[i] = sc->accel.grid_ok ? sc->accel.grid_root[i] : 0.0f;
    d.a_g_ok = sc->accel.grid_ok ? 1 : 0;
    d.a_g_ncell = sc->accel.grid_ok ? (int32_t)sc->accel.grid_off.size() - 1 : 0;
    d.a_g_nlist = sc->accel.grid_ok ? (int32_t)sc->accel.grid_list.size() : 0;
    for (int i = 0; i < 3; ++i) {
        d.a_gdim[i] = sc->accel.grid_ok ? sc->accel.grid_dim[i] : 1;
        d.a_glo[i] = sc->accel.grid_lo[i]; d.a_gcs[i] = sc->accel.grid_cs[i]; d.a_ginv[i] = sc->accel.grid_inv[i];
    }
    for (int i = 0; i < 3; ++i) { d.a_cen[i] = sc->accel.cen[i]; d.a_slo[i] = sc->accel.slo[i]; d.a_shi[i] = sc->accel.shi[i]; }
    d.tally = sc->d_tally;
    d.fm_of_surf = sc->d_fm_of_surf; d.fms = sc->d_fms; d.fm_edges = sc->d_fm_edges;
    d.tr_off = sc->tr_off;
    d.n_fm = (int32_t)sc->fms_h.size(); d.n_fm_edges = (int32_t)sc->fm_edges_h.size();
    d.counters = sc->d_counters; d.energy_left = sc->d_energy_left;
    d.hit_cap = sc->hit_cap; d.h_surf = sc->d_h_surf;
    d.h_eabs = sc->d_h[0]; d.h_ein = sc->d_h[1]; d.h_px = sc->d_h[2]; d.h_py = sc->d_h[3]; d.h_pz = sc->d_h[4];
    d.h_dx = sc->d_h[5]; d.h_dy = sc->d_h[6]; d.h_dz = sc->d_h[7];
    return d;
}

// ================================================================================================
// ray staging helpers
// ================================================================================================
struct DevRays {
    double *x = nullptr, *y = nullptr, *z = nullptr, *dx = nullptr, *dy = nullptr, *dz = nullptr, *e = nullptr;
    double *ref = nullptr, *wl = nullptr;
    uint64_t *rid = nullptr;
    bool owned[10] = {false, false, false, false, false, false, false, false, false, false};
    void release() {
        double **p[9] = {&x, &y, &z, &dx, &dy, &dz, &e, &ref, &wl};
        for (int i = 0; i < 9; ++i) { if (owned[i] && *p[i]) pool_free(*p[i]); *p[i] = nullptr; }
        if (owned[9] && rid) pool_free(rid);
        rid = nullptr;
    }
};

static int check_rays(const trc_rays *r, int64_t n, const char *who) {
    if (!r) return trc_fail(TRC_ERR_INVALID, "%s: rays is NULL", who);
    if (r->n < n) return trc_fail(TRC_ERR_INVALID, "%s: bundle holds %lld rays, %lld requested", who, (long long)r->n, (long long)n);
    if (n > 0 && (!r->x || !r->y || !r->z || !r->dx || !r->dy || !r->dz))
        return trc_fail(TRC_ERR_INVALID, "%s: vertices and directions are required", who);
    return TRC_OK;
}

// bring the required columns of a bundle to the device (no copy when already there)
static int stage_rays(const trc_rays *r, int64_t n, bool need_energy, DevRays *d) {
    const double *src[9] = {r->x, r->y, r->z, r->dx, r->dy, r->dz, r->e, r->ref_index, r->wavelength};
    double **dst[9] = {&d->x, &d->y, &d->z, &d->dx, &d->dy, &d->dz, &d->e, &d->ref, &d->wl};
    if (need_energy && !r->e) return trc_fail(TRC_ERR_INVALID, "ray energies are required");
    for (int i = 0; i < 9; ++i) {
        if (!src[i]) continue;
        if (r->on_device) { *dst[i] = (double *)src[i]; continue; }
        TRC_TRY(dev_alloc(dst[i], (size_t)n));
        d->owned[i] = true;
        HIP_TRY(hipMemcpy(*dst[i], src[i], (size_t)n * 8, hipMemcpyHostToDevice));
    }
    if (r->rid) {
        if (r->on_device) d->rid = r->rid;
        else {
            TRC_TRY(dev_alloc(&d->rid, (size_t)n));
            d->owned[9] = true;
            HIP_TRY(hipMemcpy(d->rid, r->rid, (size_t)n * 8, hipMemcpyHostToDevice));
        }
    }
    return TRC_OK;
}

static int upload_source(const trc_source_desc *src, trc_source_desc **d_src) {
    if (src->kind < TRC_SRC_PILLBOX_DISK || src->kind > TRC_SRC_VF_FRUSTUM)
        return trc_fail(TRC_ERR_UNSUPPORTED, "source kind %d is not in the native table", src->kind);
    TRC_TRY(dev_alloc(d_src, 1));
    HIP_TRY(hipMemcpy(*d_src, src, sizeof(trc_source_desc), hipMemcpyHostToDevice));
    return TRC_OK;
}

// ================================================================================================
// C-ABI: fast engine
// ================================================================================================
extern "C" int trc_trace_fast(trc_scene *sc, const trc_rays *in, const trc_source_desc *src, int64_t n, int32_t reps,
                              double min_energy, uint64_t seed, uint64_t ray_offset, int32_t flags, trc_rays *last,
                              trc_trace_stats *stats) {
    if (!sc) return trc_fail(TRC_ERR_INVALID, "scene is NULL");
    if ((in == nullptr) == (src == nullptr)) return trc_fail(TRC_ERR_INVALID, "exactly one of `in` and `src` must be given");
    if (n < 0 || reps < 0) return trc_fail(TRC_ERR_INVALID, "n and reps must be >= 0");
    if (sc->splits) return trc_fail(TRC_ERR_UNSUPPORTED, "the scene has ray-splitting optics: use trc_trace_ordered");
    // Rays that carry the imaginary part of a complex index, materials evaluated at their wavelength or a sampled spectrum are
    // traced by the streaming form (k_s_shade_x); the megakernel knows nothing of them.
    const bool carry = sc->carries || (in && (in->ref_index_im || in->spectra || in->mat));
    const int carry_W = (in && in->spectra && in->spec_wl) ? in->n_spec : 0;
    const int carry_mat = (in && in->mat) ? (int)in->n_mat : 0;
    if (carry) {
        if (!in) return trc_fail(TRC_ERR_UNSUPPORTED, "the scene has optics that read what only the rays of a given bundle carry (materials, spectra)");
        if (carry_W < 0 || carry_W > 4096 || carry_mat < 0 || carry_mat > 64) return trc_fail(TRC_ERR_INVALID, "trc_trace_fast: n_spec or n_mat out of range");
        int max_mat = -1;
        bool poly = false;
        for (const trc_surface_desc &sd : sc->surfs) {
            if (sd.optics_kind == TRC_OPT_REFRACTIVE_MATERIAL) max_mat = std::max(max_mat, std::max((int)sd.opt[4], (int)sd.opt[5]));
            if (sd.optics_kind == TRC_OPT_LAMBERTIAN_POLYCHROMATIC) poly = true;
        }
        if (max_mat >= 0 && (!in->wavelength || carry_mat <= max_mat))
            return trc_fail(TRC_ERR_INVALID, "trc_trace_fast: surfaces between materials need the rays' wavelengths and the materials evaluated at them (trc_rays.mat)");
        if (poly && carry_W < 2) return trc_fail(TRC_ERR_INVALID, "trc_trace_fast: a polychromatic wall needs rays with spectra of at least two samples");
        if ((flags & TRC_TRACE_MEGAKERNEL) || n < 64)
            return trc_fail(TRC_ERR_UNSUPPORTED, "complex refractive indices and spectra travel with the streaming form of the fast engine (64 rays or more) and with trc_trace_ordered");
    }
    // TRC_TRACE_ACCEL without a Kd-tree: the streaming form searches its own grid; the megakernel tests every box
    trc_ctx *ctx = sc->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    DevRays dr;
    trc_source_desc *d_src = nullptr;
    double *d_last[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int st = TRC_OK;
    trc_trace_stats s;
    memset(&s, 0, sizeof(s));
    double tally_before[2] = {0, 0};
    unsigned long long cnt_before[4] = {0, 0, 0, 0};
    double eleft_before = 0;
    double stream_seg = 0, stream_hits = 0;
    bool stream_counts_known = false;
    const int S = sc->n_surf;
    double *carry_d[4] = {nullptr, nullptr, nullptr, nullptr};      // Im of the index, materials, sample wavelengths, spectra
    bool carry_owned[4] = {false, false, false, false};
    do {
        if (in) {
            if ((st = check_rays(in, n, "trc_trace_fast")) || (st = stage_rays(in, n, true, &dr))) break;
            if (carry) {        // the carried columns: rows of the bundle's own length in->n apart on the host, n apart here
                struct { const double *src; int rows; double **dst; } blk[4] = {{in->ref_index_im, 1, &carry_d[0]}, {in->mat, 2 * carry_mat, &carry_d[1]},
                                                                                 {carry_W ? in->spec_wl : nullptr, carry_W, &carry_d[2]},
                                                                                 {carry_W ? in->spectra : nullptr, carry_W, &carry_d[3]}};
                for (int b = 0; b < 4 && st == TRC_OK; ++b) {
                    if (!blk[b].src || blk[b].rows <= 0) continue;
                    if (in->on_device && in->n == n) { *blk[b].dst = (double *)blk[b].src; continue; }
                    if ((st = dev_alloc(blk[b].dst, (size_t)blk[b].rows * (size_t)n))) break;
                    carry_owned[b] = true;
                    if (hipMemcpy2D(*blk[b].dst, (size_t)n * 8, blk[b].src, (size_t)in->n * 8, (size_t)n * 8, (size_t)blk[b].rows,
                                    in->on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice) != hipSuccess)
                        st = trc_fail(TRC_ERR_DEVICE, "upload of the carried columns failed");
                }
                if (st) break;
            }
        }
        else {
            // the scene keeps a device buffer for the descriptor of the call in progress (hipMalloc / hipFree per call
            // cost more than the upload)
            if (src->kind < TRC_SRC_PILLBOX_DISK || src->kind > TRC_SRC_VF_FRUSTUM) { st = trc_fail(TRC_ERR_UNSUPPORTED, "source kind %d is not in the native table", src->kind); break; }
            if (!sc->d_src_buf) { if ((st = dev_alloc(&sc->d_src_buf, 1))) break; sc->src_host_ok = false; }
            if (!(sc->src_host_ok && memcmp(&sc->src_host, src, sizeof(trc_source_desc)) == 0)) {
                sc->src_host_ok = false;
                if (hipMemcpy(sc->d_src_buf, src, sizeof(trc_source_desc), hipMemcpyHostToDevice) != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "source upload failed"); break; }
                memcpy(&sc->src_host, src, sizeof(trc_source_desc));
                sc->src_host_ok = true;
            }
            d_src = sc->d_src_buf;
        }
        int64_t last_cap = 0;
        if (flags & TRC_TRACE_KEEP_LAST) {
            if (!last || !last->x || !last->y || !last->z || !last->dx || !last->dy || !last->dz || !last->e || last->on_device) {
                st = trc_fail(TRC_ERR_INVALID, "TRC_TRACE_KEEP_LAST needs a host `last` bundle with x..e"); break;
            }
            last_cap = last->n;
            if (sc->d_last_cap < last_cap) {
                for (int i = 0; i < 7; ++i) dev_free(sc->d_last[i]);
                sc->d_last_cap = 0;
                for (int i = 0; i < 7 && st == TRC_OK; ++i) st = dev_alloc(&sc->d_last[i], (size_t)last_cap);
                if (st) { for (int i = 0; i < 7; ++i) dev_free(sc->d_last[i]); break; }
                sc->d_last_cap = last_cap;
            }
            for (int i = 0; i < 7; ++i) d_last[i] = sc->d_last[i];
        }
        // counters and the energy left live in one 64-byte block: one read before, one after
        unsigned long long blk_before[8];
        if (sc->cnt_host_ok) memcpy(blk_before, sc->cnt_host, sizeof(blk_before));
        else if (hipMemcpy(blk_before, sc->d_counters, sizeof(blk_before), hipMemcpyDeviceToHost) != hipSuccess) {
            st = trc_fail(TRC_ERR_DEVICE, "counter readback failed"); break;
        }
        sc->cnt_host_ok = false;          // (until this call has read them back at its end)
        const int64_t dirty_before = sc->hit_dirty_to;
        sc->hit_dirty_to = sc->hit_cap;   // (... and then says how far the hit buffer was used)
        for (int i = 0; i < 4; ++i) cnt_before[i] = blk_before[i];
        memcpy(&eleft_before, &blk_before[5], sizeof(double));
        // the `last` cursor restarts for every call
        if (blk_before[2] != 0ull) {
            unsigned long long zero = 0;
            if (hipMemcpy(sc->d_counters + 2, &zero, sizeof(zero), hipMemcpyHostToDevice) != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "memcpy failed"); break; }
        }

        FastParams P;
        memset(&P, 0, sizeof(P));
        P.sc = make_dscene(sc);
        P.x = dr.x; P.y = dr.y; P.z = dr.z; P.dx = dr.dx; P.dy = dr.dy; P.dz = dr.dz; P.e = dr.e;
        P.ref = dr.ref; P.wl = dr.wl; P.rid = dr.rid;
        // polychromatic hits: the captured ones keep their sample wavelengths and their spectrum before and after (3 W columns beside
        // the hit buffer, made when the first call with spectra finds a buffer to capture into)
        double *hit_x = nullptr;
        bool captures = false;
        if (sc->hit_cap > 0)
            for (int i = 0; i < S && !captures; ++i) captures = (sc->surfs[i].flags & TRC_SURF_CAPTURE_HITS) != 0;
        if (carry_W > 0 && captures) {
            const int cols = 3 * carry_W;
            if (sc->d_hx && (sc->hx_cols != cols || sc->hx_cap != sc->hit_cap)) {
                if (cnt_before[0] != 0ull) { st = trc_fail(TRC_ERR_INVALID, "the hit buffer holds hits with spectra of another sample count: read or clear them first"); break; }
                dev_free(sc->d_hx); sc->hx_cols = 0; sc->hx_cap = 0;
            }
            if (!sc->d_hx) {
                if ((st = dev_alloc(&sc->d_hx, (size_t)cols * (size_t)sc->hit_cap))) break;
                sc->hx_cols = cols; sc->hx_cap = sc->hit_cap;
            }
            hit_x = sc->d_hx;
        }
        CarryIn carry_in;
        carry_in.hit_x = hit_x; carry_in.hit_x_cap = sc->hx_cap;
        carry_in.ref_im = carry_d[0]; carry_in.mat = carry_d[1]; carry_in.spec_wl = carry_d[2]; carry_in.spec = carry_d[3]; carry_in.n_mat = carry_mat; carry_in.n_spec = carry_W;
        P.src = d_src;
        P.n = n; P.reps = reps; P.flags = flags; P.min_energy = min_energy; P.seed = seed; P.ray_offset = ray_offset;
        P.lx = d_last[0]; P.ly = d_last[1]; P.lz = d_last[2]; P.ldx = d_last[3]; P.ldy = d_last[4]; P.ldz = d_last[5]; P.le = d_last[6];
        P.last_cap = last_cap;
        P.capture = 0;
        if (sc->hit_cap > 0)
            for (int i = 0; i < S; ++i) if (sc->surfs[i].flags & TRC_SURF_CAPTURE_HITS) P.capture = 1;

        const bool accel = sc->has_kd && (flags & TRC_TRACE_ACCEL);
        size_t b_buie = src ? (size_t)TRC_BUIE_STAGED * 8 : 0;
        size_t b_tally = (size_t)(3 * S + 2) * 8;
        static int threads_env = -1, mode_env = -1;
        if (threads_env < 0) { const char *ev = getenv("TRC_FAST_THREADS"); threads_env = ev ? atoi(ev) : 0; }
        if (mode_env < 0) { const char *ev = getenv("TRC_FAST_GENERIC"); mode_env = (ev && atoi(ev)) ? 1 : 0; }
        // preferred: single-precision conservative search with everything it needs in LDS (up to 160 KiB per CU)
        bool m32 = sc->accel_ok && !mode_env && S <= 65535 &&
                   (!accel || (sc->accel_kd_ok && sc->kd_nodes <= COOP_MAX_NODES && sc->accel.kd_depth <= COOP_MAX_DEPTH));
        int threads = 512;
        if (threads_env == 256 || threads_env == 512 || threads_env == 768 || threads_env == 1024) threads = threads_env;
        size_t lds = 0;
        if (m32) {
            size_t b_acc = (size_t)6 * S * 4 + (accel ? ((size_t)2 * sc->kd_nodes * 4 + (size_t)sc->kd_nalways * 4 + (size_t)sc->kd_nleaf * 2)
                                                      : (8 + sc->accel.brute_leaf.size() * 2)) +
                           sc->accel.unbounded.size() * 4 + 32;
            for (;;) {
                lds = b_buie + b_tally + b_acc + (size_t)(threads / 64) * COOP_WAVE_BYTES(accel ? (sc->accel.kd_depth > 0 ? sc->accel.kd_depth : 1) : 1);
                if (lds <= 160 * 1024 - 512 || threads == 256) break;
                threads = threads == 1024 ? 768 : (threads == 768 ? 512 : 256);
            }
            if (lds > 160 * 1024 - 512) m32 = false;
            P.lds_tally = 1;
            P.lds_scene = 0;
        }
        if (!m32) {
            threads = 256;
            size_t b_scene = (size_t)S * sc->stride * 8;
            if (sc->has_kd) b_scene += (size_t)sc->kd_nodes * 8 + ((size_t)sc->kd_nodes * 2 + sc->kd_nleaf + sc->kd_nalways) * 4 + 8;
            const size_t LDS_MAX = 64 * 1024;
            lds = b_buie;
            P.lds_tally = (lds + b_tally <= LDS_MAX) ? 1 : 0;
            if (P.lds_tally) lds += b_tally;
            P.lds_scene = (lds + b_scene <= LDS_MAX) ? 1 : 0;
            if (P.lds_scene) lds += b_scene;
        }
        // Large calls run the streaming engine (phases as separate kernels connected by HBM queues, trc_stream.inc);
        // small ones the persistent megakernel, which needs one launch and no workspace.  TRC_TRACE_STREAM /
        // TRC_TRACE_MEGAKERNEL (or TRC_FAST_STREAM=1 / 0) force one or the other.
        static int stream_env = -2;
        if (stream_env == -2) { const char *ev = getenv("TRC_FAST_STREAM"); stream_env = ev ? (atoi(ev) ? 1 : 0) : -1; }
        const bool want_accel = (flags & TRC_TRACE_ACCEL) != 0;
        int stream_mode = 0;
        size_t stream_lds = 0;
        const bool stream_ok = !mode_env && n >= 64 && stream_plan(sc, want_accel, &stream_mode, &stream_lds);
        const bool force_stream = (flags & TRC_TRACE_STREAM) || stream_env == 1 || carry;
        const bool force_mega = !carry && ((flags & TRC_TRACE_MEGAKERNEL) || stream_env == 0);
        // (a scene on the large grid -- a mesh of 1e5 faces -- has nothing but its boxes to search in the megakernel: 2e5 rays on the
        // relief of 105 800 triangles took 570 ms there, 1.3 ms here)
        const long long stream_from = stream_mode == 3 ? 4096 : TRC_STREAM_MIN_RAYS;
        const bool use_stream = stream_ok && (force_stream || (!force_mega && n >= stream_from));
        if (carry && !use_stream) { st = trc_fail(TRC_ERR_UNSUPPORTED, "no streaming form for this scene: rays that carry complex indices or spectra go through trc_trace_ordered"); break; }
        if (use_stream) {
            if (!sc->stream_eng) {
                sc->stream_eng = new (std::nothrow) StreamEngine();
                if (!sc->stream_eng) { st = trc_fail(TRC_ERR_NOMEM, "out of host memory"); break; }
                memset(sc->stream_eng, 0, sizeof(StreamEngine));
            }
            if ((st = stream_trace(sc, P, carry_in, want_accel, src, *sc->stream_eng, &s, &stream_seg, &stream_hits))) {
                // the hits captured by the bounces that completed: wind the buffer back to where the call found it
                const std::string why = g_last_error;
                unsigned long long now = 0;
                (void)hipDeviceSynchronize();
                if (sc->hit_cap > 0 && hipMemcpy(&now, sc->d_counters, sizeof(now), hipMemcpyDeviceToHost) == hipSuccess && now > cnt_before[0]) {
                    const unsigned long long end = now < (unsigned long long)sc->hit_cap ? now : (unsigned long long)sc->hit_cap;
                    if (end > cnt_before[0]) (void)hipMemset(sc->d_h_surf + cnt_before[0], 0xFF, (size_t)(end - cnt_before[0]) * sizeof(int32_t));
                    (void)hipMemcpy(sc->d_counters, &cnt_before[0], sizeof(unsigned long long), hipMemcpyHostToDevice);
                }
                sc->hit_epoch += 1;
                g_last_error = why;
                break;
            }
            stream_counts_known = true;
        } else {
        (void)hipStreamSynchronize(ctx->stream);       // (the sums of an earlier streaming call may still be on their way into the buffer)
        if (hipMemcpy(tally_before, sc->d_tally + 3 * S, sizeof(tally_before), hipMemcpyDeviceToHost) != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "counter readback failed"); break; }
        void (*kern)(FastParams) = nullptr;
        if (m32) kern = threads == 1024 ? k_trace_coop<1024> : (threads == 768 ? k_trace_coop<768> : (threads == 512 ? k_trace_coop<512> : k_trace_coop<256>));
        else kern = k_trace_fast<256>;
        if (lds > 64 * 1024) {
            hipError_t ae = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (ae != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(ae)); break; }
        }

        // persistent grid: as many workgroups as are resident at once, never more waves than rays/64
        int blocks_per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, (const void *)kern, threads, lds) != hipSuccess || blocks_per_cu < 1)
            blocks_per_cu = 1;
        if (blocks_per_cu > 8) blocks_per_cu = 8;
        long long grid = (long long)ctx->n_cu * blocks_per_cu;
        long long max_grid = (n + threads - 1) / threads;
        if (grid > max_grid) grid = max_grid;
        if (grid < 1) grid = 1;
        if (n > 0) {
            if (hipEventRecord(ctx->ev0, ctx->stream) != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "event record failed"); break; }
            hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(threads), lds, ctx->stream, P);
            hipError_t le = hipGetLastError();
            if (le != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "k_trace_fast launch failed: %s", hipGetErrorString(le)); break; }
            if (hipEventRecord(ctx->ev1, ctx->stream) != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "event record failed"); break; }
            hipError_t se = hipStreamSynchronize(ctx->stream);
            if (se != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "k_trace_fast failed: %s", hipGetErrorString(se)); break; }
            float ms = 0;
            (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
            s.kernel_ms = ms;
            s.launches = 1;
        }
        }
        unsigned long long blk_after[8], cnt_after[4];
        double eleft_after;
        if (hipMemcpy(blk_after, sc->d_counters, sizeof(blk_after), hipMemcpyDeviceToHost) != hipSuccess) {
            st = trc_fail(TRC_ERR_DEVICE, "counter readback failed"); break;
        }
        for (int i = 0; i < 4; ++i) cnt_after[i] = blk_after[i];
        memcpy(&eleft_after, &blk_after[5], sizeof(double));
        memcpy(sc->cnt_host, blk_after, sizeof(blk_after));
        sc->cnt_host_ok = true;
        {       // every entry written lies below the cursor: chunks are reserved by advancing it
            const int64_t cur = (int64_t)(blk_after[0] < (unsigned long long)sc->hit_cap ? blk_after[0] : (unsigned long long)sc->hit_cap);
            sc->hit_dirty_to = cur > dirty_before ? cur : dirty_before;
        }
        if (stream_counts_known) {              // the streaming form counted on the host
            s.segments = (int64_t)(stream_seg + 0.5);
            s.hits = (int64_t)(stream_hits + 0.5);
        } else {
            double tally_after[2];
            if (hipMemcpy(tally_after, sc->d_tally + 3 * S, sizeof(tally_after), hipMemcpyDeviceToHost) != hipSuccess) {
                st = trc_fail(TRC_ERR_DEVICE, "counter readback failed"); break;
            }
            s.segments = (int64_t)(tally_after[0] - tally_before[0] + 0.5);
            s.hits = (int64_t)(tally_after[1] - tally_before[1] + 0.5);
        }
        s.rays_left = (int64_t)(cnt_after[3] - cnt_before[3]);
        s.hits_dropped = (int64_t)(cnt_after[1] - cnt_before[1]);
        s.energy_left = eleft_after - eleft_before;
        s.bounces = reps;
        if (flags & TRC_TRACE_KEEP_LAST) {
            int64_t m = (int64_t)cnt_after[2];
            if (m > last_cap) { st = trc_fail(TRC_ERR_CAPACITY, "%lld rays left but `last` holds %lld", (long long)m, (long long)last_cap); break; }
            double *dst[7] = {last->x, last->y, last->z, last->dx, last->dy, last->dz, last->e};
            for (int i = 0; i < 7 && m > 0; ++i)
                if (hipMemcpy(dst[i], d_last[i], (size_t)m * 8, hipMemcpyDeviceToHost) != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "memcpy failed"); break; }
            last->n = m;
        }
    } while (0);
    dr.release();
    for (int b = 0; b < 4; ++b) if (carry_owned[b]) pool_free(carry_d[b]);
    if (stats) *stats = s;
    return st;
}

// ================================================================================================
// C-ABI: source generation
// ================================================================================================
extern "C" int trc_source_start32(trc_ctx *ctx, const trc_source_desc *src, int64_t n, uint64_t seed, uint64_t ray_offset,
                                  float *lx, float *ly, double *eps) {
    if (!ctx || !src || n < 0 || !lx || !ly) return trc_fail(TRC_ERR_INVALID, "trc_source_start32: bad arguments");
    HIP_TRY(hipSetDevice(ctx->device));
    trc_fp_params F;
    double half = 0.0, theta_c = 0.0;
    const char *why = "";
    if (!trc_fp_source(*src, F, &half, &theta_c, &why)) return trc_fail(TRC_ERR_UNSUPPORTED, "no footprint map for this source: %s", why);
    if (eps) *eps = TRC_FP_EPS_REL * half;
    if (n == 0) return TRC_OK;
    float *d[2] = {nullptr, nullptr};
    int st = TRC_OK;
    do {
        if ((st = dev_alloc(&d[0], (size_t)n)) || (st = dev_alloc(&d[1], (size_t)n))) break;
        long long grid = (n + 255) / 256;
        if (grid > 8192) grid = 8192;
        hipLaunchKernelGGL(k_source_start32, dim3((unsigned)grid), dim3(256), 0, ctx->stream, F, (long long)n, (unsigned long long)seed,
                           (unsigned long long)ray_offset, d[0], d[1]);
        hipError_t se = hipStreamSynchronize(ctx->stream);
        if (se != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "k_source_start32 failed: %s", hipGetErrorString(se)); break; }
        if (hipMemcpy(lx, d[0], (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(ly, d[1], (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess)
            st = trc_fail(TRC_ERR_DEVICE, "memcpy failed");
    } while (0);
    dev_free(d[0]); dev_free(d[1]);
    return st;
}

extern "C" int trc_source_generate(trc_ctx *ctx, const trc_source_desc *src, int64_t n, uint64_t seed,
                                   uint64_t ray_offset, trc_rays *out) {
    if (!ctx || !src || n < 0) return trc_fail(TRC_ERR_INVALID, "trc_source_generate: bad arguments");
    TRC_TRY(check_rays(out, n, "trc_source_generate"));
    if (!out->e) return trc_fail(TRC_ERR_INVALID, "trc_source_generate: energy column required");
    HIP_TRY(hipSetDevice(ctx->device));
    if (n == 0) return TRC_OK;
    trc_source_desc *d_src = nullptr;
    double *d[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    uint64_t *d_rid = nullptr;
    int st = TRC_OK;
    do {
        if ((st = upload_source(src, &d_src))) break;
        double *host[7] = {out->x, out->y, out->z, out->dx, out->dy, out->dz, out->e};
        if (out->on_device) { for (int i = 0; i < 7; ++i) d[i] = host[i]; d_rid = out->rid; }
        else {
            for (int i = 0; i < 7 && st == TRC_OK; ++i) st = dev_alloc(&d[i], (size_t)n);
            if (st == TRC_OK && out->rid) st = dev_alloc(&d_rid, (size_t)n);
            if (st) break;
        }
        long long grid = (n + 255) / 256;
        if (grid > 8192) grid = 8192;
        hipLaunchKernelGGL(k_source_generate, dim3((unsigned)grid), dim3(256), 0, ctx->stream, d_src, (long long)n,
                           (unsigned long long)seed, (unsigned long long)ray_offset, d[0], d[1], d[2], d[3], d[4], d[5],
                           d[6], d_rid);
        hipError_t se = hipStreamSynchronize(ctx->stream);
        if (se != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "k_source_generate failed: %s", hipGetErrorString(se)); break; }
        if (!out->on_device) {
            for (int i = 0; i < 7; ++i)
                if (hipMemcpy(host[i], d[i], (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "memcpy failed"); break; }
            if (st == TRC_OK && out->rid && hipMemcpy(out->rid, d_rid, (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess)
                st = trc_fail(TRC_ERR_DEVICE, "memcpy failed");
        }
    } while (0);
    dev_free(d_src);
    if (!out->on_device) { for (int i = 0; i < 7; ++i) dev_free(d[i]); dev_free(d_rid); }
    return st;
}

// ================================================================================================
// C-ABI: ordered engine
// ================================================================================================
static void level_free(Level &L) {
    dev_free(L.slab);
    L.x = L.y = L.z = L.dx = L.dy = L.dz = L.e = L.ref = L.wl = L.pay = nullptr;
    L.rid = nullptr; L.parent = nullptr; L.surf = nullptr;
}

// One allocation per level (a trace of five levels used to cost 65 hipMalloc / hipFree pairs, 2 ms of a 1e5-ray call): eleven
// 8-byte columns and the surface column, each starting on a 256-byte boundary, then the carried rows (n apart, as the kernels index them).
static int level_alloc(Level &L, int64_t n, int n_pay) {
    memset(&L, 0, sizeof(L));
    L.n_total = n; L.n_live = n;
    const size_t m = ((size_t)(n > 0 ? n : 1) + 31) & ~(size_t)31;
    const size_t bytes = m * (11 * 8 + 4) + (n_pay > 0 ? (size_t)n * n_pay * 8 : 0);
    TRC_TRY(dev_alloc(&L.slab, bytes));
    double **p[9] = {&L.x, &L.y, &L.z, &L.dx, &L.dy, &L.dz, &L.e, &L.ref, &L.wl};
    for (int i = 0; i < 9; ++i) *p[i] = (double *)(L.slab + (size_t)i * m * 8);
    L.rid = (uint64_t *)(L.slab + 9 * m * 8);
    L.parent = (int64_t *)(L.slab + 10 * m * 8);
    L.surf = (int32_t *)(L.slab + 11 * m * 8);
    if (n_pay > 0) L.pay = (double *)(L.slab + 11 * m * 8 + m * 4);
    return TRC_OK;
}

// Scratch of the bounce loop: allocated for the first (usually the largest) bounce and kept; a bounce that needs more -- refractive
// surfaces can double a level -- gets a new set.
#define ORD_SCRATCH_KEEP ((size_t)1 << 26)
struct OrdScratch {
    double *o[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    uint64_t *orid = nullptr;
    double *opay = nullptr;
    uint32_t *key = nullptr, *ckey = nullptr, *cslot = nullptr, *skey = nullptr, *sslot = nullptr;
    unsigned *blk_cnt = nullptr, *blk_cul = nullptr;
    unsigned long long *blk_off = nullptr, *totals = nullptr;
    char *sort_tmp = nullptr;
    size_t cap_slots = 0, sort_bytes = 0;
    int cap_pay = 0;
    void release() {
        for (int i = 0; i < 9; ++i) dev_free(o[i]);
        dev_free(opay);
        dev_free(orid); dev_free(key); dev_free(ckey); dev_free(cslot); dev_free(skey); dev_free(sslot);
        dev_free(blk_cnt); dev_free(blk_cul); dev_free(blk_off); dev_free(totals);
        dev_free(sort_tmp);
        cap_slots = 0; sort_bytes = 0; cap_pay = 0;
    }
    int ensure(size_t slots, int n_pay) {
        if (slots <= cap_slots && n_pay <= cap_pay) return TRC_OK;
        if (slots < cap_slots) slots = cap_slots;
        release();
        cap_pay = n_pay;
        for (int i = 0; i < 9; ++i) TRC_TRY(dev_alloc(&o[i], slots));
        TRC_TRY(dev_alloc(&orid, slots));
        if (n_pay > 0) TRC_TRY(dev_alloc(&opay, slots * (size_t)n_pay));
        TRC_TRY(dev_alloc(&key, slots));
        TRC_TRY(dev_alloc(&ckey, slots)); TRC_TRY(dev_alloc(&cslot, slots));      // the occupied slots: at most all of them
        TRC_TRY(dev_alloc(&skey, slots)); TRC_TRY(dev_alloc(&sslot, slots));
        const size_t nblk = (slots + 255) / 256;
        TRC_TRY(dev_alloc(&blk_cnt, nblk)); TRC_TRY(dev_alloc(&blk_cul, nblk)); TRC_TRY(dev_alloc(&blk_off, nblk));
        TRC_TRY(dev_alloc(&totals, 2));
        cap_slots = slots;
        return TRC_OK;
    }
    int ensure_sort(size_t bytes) {
        if (bytes <= sort_bytes && sort_tmp) return TRC_OK;
        dev_free(sort_tmp);
        sort_bytes = 0;
        TRC_TRY(dev_alloc(&sort_tmp, bytes));
        sort_bytes = bytes;
        return TRC_OK;
    }
};

static void scene_free_ord_scratch(trc_scene *sc) {
    if (sc->ord_scratch) { sc->ord_scratch->release(); delete sc->ord_scratch; sc->ord_scratch = nullptr; }
}

extern "C" int trc_result_destroy(trc_result *res) {
    if (!res) return TRC_OK;
    (void)hipSetDevice(res->ctx->device);
    for (auto &L : res->levels) level_free(L);
    delete res;
    return TRC_OK;
}

extern "C" int trc_trace_ordered(trc_scene *sc, const trc_rays *in, const trc_source_desc *src, int64_t n, int32_t reps,
                                 double min_energy, uint64_t seed, uint64_t ray_offset, int32_t flags, trc_result **out,
                                 trc_trace_stats *stats) {
    if (!sc || !out) return trc_fail(TRC_ERR_INVALID, "trc_trace_ordered: bad arguments");
    *out = nullptr;
    if ((in == nullptr) == (src == nullptr)) return trc_fail(TRC_ERR_INVALID, "exactly one of `in` and `src` must be given");
    if (n < 0 || reps < 0) return trc_fail(TRC_ERR_INVALID, "n and reps must be >= 0");
    if (2 * n >= (int64_t)0xFFFFFFFFll) return trc_fail(TRC_ERR_UNSUPPORTED, "ordered engine handles fewer than 2^31 rays per call");
    if ((flags & TRC_TRACE_ACCEL) && !sc->has_kd) return trc_fail(TRC_ERR_INVALID, "TRC_TRACE_ACCEL without a Kd-tree on the scene");
    if (sc->n_surf >= (1 << 28)) return trc_fail(TRC_ERR_UNSUPPORTED, "too many surfaces for the ordering key");
    trc_ctx *ctx = sc->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    // what the rays carry beyond the nine columns (complex indices, material rows, spectra)
    PayLayout lay;
    {
        int max_mat = -1;
        bool poly = false;
        for (const auto &sd : sc->surfs) {
            if (sd.optics_kind == TRC_OPT_REFRACTIVE_MATERIAL) max_mat = std::max(max_mat, std::max((int)sd.opt[4], (int)sd.opt[5]));
            if (sd.optics_kind == TRC_OPT_LAMBERTIAN_POLYCHROMATIC) poly = true;
        }
        if (in) {
            lay.has_im = (in->ref_index_im != nullptr || max_mat >= 0) ? 1 : 0;
            lay.n_mat = in->mat ? (int)in->n_mat : 0;
            lay.W = (in->spectra && in->spec_wl) ? in->n_spec : 0;
        }
        if (max_mat >= 0 && (!in || !in->wavelength || lay.n_mat <= max_mat))
            return trc_fail(TRC_ERR_INVALID, "the scene refracts between tabulated materials: the bundle needs wavelengths and the %d materials' indices at them (trc_rays.mat)", max_mat + 1);
        if (poly && lay.W < 2)
            return trc_fail(TRC_ERR_INVALID, "the scene has polychromatic optics: the bundle needs spectra (trc_rays.spectra, spec_wl)");
        if (lay.W < 0 || lay.W > 4096 || lay.n_mat < 0 || lay.n_mat > 64) return trc_fail(TRC_ERR_INVALID, "trc_trace_ordered: n_spec or n_mat out of range");
    }
    const int n_pay = lay.rows();
    trc_result *res = new (std::nothrow) trc_result();
    if (!res) return trc_fail(TRC_ERR_NOMEM, "out of host memory");
    res->ctx = ctx;
    res->lay = lay;
    trc_trace_stats s;
    memset(&s, 0, sizeof(s));
    // the scratch of the bounce loop stays with the scene (an ordered trace of 1e7 rays allocated and freed 3 GB in twenty blocks
    // beyond the pool's sizes per call: 30 of its 45 ms)
    if (!sc->ord_scratch) { sc->ord_scratch = new (std::nothrow) OrdScratch(); if (!sc->ord_scratch) return trc_fail(TRC_ERR_NOMEM, "out of host memory"); }
    OrdScratch &sx = *sc->ord_scratch;
    trc_source_desc *d_src = nullptr;
    int st = TRC_OK;
    float total_ms = 0;
    do {
        // ---- level 0: the source bundle ----
        Level L0;
        if ((st = level_alloc(L0, n, n_pay))) { level_free(L0); break; }
        res->levels.push_back(L0);
        Level &B0 = res->levels.back();
        long long g0 = (n + 255) / 256; if (g0 > 8192) g0 = 8192; if (g0 < 1) g0 = 1;
        if (src) {
            if ((st = upload_source(src, &d_src))) break;
            hipLaunchKernelGGL(k_source_generate, dim3((unsigned)g0), dim3(256), 0, ctx->stream, d_src, (long long)n,
                               (unsigned long long)seed, (unsigned long long)ray_offset, B0.x, B0.y, B0.z, B0.dx, B0.dy,
                               B0.dz, B0.e, B0.rid);
            hipLaunchKernelGGL(k_fill_f64, dim3((unsigned)g0), dim3(256), 0, ctx->stream, B0.ref, (long long)n, 1.0);
            hipLaunchKernelGGL(k_fill_f64, dim3((unsigned)g0), dim3(256), 0, ctx->stream, B0.wl, (long long)n, 0.0);
        } else {
            if ((st = check_rays(in, n, "trc_trace_ordered"))) break;
            if (!in->e) { st = trc_fail(TRC_ERR_INVALID, "ray energies are required"); break; }
            hipMemcpyKind kind = in->on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
            const double *hs[7] = {in->x, in->y, in->z, in->dx, in->dy, in->dz, in->e};
            double *ds[7] = {B0.x, B0.y, B0.z, B0.dx, B0.dy, B0.dz, B0.e};
            bool bad = false;
            for (int i = 0; i < 7 && n > 0; ++i) if (hipMemcpy(ds[i], hs[i], (size_t)n * 8, kind) != hipSuccess) bad = true;
            if (in->ref_index) { if (n > 0 && hipMemcpy(B0.ref, in->ref_index, (size_t)n * 8, kind) != hipSuccess) bad = true; }
            else hipLaunchKernelGGL(k_fill_f64, dim3((unsigned)g0), dim3(256), 0, ctx->stream, B0.ref, (long long)n, 1.0);
            if (in->wavelength) { if (n > 0 && hipMemcpy(B0.wl, in->wavelength, (size_t)n * 8, kind) != hipSuccess) bad = true; }
            else hipLaunchKernelGGL(k_fill_f64, dim3((unsigned)g0), dim3(256), 0, ctx->stream, B0.wl, (long long)n, 0.0);
            if (in->rid) { if (n > 0 && hipMemcpy(B0.rid, in->rid, (size_t)n * 8, kind) != hipSuccess) bad = true; }
            else hipLaunchKernelGGL(k_fill_rid, dim3((unsigned)g0), dim3(256), 0, ctx->stream, B0.rid, (long long)n, (unsigned long long)ray_offset);
            if (n_pay > 0 && n > 0) {       // rows of the caller's 2-D columns are in->n apart, ours n
                if (lay.has_im) {
                    if (in->ref_index_im) { if (hipMemcpy(B0.pay, in->ref_index_im, (size_t)n * 8, kind) != hipSuccess) bad = true; }
                    else if (hipMemset(B0.pay, 0, (size_t)n * 8) != hipSuccess) bad = true;
                }
                struct { const double *src; int rows, r0; } blk[3] = {{in->mat, 2 * lay.n_mat, lay.r_mat()}, {in->spec_wl, lay.W, lay.r_wl()},
                                                                      {in->spectra, lay.W, lay.r_spec()}};
                for (auto &b : blk)
                    if (b.rows > 0 && hipMemcpy2D(B0.pay + (size_t)b.r0 * n, (size_t)n * 8, b.src, (size_t)in->n * 8, (size_t)n * 8, (size_t)b.rows, kind) != hipSuccess)
                        bad = true;
            }
            if (bad) { st = trc_fail(TRC_ERR_DEVICE, "bundle upload failed"); break; }
        }
        if (n > 0) {
            (void)hipMemsetAsync(B0.parent, 0, (size_t)n * 8, ctx->stream);
            (void)hipMemsetAsync(B0.surf, 0xFF, (size_t)n * 4, ctx->stream);
        }
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "level 0 setup failed"); break; }

        // ---- bounce loop ----
        int64_t n_cur = n;
        for (int it = 0; it < reps && n_cur > 0; ++it) {
            const Level cur = res->levels.back();
            const int64_t slots = 2 * n_cur;
            if ((st = sx.ensure((size_t)slots, n_pay))) break;
            const long long nblk = (slots + 255) / 256;

            OrdParams P;
            memset(&P, 0, sizeof(P));
            P.sc = make_dscene(sc);
            P.x = cur.x; P.y = cur.y; P.z = cur.z; P.dx = cur.dx; P.dy = cur.dy; P.dz = cur.dz; P.e = cur.e;
            P.ref = cur.ref; P.wl = cur.wl; P.rid = cur.rid;
            P.n = n_cur; P.event = it + 1; P.flags = flags; P.min_energy = min_energy; P.seed = seed;
            P.ox = sx.o[0]; P.oy = sx.o[1]; P.oz = sx.o[2]; P.odx = sx.o[3]; P.ody = sx.o[4]; P.odz = sx.o[5];
            P.oe = sx.o[6]; P.oref = sx.o[7]; P.owl = sx.o[8]; P.orid = sx.orid; P.key = sx.key;
            P.pay = cur.pay; P.pay_stride = cur.n_total; P.opay = sx.opay; P.lay = lay;

            (void)hipEventRecord(ctx->ev0, ctx->stream);
            {
                const bool use_kd = sc->has_kd && (flags & TRC_TRACE_ACCEL);
                const bool fast = sc->accel_ok && sc->n_surf <= 65535 && (!use_kd || (sc->accel_kd_ok && sc->accel.kd_depth + 2 <= ORD_STACK_DEPTH));
                const bool big = sc->accel_ok && sc->accel.big_ok && !use_kd;      // the scene stands on the large grid (and the caller brought no tree)
                if (big) hipLaunchKernelGGL(k_ord_bounce<2>, dim3((unsigned)((n_cur + 255) / 256)), dim3(256), 0, ctx->stream, P);
                else if (fast) hipLaunchKernelGGL(k_ord_bounce<1>, dim3((unsigned)((n_cur + 255) / 256)), dim3(256), 0, ctx->stream, P);
                else hipLaunchKernelGGL(k_ord_bounce<0>, dim3((unsigned)((n_cur + 255) / 256)), dim3(256), 0, ctx->stream, P);
            }
            hipLaunchKernelGGL(k_compact_count, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, sx.key, (long long)slots,
                               sx.blk_cnt, sx.blk_cul);
            hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(256), 0, ctx->stream, sx.blk_cnt, sx.blk_cul, nblk, sx.blk_off,
                               sx.totals);
            (void)hipEventRecord(ctx->ev1, ctx->stream);
            unsigned long long totals[2];
            hipError_t ce = hipMemcpyAsync(totals, sx.totals, sizeof(totals), hipMemcpyDeviceToHost, ctx->stream);
            hipError_t se = hipStreamSynchronize(ctx->stream);
            if (ce != hipSuccess || se != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "bounce %d failed: %s", it, hipGetErrorString(se != hipSuccess ? se : ce)); break; }
            float ms = 0; (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1); total_ms += ms;
            s.launches += 3;
            s.segments += n_cur;
            s.bounces = it + 1;
            const int64_t m = (int64_t)totals[0], n_culled = (int64_t)totals[1];
            // hits = rays that produced at least one child: count child-0 slots == slots < n_cur occupied; cheap bound: m minus second children
            if (m == 0) { n_cur = 0; break; }   // "Ray bundle depleted": nothing recorded (tracer_engine.py:271, :277)
            hipLaunchKernelGGL(k_compact_scatter, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, sx.key, (long long)slots,
                               sx.blk_off, sx.ckey, sx.cslot);
            // stable sort by (culled, surface, block); slot order (= parent order) is kept inside a key
            size_t tmp_bytes = 0;
            hipError_t re = rocprim::radix_sort_pairs(nullptr, tmp_bytes, sx.ckey, sx.skey, sx.cslot, sx.sslot, (size_t)m, 0, 31, ctx->stream);
            if (re != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "radix_sort_pairs(size query) failed: %s", hipGetErrorString(re)); break; }
            if ((st = sx.ensure_sort(tmp_bytes))) break;
            re = rocprim::radix_sort_pairs(sx.sort_tmp, tmp_bytes, sx.ckey, sx.skey, sx.cslot, sx.sslot, (size_t)m, 0, 31, ctx->stream);
            if (re != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "radix_sort_pairs failed: %s", hipGetErrorString(re)); break; }
            Level Ln;
            if ((st = level_alloc(Ln, m, n_pay))) { level_free(Ln); break; }
            Ln.n_live = m - n_culled;
            res->levels.push_back(Ln);
            GatherParams G;
            G.ox = sx.o[0]; G.oy = sx.o[1]; G.oz = sx.o[2]; G.odx = sx.o[3]; G.ody = sx.o[4]; G.odz = sx.o[5];
            G.oe = sx.o[6]; G.oref = sx.o[7]; G.owl = sx.o[8]; G.orid = sx.orid; G.skey = sx.skey; G.sslot = sx.sslot;
            G.m = m; G.n_parent = n_cur;
            G.x = Ln.x; G.y = Ln.y; G.z = Ln.z; G.dx = Ln.dx; G.dy = Ln.dy; G.dz = Ln.dz; G.e = Ln.e; G.ref = Ln.ref;
            G.wl = Ln.wl; G.rid = Ln.rid; G.parent = Ln.parent; G.surf = Ln.surf;
            G.recs = sc->d_recs; G.stride = sc->stride;
            G.opay = sx.opay; G.pay = Ln.pay; G.n_pay = n_pay;
            hipLaunchKernelGGL(k_ord_gather, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream, G);
            se = hipStreamSynchronize(ctx->stream);
            if (se != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "ordering of bounce %d failed: %s", it, hipGetErrorString(se)); break; }
            s.launches += 3;
            s.hits += m;
            n_cur = Ln.n_live;
        }
        if (st) break;
        s.rays_left = n_cur;
        s.kernel_ms = total_ms;
        if (n_cur > 0) {
            // energy of the live part of the last level
            const Level &LL = res->levels.back();
            std::vector<double> eh((size_t)n_cur);
            if (hipMemcpy(eh.data(), LL.e, (size_t)n_cur * 8, hipMemcpyDeviceToHost) == hipSuccess)
                for (double v : eh) s.energy_left += v;
        }
    } while (0);
    if (sx.cap_slots > ORD_SCRATCH_KEEP) sx.release();      // (beyond 2^26 slots -- 10 GB -- the scratch goes back after the call)
    dev_free(d_src);
    if (stats) *stats = s;
    if (st != TRC_OK) { trc_result_destroy(res); return st; }
    *out = res;
    return TRC_OK;
}

extern "C" int trc_result_num_levels(trc_result *res, int32_t *n_levels) {
    if (!res || !n_levels) return trc_fail(TRC_ERR_INVALID, "bad arguments");
    *n_levels = (int32_t)res->levels.size();
    return TRC_OK;
}

extern "C" int trc_result_level_size(trc_result *res, int32_t level, int64_t *n_total, int64_t *n_live) {
    if (!res || level < 0 || level >= (int)res->levels.size()) return trc_fail(TRC_ERR_INVALID, "level out of range");
    if (n_total) *n_total = res->levels[level].n_total;
    if (n_live) *n_live = res->levels[level].n_live;
    return TRC_OK;
}

extern "C" int trc_result_level_get(trc_result *res, int32_t level, trc_rays *out, int32_t *surf) {
    if (!res || level < 0 || level >= (int)res->levels.size() || !out) return trc_fail(TRC_ERR_INVALID, "bad arguments");
    const Level &L = res->levels[level];
    if (out->n < L.n_total) return trc_fail(TRC_ERR_CAPACITY, "output bundle holds %lld rays, level has %lld", (long long)out->n, (long long)L.n_total);
    if (out->on_device) return trc_fail(TRC_ERR_INVALID, "host output expected");
    HIP_TRY(hipSetDevice(res->ctx->device));
    const size_t n = (size_t)L.n_total;
    const int64_t cap = out->n;
    out->n = L.n_total;
    if (n == 0) return TRC_OK;
    struct { void *dst; const void *src; size_t w; } cp[] = {
        {out->x, L.x, 8}, {out->y, L.y, 8}, {out->z, L.z, 8}, {out->dx, L.dx, 8}, {out->dy, L.dy, 8}, {out->dz, L.dz, 8},
        {out->e, L.e, 8}, {out->parent, L.parent, 8}, {out->ref_index, L.ref, 8}, {out->wavelength, L.wl, 8},
        {out->rid, L.rid, 8}, {surf, L.surf, 4}};
    for (auto &c : cp)
        if (c.dst) HIP_TRY(hipMemcpy(c.dst, c.src, n * c.w, hipMemcpyDeviceToHost));
    // what the rays carry beyond that (rows of the caller's 2-D columns: the capacity it handed over apart)
    const PayLayout &lay = res->lay;
    if (out->ref_index_im) {
        if (lay.has_im) HIP_TRY(hipMemcpy(out->ref_index_im, L.pay, n * 8, hipMemcpyDeviceToHost));
        else memset(out->ref_index_im, 0, n * 8);
    }
    if (out->spectra || out->spec_wl) {
        if (out->n_spec != lay.W) return trc_fail(TRC_ERR_INVALID, "the level's rays carry %d spectral samples, the output has room for %d", lay.W, out->n_spec);
        if (out->spec_wl && lay.W) HIP_TRY(hipMemcpy2D(out->spec_wl, (size_t)cap * 8, L.pay + (size_t)lay.r_wl() * n, n * 8, n * 8, (size_t)lay.W, hipMemcpyDeviceToHost));
        if (out->spectra && lay.W) HIP_TRY(hipMemcpy2D(out->spectra, (size_t)cap * 8, L.pay + (size_t)lay.r_spec() * n, n * 8, n * 8, (size_t)lay.W, hipMemcpyDeviceToHost));
    }
    return TRC_OK;
}

// ================================================================================================
// C-ABI: KdTree.traversal on its own (accel_tree.py:213-312)
// ================================================================================================
struct KdTravParams {
    const int32_t *flag, *child, *leaf_off, *leaf_cnt, *leaf_surfs;
    const double *split;
    double lo[3], hi[3];
    const double *x, *y, *z, *dx, *dy, *dz;
    long long n;
    uint8_t *rel;       // n_surf rows of n bytes
    int *flags;         // [0]: some ray meets the root box; [1]: a ray needed more than KD_TRAV_STACK pending nodes
};

#define KD_TRAV_STACK 64

// numpy's maximum / minimum hand a nan on (intersect_bounds :325-326)
__device__ inline double np_maximum(double a, double b) { return (a != a || b != b) ? NAN : (a > b ? a : b); }
__device__ inline double np_minimum(double a, double b) { return (a != a || b != b) ? NAN : (a < b ? a : b); }

__global__ __launch_bounds__(256) void k_kd_traversal(KdTravParams P) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= P.n) return;
    const double pos[3] = {P.x[r], P.y[r], P.z[r]}, dir[3] = {P.dx[r], P.dy[r], P.dz[r]};
    const double inv[3] = {1.0 / dir[0], 1.0 / dir[1], 1.0 / dir[2]};     // :224
    double t_lo = 0.0, t_hi = INFINITY;                                     // intersect_bounds :314-330
    for (int i = 0; i < 3; ++i) {
        const bool neg = dir[i] < 0.0;
        double a = ((neg ? P.hi[i] : P.lo[i]) - pos[i]) * inv[i];
        double b = ((neg ? P.lo[i] : P.hi[i]) - pos[i]) * inv[i];
        if (a > b) { const double t = a; a = b; b = t; }
        t_lo = np_maximum(t_lo, a);
        t_hi = np_minimum(t_hi, b);
    }
    if (!(t_hi > 0.0) || (t_lo > t_hi)) return;
    P.flags[0] = 1;
    int node = 0, top = 0;
    int st_node[KD_TRAV_STACK];
    double st_min[KD_TRAV_STACK], st_max[KD_TRAV_STACK];
    double t_min = t_lo, t_max = t_hi;
    while (true) {
        if (t_hi < t_min) break;                                            // :243 (the root's exit against the current entry)
        const int f = P.flag[node];
        if (f != 3) {
            const double sp = P.split[node];
            const double t_plane = (sp - pos[f]) * inv[f];                  // :249
            int c1 = P.child[node], c2 = c1 + 1;
            const bool below_first = (pos[f] < sp) || (pos[f] == sp && dir[f] <= 0.0);
            if (!below_first) { const int t = c1; c1 = c2; c2 = t; }
            if (t_plane > t_max || t_plane <= 0.0) node = c1;               // :258-259
            else if (t_plane < t_min) node = c2;
            else {
                if (top >= KD_TRAV_STACK) { P.flags[1] = 1; return; }
                st_node[top] = c2; st_min[top] = t_plane; st_max[top] = t_max;
                ++top;
                node = c1;
                t_max = t_plane;
            }
        } else {
            const int off = P.leaf_off[node], cnt = P.leaf_cnt[node];
            for (int k = 0; k < cnt; ++k) P.rel[(size_t)P.leaf_surfs[off + k] * P.n + r] = 1;       // :288
            if (top > 0) { --top; node = st_node[top]; t_min = st_min[top]; t_max = st_max[top]; }
            else break;
        }
    }
}

extern "C" int trc_kdtree_traversal(trc_ctx *ctx, const trc_kdtree_desc *kd, int32_t n_surf, const trc_rays *rays, int64_t n,
                                    uint8_t *relevancy, int32_t *any_inter) {
    if (!ctx || !kd || !relevancy || n_surf <= 0 || n < 0) return trc_fail(TRC_ERR_INVALID, "trc_kdtree_traversal: bad arguments");
    TRC_TRY(check_rays(rays, n, "trc_kdtree_traversal"));
    if (rays->on_device) return trc_fail(TRC_ERR_INVALID, "trc_kdtree_traversal: host bundle expected");
    if (kd->n_nodes <= 0 || !kd->flag || !kd->split || !kd->child || !kd->leaf_off || !kd->leaf_cnt || (kd->n_leaf_surfs > 0 && !kd->leaf_surfs))
        return trc_fail(TRC_ERR_INVALID, "trc_kdtree_traversal: incomplete tree");
    for (int i = 0; i < kd->n_nodes; ++i) {
        if (kd->flag[i] < 0 || kd->flag[i] > 3) return trc_fail(TRC_ERR_INVALID, "node %d: flag %d", i, kd->flag[i]);
        if (kd->flag[i] != 3 && (kd->child[i] <= i || kd->child[i] + 1 >= kd->n_nodes)) return trc_fail(TRC_ERR_INVALID, "node %d: children out of range", i);
        if (kd->flag[i] == 3 && (kd->leaf_off[i] < 0 || kd->leaf_cnt[i] < 0 || kd->leaf_off[i] + kd->leaf_cnt[i] > kd->n_leaf_surfs))
            return trc_fail(TRC_ERR_INVALID, "node %d: leaf list out of range", i);
    }
    for (int i = 0; i < kd->n_leaf_surfs; ++i)
        if (kd->leaf_surfs[i] < 0 || kd->leaf_surfs[i] >= n_surf) return trc_fail(TRC_ERR_INVALID, "leaf surface %d out of range", kd->leaf_surfs[i]);
    for (int i = 0; i < kd->n_always; ++i)
        if (kd->always_relevant[i] < 0 || kd->always_relevant[i] >= n_surf) return trc_fail(TRC_ERR_INVALID, "always-relevant surface out of range");
    HIP_TRY(hipSetDevice(ctx->device));
    int32_t *d_i32[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    double *d_split = nullptr, *d_r[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    uint8_t *d_rel = nullptr;
    int *d_flags = nullptr;
    int st = TRC_OK;
    int any = 0;
    do {
        const int32_t *hs[5] = {kd->flag, kd->child, kd->leaf_off, kd->leaf_cnt, kd->leaf_surfs};
        const size_t cnt[5] = {(size_t)kd->n_nodes, (size_t)kd->n_nodes, (size_t)kd->n_nodes, (size_t)kd->n_nodes, (size_t)std::max(kd->n_leaf_surfs, 1)};
        for (int i = 0; i < 5 && st == TRC_OK; ++i) {
            if ((st = dev_alloc(&d_i32[i], cnt[i]))) break;
            if (hs[i] && (i < 4 || kd->n_leaf_surfs > 0) && hipMemcpy(d_i32[i], hs[i], cnt[i] * 4, hipMemcpyHostToDevice) != hipSuccess) st = trc_fail(TRC_ERR_DEVICE, "memcpy failed");
        }
        if (st) break;
        if ((st = dev_alloc(&d_split, (size_t)kd->n_nodes))) break;
        if (hipMemcpy(d_split, kd->split, (size_t)kd->n_nodes * 8, hipMemcpyHostToDevice) != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "memcpy failed"); break; }
        if ((st = dev_alloc(&d_flags, 2))) break;
        (void)hipMemset(d_flags, 0, 8);
        const size_t nn = (size_t)std::max<int64_t>(n, 1);
        if ((st = dev_alloc(&d_rel, (size_t)n_surf * nn))) break;
        (void)hipMemset(d_rel, 0, (size_t)n_surf * nn);
        for (int i = 0; i < kd->n_always; ++i) (void)hipMemset(d_rel + (size_t)kd->always_relevant[i] * nn, 1, nn);        // :236
        if (n > 0) {
            const double *src[6] = {rays->x, rays->y, rays->z, rays->dx, rays->dy, rays->dz};
            for (int i = 0; i < 6 && st == TRC_OK; ++i) {
                if ((st = dev_alloc(&d_r[i], (size_t)n))) break;
                if (hipMemcpy(d_r[i], src[i], (size_t)n * 8, hipMemcpyHostToDevice) != hipSuccess) st = trc_fail(TRC_ERR_DEVICE, "memcpy failed");
            }
            if (st) break;
            KdTravParams P;
            P.flag = d_i32[0]; P.child = d_i32[1]; P.leaf_off = d_i32[2]; P.leaf_cnt = d_i32[3]; P.leaf_surfs = d_i32[4]; P.split = d_split;
            for (int i = 0; i < 3; ++i) { P.lo[i] = kd->bounds[i]; P.hi[i] = kd->bounds[3 + i]; }
            P.x = d_r[0]; P.y = d_r[1]; P.z = d_r[2]; P.dx = d_r[3]; P.dy = d_r[4]; P.dz = d_r[5];
            P.n = n; P.rel = d_rel; P.flags = d_flags;
            hipLaunchKernelGGL(k_kd_traversal, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, P);
            hipError_t se = hipStreamSynchronize(ctx->stream);
            if (se != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "k_kd_traversal failed: %s", hipGetErrorString(se)); break; }
            int fl[2] = {0, 0};
            if (hipMemcpy(fl, d_flags, 8, hipMemcpyDeviceToHost) != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "memcpy failed"); break; }
            if (fl[1]) { st = trc_fail(TRC_ERR_CAPACITY, "a ray had more than %d pending nodes: the tree is deeper than the traversal's stack", KD_TRAV_STACK); break; }
            any = fl[0];
            if (hipMemcpy(relevancy, d_rel, (size_t)n_surf * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "memcpy failed"); break; }
        }
        // `if inters.any() or self.always_relevant.any()` (:238): any() of the array of surface indices -- a non-zero index
        for (int i = 0; i < kd->n_always; ++i) if (kd->always_relevant[i] != 0) any = 1;
    } while (0);
    for (int i = 0; i < 5; ++i) dev_free(d_i32[i]);
    for (int i = 0; i < 6; ++i) dev_free(d_r[i]);
    dev_free(d_split); dev_free(d_rel); dev_free(d_flags);
    if (st == TRC_OK && any_inter) *any_inter = any;
    return st;
}

// ================================================================================================
// C-ABI: per-surface protocol
// ================================================================================================
static int upload_record(const trc_surface_desc *surf, int32_t n_extra, const double *extra, double **d_rec, double **d_opt,
                         double **d_extra) {
    TRC_TRY(validate_surface(*surf, 0, n_extra));
    int stride = TRC_REC_HDR + 16;
    std::vector<double> rec(stride);
    pack_record(*surf, rec.data(), stride);
    TRC_TRY(dev_alloc(d_rec, (size_t)stride));
    HIP_TRY(hipMemcpy(*d_rec, rec.data(), stride * sizeof(double), hipMemcpyHostToDevice));
    if (d_opt) {
        TRC_TRY(dev_alloc(d_opt, 8));
        HIP_TRY(hipMemcpy(*d_opt, surf->opt, 8 * sizeof(double), hipMemcpyHostToDevice));
    }
    TRC_TRY(dev_alloc(d_extra, (size_t)(n_extra > 0 ? n_extra : 1)));
    if (n_extra > 0 && extra) HIP_TRY(hipMemcpy(*d_extra, extra, (size_t)n_extra * sizeof(double), hipMemcpyHostToDevice));
    return TRC_OK;
}

extern "C" int trc_gm_find_intersections(trc_ctx *ctx, const trc_surface_desc *surf, int32_t n_extra, const double *extra,
                                         const trc_rays *rays, double *t_out, double *hx, double *hy, double *hz) {
    if (!ctx || !surf || !rays || !t_out) return trc_fail(TRC_ERR_INVALID, "trc_gm_find_intersections: bad arguments");
    if (rays->on_device) return trc_fail(TRC_ERR_INVALID, "host bundle expected");
    const int64_t n = rays->n;
    TRC_TRY(check_rays(rays, n, "trc_gm_find_intersections"));
    HIP_TRY(hipSetDevice(ctx->device));
    if (n == 0) return TRC_OK;
    double *d_rec = nullptr, *d_extra = nullptr, *d_t = nullptr, *d_h[3] = {nullptr, nullptr, nullptr};
    DevRays dr;
    int st = TRC_OK;
    do {
        if ((st = upload_record(surf, n_extra, extra, &d_rec, nullptr, &d_extra))) break;
        if ((st = stage_rays(rays, n, false, &dr))) break;
        if ((st = dev_alloc(&d_t, (size_t)n))) break;
        const bool want_h = hx && hy && hz;
        if (want_h) for (int i = 0; i < 3 && st == TRC_OK; ++i) st = dev_alloc(&d_h[i], (size_t)n);
        if (st) break;
        hipLaunchKernelGGL(k_gm_intersect, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_rec, d_extra,
                           (long long)n, dr.x, dr.y, dr.z, dr.dx, dr.dy, dr.dz, d_t, d_h[0], d_h[1], d_h[2]);
        hipError_t se = hipStreamSynchronize(ctx->stream);
        if (se != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "k_gm_intersect failed: %s", hipGetErrorString(se)); break; }
        if (hipMemcpy(t_out, d_t, (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "memcpy failed"); break; }
        if (want_h) {
            double *dst[3] = {hx, hy, hz};
            for (int i = 0; i < 3; ++i)
                if (hipMemcpy(dst[i], d_h[i], (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "memcpy failed"); break; }
        }
    } while (0);
    dr.release();
    dev_free(d_rec); dev_free(d_extra); dev_free(d_t);
    for (int i = 0; i < 3; ++i) dev_free(d_h[i]);
    return st;
}

extern "C" int trc_gm_get_normals(trc_ctx *ctx, const trc_surface_desc *surf, int64_t n, const double *hx, const double *hy,
                                  const double *hz, const double *dx, const double *dy, const double *dz, double *nx,
                                  double *ny, double *nz) {
    if (!ctx || !surf || n < 0) return trc_fail(TRC_ERR_INVALID, "trc_gm_get_normals: bad arguments");
    if (n == 0) return TRC_OK;
    if (!hx || !hy || !hz || !dx || !dy || !dz || !nx || !ny || !nz) return trc_fail(TRC_ERR_INVALID, "trc_gm_get_normals: NULL array");
    HIP_TRY(hipSetDevice(ctx->device));
    double *d_rec = nullptr, *d_extra = nullptr, *d[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const double *src[6] = {hx, hy, hz, dx, dy, dz};
    int st = TRC_OK;
    do {
        trc_surface_desc tmp = *surf;
        tmp.optics_kind = TRC_OPT_TRANSPARENT;   // normals do not depend on the optics
        if (tmp.gm_kind == TRC_GM_RECT_PERFORATED) { tmp.gm_kind = TRC_GM_RECT; }
        if ((st = upload_record(&tmp, 0, nullptr, &d_rec, nullptr, &d_extra))) break;
        for (int i = 0; i < 9 && st == TRC_OK; ++i) st = dev_alloc(&d[i], (size_t)n);
        if (st) break;
        for (int i = 0; i < 6; ++i)
            if (hipMemcpy(d[i], src[i], (size_t)n * 8, hipMemcpyHostToDevice) != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "memcpy failed"); break; }
        if (st) break;
        hipLaunchKernelGGL(k_gm_normals, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_rec, (long long)n, d[0],
                           d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8]);
        hipError_t se = hipStreamSynchronize(ctx->stream);
        if (se != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "k_gm_normals failed: %s", hipGetErrorString(se)); break; }
        double *dst[3] = {nx, ny, nz};
        for (int i = 0; i < 3; ++i)
            if (hipMemcpy(dst[i], d[6 + i], (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "memcpy failed"); break; }
    } while (0);
    dev_free(d_rec); dev_free(d_extra);
    for (int i = 0; i < 9; ++i) dev_free(d[i]);
    return st;
}

__global__ __launch_bounds__(256) void k_fresnel_attenuating(long long n, double n1, const double *m_re, const double *m_im,
                                                             const double *th, double *rp, double *rs, double *t2) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) trc_fresnel_attenuating(th[i], n1, m_re[i], m_im[i], &rp[i], &rs[i], &t2[i]);
}

extern "C" int trc_optics_fresnel_attenuating(trc_ctx *ctx, int64_t n, double n1, const double *m_re, const double *m_im,
                                              const double *theta1, double *r_p, double *r_s, double *theta2) {
    if (!ctx || n < 0 || (n > 0 && (!m_re || !m_im || !theta1 || !r_p || !r_s || !theta2)))
        return trc_fail(TRC_ERR_INVALID, "trc_optics_fresnel_attenuating: bad arguments");
    if (n == 0) return TRC_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    double *d[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const double *src[3] = {m_re, m_im, theta1};
    double *dst[3] = {r_p, r_s, theta2};
    int st = TRC_OK;
    do {
        for (int i = 0; i < 6 && st == TRC_OK; ++i) st = dev_alloc(&d[i], (size_t)n);
        if (st) break;
        for (int i = 0; i < 3; ++i)
            if (hipMemcpy(d[i], src[i], (size_t)n * 8, hipMemcpyHostToDevice) != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "memcpy failed"); break; }
        if (st) break;
        hipLaunchKernelGGL(k_fresnel_attenuating, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (long long)n, n1,
                           d[0], d[1], d[2], d[3], d[4], d[5]);
        hipError_t se = hipStreamSynchronize(ctx->stream);
        if (se != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "k_fresnel_attenuating failed: %s", hipGetErrorString(se)); break; }
        for (int i = 0; i < 3; ++i)
            if (hipMemcpy(dst[i], d[3 + i], (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "memcpy failed"); break; }
    } while (0);
    for (int i = 0; i < 6; ++i) dev_free(d[i]);
    return st;
}

extern "C" int trc_optics_apply(trc_ctx *ctx, const trc_surface_desc *surf, int32_t n_extra, const double *extra,
                                const trc_rays *in, const double *hx, const double *hy, const double *hz, const double *nx,
                                const double *ny, const double *nz, uint64_t seed, int32_t bounce, trc_rays *out) {
    if (!ctx || !surf || !in || !out) return trc_fail(TRC_ERR_INVALID, "trc_optics_apply: bad arguments");
    if (in->on_device || out->on_device) return trc_fail(TRC_ERR_INVALID, "host bundles expected");
    const int64_t n = in->n;
    if (out->n < 2 * n) return trc_fail(TRC_ERR_CAPACITY, "output bundle must hold 2n rays");
    if (n == 0) { out->n = 0; return TRC_OK; }
    if (!in->dx || !in->dy || !in->dz || !in->e || !hx || !hy || !hz || !nx || !ny || !nz)
        return trc_fail(TRC_ERR_INVALID, "trc_optics_apply: directions, energies, hit points and normals are required");
    if (!out->x || !out->y || !out->z || !out->dx || !out->dy || !out->dz || !out->e || !out->parent)
        return trc_fail(TRC_ERR_INVALID, "trc_optics_apply: output needs x..e and parent");
    HIP_TRY(hipSetDevice(ctx->device));
    double *d_rec = nullptr, *d_opt = nullptr, *d_extra = nullptr;
    double *d_in[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // dx dy dz e ref wl nx ny nz
    uint64_t *d_rid = nullptr;
    double *d_path = nullptr;
    double *d_out[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int32_t *d_blk = nullptr;
    double *d_shift = nullptr;
    // complex indices, material rows, spectra
    const int W = (in->spectra && in->spec_wl) ? in->n_spec : 0;
    const int n_mat = in->mat ? (int)in->n_mat : 0;
    if (W < 0 || W > 4096 || n_mat < 0 || n_mat > 64) return trc_fail(TRC_ERR_INVALID, "trc_optics_apply: n_spec or n_mat out of range");
    if (surf->optics_kind == TRC_OPT_LAMBERTIAN_POLYCHROMATIC && (W < 2 || !out->spectra))
        return trc_fail(TRC_ERR_INVALID, "polychromatic optics: the bundles need spectra (trc_rays.spectra, spec_wl)");
    if (surf->optics_kind == TRC_OPT_REFRACTIVE_MATERIAL && (!in->wavelength || n_mat <= std::max((int)surf->opt[4], (int)surf->opt[5]) || !out->ref_index_im))
        return trc_fail(TRC_ERR_INVALID, "refraction between tabulated materials: the bundle needs wavelengths and the materials' indices at them (trc_rays.mat), the output ref_index_im");
    if (W > 0 && out->spectra && out->n_spec != W) return trc_fail(TRC_ERR_INVALID, "trc_optics_apply: the output's n_spec differs from the input's");
    const int64_t out_cap = out->n;
    double *d_im = nullptr, *d_mat = nullptr, *d_swl = nullptr, *d_spec = nullptr, *d_oim = nullptr, *d_ospec = nullptr;
    int st = TRC_OK;
    do {
        if ((st = upload_record(surf, n_extra, extra, &d_rec, &d_opt, &d_extra))) break;
        if (in->ref_index_im) {
            if ((st = dev_alloc(&d_im, (size_t)n))) break;
            if (hipMemcpy(d_im, in->ref_index_im, (size_t)n * 8, hipMemcpyHostToDevice) != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "memcpy failed"); break; }
        }
        {
            struct { const double *src; int rows; double **dst; } blk2[3] = {{in->mat, 2 * n_mat, &d_mat}, {in->spec_wl, W, &d_swl}, {in->spectra, W, &d_spec}};
            for (auto &b : blk2) {
                if (b.rows <= 0) continue;
                if ((st = dev_alloc(b.dst, (size_t)n * b.rows))) break;
                if (hipMemcpy2D(*b.dst, (size_t)n * 8, b.src, (size_t)in->n * 8, (size_t)n * 8, (size_t)b.rows, hipMemcpyHostToDevice) != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "memcpy failed"); break; }
            }
            if (st) break;
        }
        if (out->ref_index_im && (st = dev_alloc(&d_oim, (size_t)(2 * n)))) break;
        if (W > 0 && out->spectra && (st = dev_alloc(&d_ospec, (size_t)(2 * n) * W))) break;
        const double *src[9] = {in->dx, in->dy, in->dz, in->e, in->ref_index, in->wavelength, nx, ny, nz};
        for (int i = 0; i < 9; ++i) {
            if (!src[i]) continue;
            if ((st = dev_alloc(&d_in[i], (size_t)n))) break;
            if (hipMemcpy(d_in[i], src[i], (size_t)n * 8, hipMemcpyHostToDevice) != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "memcpy failed"); break; }
        }
        if (st) break;
        if (in->x && in->y && in->z) {      // path lengths for the attenuating optics (Absorbant.attenuate, :874-877)
            std::vector<double> path((size_t)n);
            for (int64_t i = 0; i < n; ++i) {
                double ax = hx[i] - in->x[i], ay = hy[i] - in->y[i], az = hz[i] - in->z[i];
                path[(size_t)i] = std::sqrt(ax * ax + ay * ay + az * az);
            }
            if ((st = dev_alloc(&d_path, (size_t)n))) break;
            if (hipMemcpy(d_path, path.data(), (size_t)n * 8, hipMemcpyHostToDevice) != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "memcpy failed"); break; }
        }
        if (in->rid) {
            if ((st = dev_alloc(&d_rid, (size_t)n))) break;
            if (hipMemcpy(d_rid, in->rid, (size_t)n * 8, hipMemcpyHostToDevice) != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "memcpy failed"); break; }
        }
        for (int i = 0; i < 5 && st == TRC_OK; ++i) st = dev_alloc(&d_out[i], (size_t)(2 * n));
        if (st == TRC_OK) st = dev_alloc(&d_blk, (size_t)(2 * n));
        if (st == TRC_OK) st = dev_alloc(&d_shift, (size_t)(2 * n));
        if (st) break;
        OpticsParams P;
        memset(&P, 0, sizeof(P));
        P.rec = d_rec; P.opt = d_opt; P.extra = d_extra; P.n = n;
        P.dx = d_in[0]; P.dy = d_in[1]; P.dz = d_in[2]; P.e = d_in[3]; P.ref = d_in[4]; P.wl = d_in[5];
        P.rid = d_rid; P.ray_offset = 0;
        P.nx = d_in[6]; P.ny = d_in[7]; P.nz = d_in[8]; P.path = d_path;
        P.seed = seed; P.event = bounce;
        P.odx = d_out[0]; P.ody = d_out[1]; P.odz = d_out[2]; P.oe = d_out[3]; P.oref = d_out[4]; P.oblk = d_blk; P.oshift = d_shift;
        P.ref_im = d_im; P.mat = d_mat; P.spec_wl = d_swl; P.spec = d_spec; P.n_mat = n_mat; P.W = W; P.o_im = d_oim; P.o_spec = d_ospec;
        hipLaunchKernelGGL(k_optics_apply, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, P);
        hipError_t se = hipStreamSynchronize(ctx->stream);
        if (se != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "k_optics_apply failed: %s", hipGetErrorString(se)); break; }
        std::vector<double> h[5];
        std::vector<int32_t> blk((size_t)(2 * n));
        for (int i = 0; i < 5; ++i) {
            h[i].resize((size_t)(2 * n));
            if (hipMemcpy(h[i].data(), d_out[i], (size_t)(2 * n) * 8, hipMemcpyDeviceToHost) != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "memcpy failed"); break; }
        }
        if (st) break;
        if (hipMemcpy(blk.data(), d_blk, (size_t)(2 * n) * 4, hipMemcpyDeviceToHost) != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "memcpy failed"); break; }
        std::vector<double> h_shift((size_t)(2 * n));
        if (hipMemcpy(h_shift.data(), d_shift, (size_t)(2 * n) * 8, hipMemcpyDeviceToHost) != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "memcpy failed"); break; }
        std::vector<double> h_im, h_spec;
        if (d_oim) {
            h_im.resize((size_t)(2 * n));
            if (hipMemcpy(h_im.data(), d_oim, (size_t)(2 * n) * 8, hipMemcpyDeviceToHost) != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "memcpy failed"); break; }
        }
        if (d_ospec) {
            h_spec.resize((size_t)(2 * n) * W);
            if (hipMemcpy(h_spec.data(), d_ospec, h_spec.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) { st = trc_fail(TRC_ERR_DEVICE, "memcpy failed"); break; }
        }
        // reflected block first, then refracted block, each in selector order (optics_callables.py:852-857)
        int64_t m = 0;
        for (int b = 0; b < 2; ++b)
            for (int64_t slot = 0; slot < 2 * n; ++slot) {
                if (blk[(size_t)slot] != b) continue;
                int64_t i = slot < n ? slot : slot - n;
                const double sh = h_shift[(size_t)slot];        // (0 but for a periodic boundary: the ray goes on one period along the normal)
                out->x[m] = hx[i] + sh * nx[i]; out->y[m] = hy[i] + sh * ny[i]; out->z[m] = hz[i] + sh * nz[i];
                out->dx[m] = h[0][(size_t)slot]; out->dy[m] = h[1][(size_t)slot]; out->dz[m] = h[2][(size_t)slot];
                out->e[m] = h[3][(size_t)slot];
                if (out->ref_index) out->ref_index[m] = h[4][(size_t)slot];
                if (out->wavelength) out->wavelength[m] = in->wavelength ? in->wavelength[i] : 0.0;
                if (out->ref_index_im) out->ref_index_im[m] = h_im[(size_t)slot];
                if (d_ospec)
                    for (int w = 0; w < W; ++w) {
                        out->spectra[(size_t)w * out_cap + m] = h_spec[(size_t)w * 2 * n + slot];
                        if (out->spec_wl) out->spec_wl[(size_t)w * out_cap + m] = in->spec_wl[(size_t)w * in->n + i];
                    }
                out->parent[m] = i;
                ++m;
            }
        out->n = m;
    } while (0);
    dev_free(d_rec); dev_free(d_opt); dev_free(d_extra); dev_free(d_rid); dev_free(d_blk); dev_free(d_shift); dev_free(d_path);
    dev_free(d_im); dev_free(d_mat); dev_free(d_swl); dev_free(d_spec); dev_free(d_oim); dev_free(d_ospec);
    for (int i = 0; i < 9; ++i) dev_free(d_in[i]);
    for (int i = 0; i < 5; ++i) dev_free(d_out[i]);
    return st;
}
