// C ABI of libmmc_hip.so (see include/mmc_hip.h): handle management, input
// validation in the spirit of MultiframeMotionCoupling.init, the one-off
// set-up that replaces init_u_prost's sparse-matrix construction, and the
// PDHG loop that replaces prost.solve / flexBox.runAlgorithm.
#include "../../include/mmc_hip.h"
#include "mmc_internal.h"

#include <algorithm>
#include <array>
#include <map>
#include <mutex>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <limits.h>
#include <string>
#include <vector>

struct mmc_handle {
    mmc_params prm;
    Geom g;
    Arrays a;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int L = 0, has_prev = 0, has_next = 0;
    int pair_lo = 0, npairs = 0;          // global pairs [pair_lo, pair_lo+npairs) this block touches
    std::vector<float> blur;              // kh x kw column-major; empty = no blur
    int kh = 1, kw = 1;
    // 1-D factors of the sampling operator in banded form (mmc_set_sampling_taps); empty: 'average' / 'stride'
    std::vector<int> sfy, sfx;
    std::vector<double> swy, swx;
    int sty = 0, stx = 0;
    bool dataop_built = false;            // composite kernel, step sizes and border correction are current
    std::vector<void *> eallocs;          // device buffers of the border correction
    bool have_f = false, have_flow = false, have_state = false, inited = false, have_h = false;
    float h = 1.f;
    std::vector<void *> allocs;
    unsigned long long *scratch = nullptr;   // one frame of 64-bit words for mmc_init
    int *ovf_cursor = nullptr;               // (L+1)*ntiles counters for building the overflow lists
    size_t ovf_capacity = 0;                 // entries allocated for a.ovf_idx
    size_t ovf_total = 0;                    // entries in use
    std::string err;
    bool geometry_only = false;              // no per-pixel state (mmc_internal_create_geometry)
    size_t plane_skew = 0;                   // bytes between the starts of consecutive per-pixel planes modulo their allocation alignment
    int diagnostics = MMC_DIAG_AUTO;         // when the second group of mmc_stats is evaluated
    double balance = 1.0;                    // step balance s: tau * s, sigma / s
    bool adapt = false;                      // mmc_solve chooses s by residual balancing
    double adapt_alpha = 0.5;
    bool profiling = false;
    std::vector<hipEvent_t> events;       // 4 per profiled iteration
    double prof_ms[3] = {0, 0, 0};
    int prof_iters = 0;
    // sweep-level API: (kind, start, stop) per timed launch group, kind 0 = primal, 1 = dual
    std::vector<hipEvent_t> sweep_events;
    std::vector<int> sweep_kinds;
    size_t sweep_used = 0;
};

enum { PROF_MAX_ITERS = 512 };

static thread_local std::string g_create_err;
static thread_local const char *g_static_err = nullptr;   // set by handle-free entry points, cleared by mmc_create

static int fail(mmc_handle *h, int code, const std::string &msg) {
    if (h) h->err = msg; else { g_create_err = msg; g_static_err = nullptr; }
    return code;
}

int mmc_static_fail(int code, const char *literal) {
    g_static_err = literal;
    return code;
}

#define HIPCHK(h, call)                                                                   \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail((h), MMC_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

#define REQUIRE_HANDLE(h) do { if (!(h)) return MMC_ERR_INVALID_ARG; } while (0)

static size_t frame_slots(const mmc_handle *h) { return (size_t)h->L + 2; }

// `skew` bytes (a multiple of 256) are left unused in front of the array, see PLANE_SKEW.
template <typename T>
static int dev_alloc(mmc_handle *h, T **ptr, size_t count, bool zero = true, size_t skew = 0) {
    void *p = nullptr;
    size_t bytes = (count ? count : 1) * sizeof(T);
    // + 16: the last row quad of a plane is loaded whole even when its column ends after one to three rows
    HIPCHK(h, hipMalloc(&p, bytes + skew + 16));
    h->allocs.push_back(p);
    p = static_cast<char *>(p) + skew;
    if (zero) HIPCHK(h, hipMemsetAsync(p, 0, bytes, h->stream));
    *ptr = (T *)p;
    return MMC_OK;
}

// Large allocations come back aligned to 2 MiB, so the per-pixel planes would all start at the same
// address modulo any power of two.  A sweep streams through ten or more of them at the same offset,
// and on MI355X planes whose distance is a multiple of 256 KiB move measurably slower
// (tools/stream_bench8.hip: 10 read streams, 6 written back: 5.25 TB/s, against 5.49 TB/s with the
// planes 320 KiB further apart each).  Which distance is best depends on the frame size (it acts
// together with the distance between the ranges the eight XCD classes walk), so plane i is shifted
// by i * skew inside its allocation and mmc_create picks the skew for large blocks by timing a
// streaming probe of the sweeps' access pattern over the still empty planes.
enum : size_t { PLANE_SKEW_DEFAULT = 320u << 10, PLANE_SKEW_MAX = 1u << 20, PLANE_SLOTS = 16 };
// candidates 64 KiB + k * 32 KiB, k < 16; k = 8 is the default
enum { PLANE_SKEW_STEPS = 16, PLANE_SKEW_DEFAULT_STEP = 8, PLANE_PROBE_FRAMES = 12, PLANE_PROBE_ROUNDS = 6 };
static const float PLANE_PROBE_WARM_MS = 40.f, PLANE_PROBE_MARGIN = 0.985f;

extern "C" int mmc_abi_version(void) { return MMC_ABI_VERSION; }

// ---- per-device pool of streams and events (see mmc_internal.h) -------------------------------
struct DevicePool {
    std::vector<hipStream_t> streams;
    std::vector<hipEvent_t> events[2];      // [0] timing disabled, [1] timing enabled
};
static std::mutex g_pool_mutex;
static std::map<int, DevicePool> g_pool;

hipStream_t mmc_pool_stream_acquire(int device) {
#ifndef MMC_NO_POOL
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        DevicePool &dp = g_pool[device];
        if (!dp.streams.empty()) { hipStream_t s = dp.streams.back(); dp.streams.pop_back(); return s; }
    }
#endif
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
    return s;
}

void mmc_pool_stream_release(int device, hipStream_t s) {
    if (!s) return;
#ifdef MMC_NO_POOL
    (void)device;
    hipStreamDestroy(s);
#else
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    g_pool[device].streams.push_back(s);
#endif
}

hipEvent_t mmc_pool_event_acquire(int device, bool timing) {
#ifndef MMC_NO_POOL
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        std::vector<hipEvent_t> &ev = g_pool[device].events[timing ? 1 : 0];
        if (!ev.empty()) { hipEvent_t e = ev.back(); ev.pop_back(); return e; }
    }
#endif
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, timing ? hipEventDefault : hipEventDisableTiming) != hipSuccess) return nullptr;
    return e;
}

void mmc_pool_event_release(int device, bool timing, hipEvent_t e) {
    if (!e) return;
#ifdef MMC_NO_POOL
    (void)device; (void)timing;
    hipEventDestroy(e);
#else
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    g_pool[device].events[timing ? 1 : 0].push_back(e);
#endif
}

// HIP the library was compiled against and the runtime it found loaded in this process (a host
// program that loads PyTorch-ROCm first runs every HIP library on the copy PyTorch brings along).
extern "C" int mmc_hip_versions(int *built_against, int *runtime) {
    if (built_against) *built_against = HIP_VERSION;
    if (runtime) {
        *runtime = 0;
        if (hipRuntimeGetVersion(runtime) != hipSuccess) { (void)hipGetLastError(); return MMC_ERR_HIP; }
    }
    return MMC_OK;
}

extern "C" const char *mmc_last_error(const mmc_handle *h) {
    if (h) return h->err.c_str();
    return g_static_err ? g_static_err : g_create_err.c_str();
}

// Picks the shift between consecutive planes (see PLANE_SKEW_DEFAULT).  Blocks of at least 64 MiB per
// plane (on smaller ones the timings scatter more than the shifts differ) whose frames take the row-quad sweeps are probed: each candidate is timed with one dual-like
// and one primal-like pass of the streaming probe over the first frames of the (zeroed) planes.
// The clocks of an idle GPU settle only after some milliseconds of load and single timings scatter by
// about as much as neighbouring candidates differ, so the probe first runs untimed for a while, then
// times every candidate PLANE_PROBE_ROUNDS times in alternating order, keeps the minimum of each and
// takes the best 1-2-1 weighted mean over neighbouring candidates (the response is smooth in the shift).
// The probe ranks shifts as the sweeps do where they differ by several per cent (2160-row frames);
// differences of about one per cent between its timings do not carry over to the sweeps (1080-row
// frames: the default is the sweeps' optimum, the probe is flat), so the default stays unless a
// candidate beats it by PLANE_PROBE_MARGIN.  The choice is kept per frame size and device (g_skew_cache).
static std::mutex g_skew_mutex;
static std::map<std::array<int, 4>, size_t> g_skew_cache;     // (Ny, Nx, probed frames, device) -> chosen shift

// Blocks of at least 64 MiB per plane whose frames take the row-quad sweeps are probed.
static bool plane_is_probed(const mmc_handle *h) {
    const size_t plane_bytes = frame_slots(h) * (size_t)h->g.N * sizeof(float);
    return plane_bytes >= ((size_t)64 << 20) && h->g.Ny % 4 == 0;
}

static_assert(((size_t)64 << 10) + PLANE_SKEW_DEFAULT_STEP * ((size_t)32 << 10) == PLANE_SKEW_DEFAULT, "default shift must be a probe candidate");
static int choose_plane_skew(mmc_handle *h, const std::vector<float *> &raw, size_t *skew) {
    const Geom &g = h->g;
    // tuning override: MMC_PLANE_SKEW=<bytes> fixes the shift, MMC_PROBE_TRACE=1 prints the probe's timings
    if (const char *e = getenv("MMC_PLANE_SKEW")) { *skew = std::min((size_t)atol(e) & ~(size_t)255, (size_t)PLANE_SKEW_MAX); return MMC_OK; }
    if (!plane_is_probed(h) || raw.size() < 12) return MMC_OK;
    // one probe per frame size and device in a process: the blocks of a group share its result
    const int nprobe = std::min(h->L, (int)PLANE_PROBE_FRAMES);
    const std::array<int, 4> key = {g.Ny, g.Nx, nprobe, h->prm.device};
    {
        std::lock_guard<std::mutex> lock(g_skew_mutex);
        auto it = g_skew_cache.find(key);
        if (it != g_skew_cache.end() && !getenv("MMC_PROBE_TRACE")) { *skew = it->second; return MMC_OK; }
    }
    hipEvent_t e0 = mmc_pool_event_acquire(h->prm.device, true), e1 = mmc_pool_event_acquire(h->prm.device, true);
    if (!e0 || !e1) {
        mmc_pool_event_release(h->prm.device, true, e0);
        mmc_pool_event_release(h->prm.device, true, e1);
        return fail(h, MMC_ERR_HIP, "plane placement probe: hipEventCreate failed");
    }
    int rc = MMC_OK;
    const int nfp = std::min(h->L, (int)PLANE_PROBE_FRAMES);
    float tmin[PLANE_SKEW_STEPS];
    for (float &t : tmin) t = 0.f;
    auto cand_of = [](int kc) { return ((size_t)64 << 10) + (size_t)kc * ((size_t)32 << 10); };
    auto pass = [&](int kc, float *ms) -> hipError_t {
        float *pl[12];
        for (int i = 0; i < 12; ++i) pl[i] = raw[(size_t)i] + (size_t)i * cand_of(kc) / sizeof(float);
        hipError_t e = hipEventRecord(e0, h->stream);
        if (e == hipSuccess) e = mmc_launch_stream_probe(g, pl, nfp, true, h->stream);
        if (e == hipSuccess) e = mmc_launch_stream_probe(g, pl, nfp, false, h->stream);
        if (e == hipSuccess) e = hipEventRecord(e1, h->stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipEventElapsedTime(ms, e0, e1);
        return e;
    };
    hipError_t e = hipSuccess;
    float ms = 0.f, warm = 0.f;
    for (int i = 0; i < 400 && warm < PLANE_PROBE_WARM_MS && e == hipSuccess; ++i) { e = pass(i % PLANE_SKEW_STEPS, &ms); warm += ms; }
    for (int round = 0; round < PLANE_PROBE_ROUNDS && e == hipSuccess; ++round)
        for (int i = 0; i < PLANE_SKEW_STEPS && e == hipSuccess; ++i) {
            const int kc = (round & 1) ? PLANE_SKEW_STEPS - 1 - i : i;
            e = pass(kc, &ms);
            if (e == hipSuccess && (tmin[kc] == 0.f || ms < tmin[kc])) tmin[kc] = ms;
        }
    if (e != hipSuccess) rc = fail(h, MMC_ERR_HIP, std::string("plane placement probe: ") + hipGetErrorString(e));
    else {
        float best = 0.f, tdef = 0.f;
        int kbest = PLANE_SKEW_DEFAULT_STEP;
        for (int kc = 0; kc < PLANE_SKEW_STEPS; ++kc) {
            const float lo = tmin[kc > 0 ? kc - 1 : kc], hi = tmin[kc + 1 < PLANE_SKEW_STEPS ? kc + 1 : kc];
            const float t = 0.25f * lo + 0.5f * tmin[kc] + 0.25f * hi;
            if (kc == PLANE_SKEW_DEFAULT_STEP) tdef = t;
            if (best == 0.f || t < best) { best = t; kbest = kc; }
        }
        *skew = cand_of(best < PLANE_PROBE_MARGIN * tdef ? kbest : PLANE_SKEW_DEFAULT_STEP);
        if (getenv("MMC_PROBE_TRACE")) {
            fprintf(stderr, "plane shift probe (%d frames, warm %.1f ms):", nfp, warm);
            for (int kc = 0; kc < PLANE_SKEW_STEPS; ++kc) fprintf(stderr, " %zu:%.4f", cand_of(kc) >> 10, tmin[kc]);
            fprintf(stderr, " -> %zu\n", *skew);
        }
    }
    mmc_pool_event_release(h->prm.device, true, e0);
    mmc_pool_event_release(h->prm.device, true, e1);
    if (rc == MMC_OK) {
        std::lock_guard<std::mutex> lock(g_skew_mutex);
        g_skew_cache[key] = *skew;
    }
    return rc;
}

extern "C" size_t mmc_debug_plane_skew(const mmc_handle *h) { return h ? h->plane_skew : 0; }
extern "C" int mmc_debug_separable(const mmc_handle *h) { return h ? h->g.sep : 0; }

// geometry_only: the handle carries the geometry and the data operator D*B (composite kernel, its row
// and column sums) and none of the per-pixel state; the single-frame solve owns such a handle.
static int create_handle(const mmc_params *p, bool geometry_only, mmc_handle **out) {
    if (!p || !out) return fail(nullptr, MMC_ERR_INVALID_ARG, "mmc_create: null argument");
    *out = nullptr;
    if (p->size != sizeof(mmc_params) || p->version != MMC_ABI_VERSION)
        return fail(nullptr, MMC_ERR_INVALID_ARG, "mmc_create: params size/version mismatch");
    // MultiframeMotionCoupling.m:228-231 checkposScalar
    if (!(p->factor > 0)) return fail(nullptr, MMC_ERR_INVALID_ARG, "Input must be > 0 (factor)");
    if (!(p->alpha > 0.f)) return fail(nullptr, MMC_ERR_INVALID_ARG, "Input must be > 0 (alpha)");
    if (!std::isnan(p->kappa) && !(p->kappa > 0.f))
        return fail(nullptr, MMC_ERR_INVALID_ARG, "Input must be > 0 (kappa)");
    if (p->ny <= 0 || p->nx <= 0) return fail(nullptr, MMC_ERR_INVALID_ARG, "Load an actual video sequence (empty frame)");
    if (p->nc_global <= 0 || p->frame_begin < 0 || p->frame_end > p->nc_global || p->frame_end <= p->frame_begin)
        return fail(nullptr, MMC_ERR_INVALID_ARG, "mmc_create: frame block must be a non-empty range inside [0, nc_global)");
    if (p->flow_direction < 0 || p->flow_direction > 2)
        return fail(nullptr, MMC_ERR_INVALID_ARG, "Invalid flow direction.");
    if (p->sampling != MMC_SAMPLING_AVERAGE && p->sampling != MMC_SAMPLING_STRIDE && p->sampling != MMC_SAMPLING_TAPS)
        return fail(nullptr, MMC_ERR_UNSUPPORTED, "interpMethod: unknown sampling code");
    if (p->sampling == MMC_SAMPLING_AVERAGE && p->factor != 4)
        return fail(nullptr, MMC_ERR_UNSUPPORTED,
                    "interpMethod 'average' is a block mean only for factor 4 (samplingOperator.m:70-73); for other factors pass "
                    "the rows of the sampling matrix with mmc_set_sampling_taps");
    if ((long long)p->ny * p->factor * (long long)p->nx * p->factor > (long long)INT_MAX / 4)
        return fail(nullptr, MMC_ERR_INVALID_ARG, "mmc_create: frame too large for 32-bit pixel indexing");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, MMC_ERR_NO_DEVICE, "mmc_create: no HIP device available");
    if (p->device < 0 || p->device >= ndev)
        return fail(nullptr, MMC_ERR_INVALID_ARG, "mmc_create: device ordinal out of range");
    mmc_handle *h = new mmc_handle();
    h->prm = *p;
    auto bail = [&](int code) {
        g_create_err = h->err;
        mmc_destroy(h);
        return code;
    };
    if (hipSetDevice(p->device) != hipSuccess) { h->err = "hipSetDevice failed"; return bail(MMC_ERR_HIP); }
    if (p->stream) {
        h->stream = (hipStream_t)p->stream;
    } else {
        h->stream = mmc_pool_stream_acquire(p->device);
        if (!h->stream) {
            int built = 0, rt = 0;
            mmc_hip_versions(&built, &rt);
            h->err = "hipStreamCreate failed (library built against HIP " + std::to_string(built) + ", runtime " + std::to_string(rt) + ")";
            return bail(MMC_ERR_HIP);
        }
        h->own_stream = true;
    }
    h->L = p->frame_end - p->frame_begin;
    h->has_prev = p->frame_begin > 0;
    h->has_next = p->frame_end < p->nc_global;
    h->pair_lo = p->frame_begin > 0 ? p->frame_begin - 1 : 0;
    {
        int hi = p->frame_end < p->nc_global - 1 ? p->frame_end : p->nc_global - 1;
        h->npairs = hi > h->pair_lo ? hi - h->pair_lo : 0;
    }

    Geom &g = h->g;
    memset(&g, 0, sizeof(g));
    g.ny = p->ny; g.nx = p->nx; g.n = p->ny * p->nx;
    g.f = p->factor;
    g.Ny = p->ny * p->factor; g.Nx = p->nx * p->factor; g.N = g.Ny * g.Nx;
    g.L = h->L; g.frame_begin = p->frame_begin; g.nc_global = p->nc_global;
    g.dir = p->flow_direction;
    g.alpha = p->alpha;
    if (std::isnan(p->kappa)) { g.model = MODEL_ADD; g.kappa = 1.f; g.radius1 = p->alpha; g.radius2 = p->alpha; }
    else if (p->kappa == 1.f) { g.model = MODEL_K1; g.kappa = 1.f; g.radius1 = 2.f * p->alpha; g.radius2 = 0.f; }
    else { g.model = MODEL_IC; g.kappa = p->kappa; g.radius1 = p->alpha; g.radius2 = p->alpha; }
    g.tiles_r = (g.Ny + TILE_R - 1) / TILE_R;
    g.tiles_c = (g.Nx + TILE_C - 1) / TILE_C;
    g.ntiles = g.tiles_r * g.tiles_c;
    {
        const int q = (g.ntiles + 7) / 8;
        const int full = g.ntiles / q, rest = g.ntiles - full * q;     // classes 0..full-1 hold q tiles, class `full` the rest
        g.dv_tiles_r = make_fastdiv((unsigned)g.tiles_r);
        g.dv_Ny = make_fastdiv((unsigned)g.Ny);
        g.dv_q = make_fastdiv((unsigned)q);
        g.dv_qlast = make_fastdiv((unsigned)(rest > 0 ? rest : q));
        // The second use of a frame's planes (scatter sources, warp taps) comes one sweep over the
        // range later; keeping a range near 290 tiles per class (what a 1080p frame has) keeps that
        // distance inside the Infinity Cache for larger frames as well.
        const int nsub = (q + 287) / 288;
        g.chunk = nsub > 1 ? (q + nsub - 1) / nsub : 0;
        g.dv_chunk = make_fastdiv((unsigned)(g.chunk > 0 ? g.chunk : 1));
    }
    g.inv_h = 1.f;
    g.wsgn = -1.f;
    g.tau_s = g.sigma_s = 1.f;
#ifdef MMC_ABLATION
    if (const char *e = getenv("MMC_DEBUG")) g.dbg = atoi(e);
#endif
    if (p->h > 0.f) { h->h = p->h; h->have_h = true; g.inv_h = 1.f / p->h; }

    Arrays &a = h->a;
    memset(&a, 0, sizeof(a));
    const size_t fs = frame_slots(h) * (size_t)g.N;
    float **per_px[] = {&a.u, &a.w, &a.ub, &a.wb, &a.g1t, &a.g1x, &a.g1y, &a.g2t, &a.g2x, &a.g2y, &a.vx, &a.vy, &a.cW};
    int rc;
    h->geometry_only = geometry_only;
    if (geometry_only) {
        if ((rc = dev_alloc(h, &a.cA, (size_t)g.N)) != MMC_OK) return bail(rc);
        if ((rc = dev_alloc(h, &a.sigp, (size_t)g.n)) != MMC_OK) return bail(rc);
        if ((rc = dev_alloc(h, &a.cker, (size_t)MMC_MAX_CK * MMC_MAX_CK)) != MMC_OK) return bail(rc);
        if ((rc = dev_alloc(h, &a.ckq, (size_t)(MMC_MAX_CK + 8) * MMC_MAX_CK)) != MMC_OK) return bail(rc);
        if ((rc = dev_alloc(h, &a.ksep, (size_t)3 * MMC_KSEP_PITCH)) != MMC_OK) return bail(rc);
        if ((rc = dev_alloc(h, &a.ckp, (size_t)(MMC_MAX_CK + p->factor) * (MMC_MAX_CK + p->factor))) != MMC_OK) return bail(rc);
        if ((rc = dev_alloc(h, &a.sums, (size_t)MMC_NUM_SUMS_INT * MMC_SUM_SLOTS)) != MMC_OK) return bail(rc);
        if (hipStreamSynchronize(h->stream) != hipSuccess) { h->err = "hipStreamSynchronize failed"; return bail(MMC_ERR_HIP); }
        *out = h;
        return MMC_OK;
    }
    {
        // Every plane gets room for the largest shift it can end up with: the fixed one (default or
        // MMC_PLANE_SKEW) when the block is not probed, the largest probe candidate otherwise.  Only the
        // frames themselves are zeroed.
        const size_t nplanes = sizeof(per_px) / sizeof(per_px[0]);
        size_t max_skew = PLANE_SKEW_DEFAULT;
        if (const char *e = getenv("MMC_PLANE_SKEW")) max_skew = std::min((size_t)atol(e) & ~(size_t)255, (size_t)PLANE_SKEW_MAX);
        else if (plane_is_probed(h)) max_skew = ((size_t)64 << 10) + (size_t)(PLANE_SKEW_STEPS - 1) * ((size_t)32 << 10);
        const size_t room = (nplanes - 1) * max_skew / sizeof(float);
        std::vector<float *> raw;
        for (float **pp : per_px) {
            if ((rc = dev_alloc(h, pp, fs + room, false)) != MMC_OK) return bail(rc);
            raw.push_back(*pp);
        }
        size_t skew = PLANE_SKEW_DEFAULT;
        if ((rc = choose_plane_skew(h, raw, &skew)) != MMC_OK) return bail(rc);
        h->plane_skew = skew;
        for (size_t i = 0; i < raw.size(); ++i) {
            *per_px[i] = raw[i] + i * skew / sizeof(float);
            if (hipMemsetAsync(*per_px[i], 0, fs * sizeof(float), h->stream) != hipSuccess) { h->err = "hipMemsetAsync failed"; return bail(MMC_ERR_HIP); }
        }
    }
    if ((rc = dev_alloc(h, &a.idq, (size_t)h->L * MMC_IDQ_PER_FRAME(g), true, 13 * PLANE_SKEW_DEFAULT)) != MMC_OK) return bail(rc);
    if ((rc = dev_alloc(h, &a.p, (size_t)h->L * g.n, true, 14 * PLANE_SKEW_DEFAULT)) != MMC_OK) return bail(rc);
    if ((rc = dev_alloc(h, &a.f, (size_t)h->L * g.n, true, 15 * PLANE_SKEW_DEFAULT)) != MMC_OK) return bail(rc);
    if ((rc = dev_alloc(h, &a.cA, (size_t)g.N)) != MMC_OK) return bail(rc);
    if ((rc = dev_alloc(h, &a.sigp, (size_t)g.n)) != MMC_OK) return bail(rc);
    if ((rc = dev_alloc(h, &a.cker, (size_t)MMC_MAX_CK * MMC_MAX_CK)) != MMC_OK) return bail(rc);
    if ((rc = dev_alloc(h, &a.ckq, (size_t)(MMC_MAX_CK + 8) * MMC_MAX_CK)) != MMC_OK) return bail(rc);
    if ((rc = dev_alloc(h, &a.ksep, (size_t)3 * MMC_KSEP_PITCH)) != MMC_OK) return bail(rc);
    if ((rc = dev_alloc(h, &a.ckp, (size_t)(MMC_MAX_CK + p->factor) * (MMC_MAX_CK + p->factor))) != MMC_OK) return bail(rc);
    if ((rc = dev_alloc(h, &a.bbox, (size_t)(h->L + 1) * g.ntiles * 4)) != MMC_OK) return bail(rc);
    { float **fx[] = {&a.q1, &a.q2, &a.u0, &a.w0}; for (float **pp : fx) if ((rc = dev_alloc(h, pp, (size_t)g.N)) != MMC_OK) return bail(rc); }
    if ((rc = dev_alloc(h, &a.tbox, (size_t)h->L * g.ntiles * 4)) != MMC_OK) return bail(rc);
    if ((rc = dev_alloc(h, &a.bstat, (size_t)(h->L + 1) * g.ntiles * 3)) != MMC_OK) return bail(rc);
    if ((rc = dev_alloc(h, &a.ovf_ptr, (size_t)(h->L + 1) * g.ntiles + 1)) != MMC_OK) return bail(rc);
    if ((rc = dev_alloc(h, &h->ovf_cursor, (size_t)(h->L + 1) * g.ntiles)) != MMC_OK) return bail(rc);
    if ((rc = dev_alloc(h, &h->scratch, (size_t)g.N)) != MMC_OK) return bail(rc);
    if ((rc = dev_alloc(h, &a.sums, (size_t)MMC_NUM_SUMS_INT * MMC_SUM_SLOTS)) != MMC_OK) return bail(rc);
    if (hipStreamSynchronize(h->stream) != hipSuccess) { h->err = "hipStreamSynchronize failed"; return bail(MMC_ERR_HIP); }
    *out = h;
    return MMC_OK;
}

extern "C" int mmc_create(const mmc_params *p, mmc_handle **out) { return create_handle(p, false, out); }
int mmc_internal_create_geometry(const mmc_params *p, mmc_handle **out) { return create_handle(p, true, out); }

extern "C" int mmc_destroy(mmc_handle *h) {
    if (!h) return MMC_OK;
    // tear-down: nothing useful can be done about a failing release
    (void)hipSetDevice(h->prm.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (void *p : h->allocs) (void)hipFree(p);
    for (void *p : h->eallocs) (void)hipFree(p);
    if (h->a.ovf_idx) (void)hipFree(h->a.ovf_idx);
    for (hipEvent_t e : h->events) mmc_pool_event_release(h->prm.device, true, e);
    for (hipEvent_t e : h->sweep_events) mmc_pool_event_release(h->prm.device, true, e);
    if (h->own_stream && h->stream) mmc_pool_stream_release(h->prm.device, h->stream);      // idle: synchronised above
    delete h;
    return MMC_OK;
}

static int use_device(mmc_handle *h) {
    HIPCHK(h, hipSetDevice(h->prm.device));
    // the launchers report hipGetLastError(): drop what an unrelated earlier call of this thread left behind
    (void)hipGetLastError();
    return MMC_OK;
}

extern "C" int mmc_set_lowres(mmc_handle *h, const float *f) {
    REQUIRE_HANDLE(h);
    if (!f) return fail(h, MMC_ERR_INVALID_ARG, "mmc_set_lowres: null data");
    int rc = use_device(h); if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(h->a.f, f, (size_t)h->L * h->g.n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->have_f = true;
    return MMC_OK;
}

extern "C" int mmc_set_blur(mmc_handle *h, const float *k, int kh, int kw) {
    REQUIRE_HANDLE(h);
    if (!k) { h->blur.clear(); h->kh = h->kw = 1; h->inited = false; h->dataop_built = false; return MMC_OK; }
    // helpers/RepConvMtx.m:6-8
    if (kh <= 0 || kw <= 0 || (kh % 2) != 1 || (kw % 2) != 1)
        return fail(h, MMC_ERR_INVALID_ARG, "kernel has even dimensions (which is not implemented)");
    if (kh > MMC_MAX_CK || kw > MMC_MAX_CK)
        return fail(h, MMC_ERR_UNSUPPORTED, "mmc_set_blur: blur kernel too large");
    h->blur.assign(k, k + (size_t)kh * kw);
    h->kh = kh; h->kw = kw;
    h->inited = false;
    h->dataop_built = false;
    return MMC_OK;
}

extern "C" int mmc_set_sampling_taps(mmc_handle *h, const int32_t *first_y, const float *w_y, int taps_y,
                                     const int32_t *first_x, const float *w_x, int taps_x) {
    REQUIRE_HANDLE(h);
    if (h->prm.sampling != MMC_SAMPLING_TAPS)
        return fail(h, MMC_ERR_STATE, "mmc_set_sampling_taps: the handle was not created with MMC_SAMPLING_TAPS");
    if (!first_y || !w_y || !first_x || !w_x || taps_y <= 0 || taps_x <= 0)
        return fail(h, MMC_ERR_INVALID_ARG, "mmc_set_sampling_taps: null or empty taps");
    const Geom &g = h->g;
    for (int ax = 0; ax < 2; ++ax) {
        const int nout = ax ? g.nx : g.ny, nin = ax ? g.Nx : g.Ny, T = ax ? taps_x : taps_y;
        const int32_t *first = ax ? first_x : first_y;
        const float *w = ax ? w_x : w_y;
        for (int R = 0; R < nout; ++R)
            for (int m = 0; m < T; ++m) {
                const float v = w[(size_t)R * T + m];
                if (!std::isfinite(v)) return fail(h, MMC_ERR_INVALID_ARG, "mmc_set_sampling_taps: weight is not finite");
                if (v != 0.f && (first[R] + m < 0 || first[R] + m >= nin))
                    return fail(h, MMC_ERR_INVALID_ARG, "mmc_set_sampling_taps: non-zero weight outside the frame");
            }
    }
    h->sfy.assign(first_y, first_y + g.ny); h->swy.assign(w_y, w_y + (size_t)g.ny * taps_y); h->sty = taps_y;
    h->sfx.assign(first_x, first_x + g.nx); h->swx.assign(w_x, w_x + (size_t)g.nx * taps_x); h->stx = taps_x;
    h->inited = false;
    h->dataop_built = false;
    return MMC_OK;
}

extern "C" int mmc_num_flow_pairs(const mmc_handle *h) { return h ? h->npairs : MMC_ERR_INVALID_ARG; }
extern "C" int mmc_num_state_frames(const mmc_handle *h) { return h ? h->L + h->has_next : MMC_ERR_INVALID_ARG; }

// Flow of the block's pairs from two host planes (displacement along columns, along rows), each
// Ny x Nx x npairs; occ as in mmc_set_flow.
int mmc_internal_set_flow_planes(mmc_handle *h, const float *vcol, const float *vrow, const uint8_t *occ) {
    REQUIRE_HANDLE(h);
    int rc = use_device(h); if (rc) return rc;
    const Geom &g = h->g;
    if (h->npairs > 0) {
        if (!vcol || !vrow) return fail(h, MMC_ERR_INVALID_ARG, "mmc_set_flow: null flow");
        // pair with global index gq lives in frame slot (gq - frame_begin) + 1
        const size_t slot0 = (size_t)(h->pair_lo - g.frame_begin + 1);
        const size_t cnt = (size_t)h->npairs * g.N;
        float *dvx = h->a.vx + slot0 * g.N, *dvy = h->a.vy + slot0 * g.N;
        HIPCHK(h, hipMemcpyAsync(dvx, vcol, cnt * sizeof(float), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(dvy, vrow, cnt * sizeof(float), hipMemcpyHostToDevice, h->stream));
        if (occ) {
            uint8_t *docc = nullptr;
            HIPCHK(h, hipMalloc((void **)&docc, cnt));
            hipError_t e = hipMemcpyAsync(docc, occ, cnt, hipMemcpyHostToDevice, h->stream);
            if (e == hipSuccess) e = mmc_launch_fold_occ(g, dvx, docc, cnt, h->stream);
            hipError_t e2 = hipStreamSynchronize(h->stream);
            (void)hipFree(docc);
            HIPCHK(h, e);
            HIPCHK(h, e2);
        }
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    h->have_flow = true;
    h->inited = false;
    return MMC_OK;
}

extern "C" int mmc_set_flow(mmc_handle *h, const float *v, const uint8_t *occ) {
    REQUIRE_HANDLE(h);
    const size_t cnt = (size_t)h->npairs * h->g.N;
    return mmc_internal_set_flow_planes(h, v, v ? v + cnt : nullptr, occ);
}

hipStream_t mmc_internal_stream(const mmc_handle *h) { return h ? h->stream : nullptr; }

// Keys cubic (a = -1/2), the kernel of imresize's 'bicubic'
static double keys_cubic(double x) {
    x = std::fabs(x);
    if (x <= 1.0) return 1.5 * x * x * x - 2.5 * x * x + 1.0;
    if (x <= 2.0) return -0.5 * x * x * x + 2.5 * x * x - 4.0 * x + 2.0;
    return 0.0;
}

// imresize's contribution rule for an integer zoom: output sample x (1-based) sits at
// u = x/zoom + (1 - 1/zoom)/2, taps floor(u - 2) + {0..5} (1-based input indices)
static void make_bicubic_tab(int zoom, BicubicTab *tab) {
    memset(tab, 0, sizeof(*tab));
    for (int ph = 0; ph < zoom; ++ph) {
        const double x = ph + 1.0;                         // first output sample of this phase; low-res index 0
        const double u = x / zoom + 0.5 * (1.0 - 1.0 / zoom);
        const double left = std::floor(u - 2.0);
        double w[MMC_BICUBIC_TAPS], sum = 0.0;
        for (int k = 0; k < MMC_BICUBIC_TAPS; ++k) { w[k] = keys_cubic(u - (left + k)); sum += w[k]; }
        for (int k = 0; k < MMC_BICUBIC_TAPS; ++k) tab->w[ph][k] = (float)(w[k] / sum);
        tab->first[ph] = (int)left - 1;                    // 0-based offset from the low-res index x_out / zoom
    }
}

extern "C" int mmc_set_state(mmc_handle *h, const float *u, const float *w) {
    REQUIRE_HANDLE(h);
    int rc = use_device(h); if (rc) return rc;
    const Geom &g = h->g;
    if (u && g.model == MODEL_IC && !w) return fail(h, MMC_ERR_INVALID_ARG, "mmc_set_state: this model needs w");
    if (!u && w) return fail(h, MMC_ERR_INVALID_ARG, "mmc_set_state: w without u");
    const size_t nloc = (size_t)h->L * g.N, nall = (size_t)(h->L + h->has_next) * g.N;
    Arrays &a = h->a;
    if (u) {
        HIPCHK(h, hipMemcpyAsync(a.u + g.N, u, nloc * sizeof(float), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(a.ub + g.N, u, nall * sizeof(float), hipMemcpyHostToDevice, h->stream));
        if (g.model == MODEL_IC) {
            HIPCHK(h, hipMemcpyAsync(a.w + g.N, w, nloc * sizeof(float), hipMemcpyHostToDevice, h->stream));
            HIPCHK(h, hipMemcpyAsync(a.wb + g.N, w, nall * sizeof(float), hipMemcpyHostToDevice, h->stream));
        }
    } else {
        // u = w = imresize(f, factor), computed on the device (MultiframeMotionCoupling.m:415-419, :686-687)
        if (!h->have_f) return fail(h, MMC_ERR_STATE, "mmc_set_state: the bicubic start point needs the low-resolution data (mmc_set_lowres)");
        if (g.f > MMC_BICUBIC_MAX_ZOOM)
            return fail(h, MMC_ERR_UNSUPPORTED, "mmc_set_state: the built-in bicubic start point covers zoom factors up to 8; pass u and w");
        BicubicTab tab;
        make_bicubic_tab(g.f, &tab);
        HIPCHK(h, mmc_launch_bicubic_start(g, a, tab, h->stream));
    }
    // duals start at zero
    const size_t fs = frame_slots(h) * (size_t)g.N * sizeof(float);
    float *duals[] = {a.g1t, a.g1x, a.g1y, a.g2t, a.g2x, a.g2y};
    for (float *d : duals) HIPCHK(h, hipMemsetAsync(d, 0, fs, h->stream));
    HIPCHK(h, hipMemsetAsync(a.p, 0, (size_t)h->L * g.n * sizeof(float), h->stream));
    HIPCHK(h, hipMemsetAsync(a.q1, 0, (size_t)g.N * sizeof(float), h->stream));
    HIPCHK(h, hipMemsetAsync(a.q2, 0, (size_t)g.N * sizeof(float), h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->have_state = true;
    return MMC_OK;
}

extern "C" int mmc_set_fixed_frame(mmc_handle *h, const float *u0, const float *w0) {
    REQUIRE_HANDLE(h);
    if ((u0 == nullptr) != (w0 == nullptr))
        return fail(h, MMC_ERR_INVALID_ARG, "Give both fix frames u0 and w0.");      // MultiframeMotionCoupling.m:190-192
    if (!u0) { h->g.fixed = 0; return MMC_OK; }
    if (h->g.model != MODEL_IC)
        return fail(h, MMC_ERR_INVALID_ARG, "mmc_set_fixed_frame: the fixed first frame exists only for the infimal-convolution model (kappa not 1, not NaN)");
    int rc = use_device(h); if (rc) return rc;
    const size_t B = (size_t)h->g.N * sizeof(float);
    HIPCHK(h, hipMemcpyAsync(h->a.u0, u0, B, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->a.w0, w0, B, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemsetAsync(h->a.q1, 0, B, h->stream));
    HIPCHK(h, hipMemsetAsync(h->a.q2, 0, B, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->g.fixed = 1;      // only the block that holds frame 0 of the sequence acts on it
    return MMC_OK;
}

extern "C" int mmc_set_flexbox_backward_terms(mmc_handle *h, int enable) {
    REQUIRE_HANDLE(h);
    if (enable && h->g.model != MODEL_IC)
        return fail(h, MMC_ERR_INVALID_ARG, "mmc_set_flexbox_backward_terms: only the infimal-convolution model has w (kappa not 1, not NaN)");
    h->g.wsgn = enable ? 1.f : -1.f;
    return MMC_OK;
}

// One axis of the sampling operator: the interior row shape (taps d, first tap at f*R + od for
// low-res index R) and the rows that deviate from it (near the border, where imresize folds the
// taps beyond the edge onto the edge pixel, helpers/samplingOperator.m:118-121).
struct AxisOp {
    std::vector<int> first;       // banded rows as given
    std::vector<double> w;
    int T = 0;
    std::vector<double> d;        // interior taps
    int od = 0;
    std::vector<char> special;    // row is not the shifted interior row
};

static int analyse_axis(mmc_handle *h, AxisOp &ax, int f, int nout, int nin) {
    const int Rm = nout / 2, T = ax.T;
    int lo = 0, hi = T - 1;
    while (lo < T && ax.w[(size_t)Rm * T + lo] == 0.0) ++lo;
    while (hi >= 0 && ax.w[(size_t)Rm * T + hi] == 0.0) --hi;
    if (hi < lo) return fail(h, MMC_ERR_INVALID_ARG, "sampling taps: the middle row is empty");
    ax.d.assign(ax.w.begin() + (size_t)Rm * T + lo, ax.w.begin() + (size_t)Rm * T + hi + 1);
    ax.od = ax.first[Rm] + lo - f * Rm;
    double dmax = 0.0;
    for (double v : ax.d) dmax = std::max(dmax, std::fabs(v));
    const double tol = 1e-6 * dmax;
    const int nd = (int)ax.d.size();
    ax.special.assign((size_t)nout, 0);
    int nspecial = 0;
    for (int R = 0; R < nout; ++R) {
        const int base = f * R + ax.od;
        bool same = base >= 0 && base + nd <= nin;
        for (int m = 0; same && m < T; ++m) {                 // every given weight is the interior one
            const int k = ax.first[R] + m - base;
            const double want = (k >= 0 && k < nd) ? ax.d[k] : 0.0;
            same = std::fabs(ax.w[(size_t)R * T + m] - want) <= tol;
        }
        for (int k = 0; same && k < nd; ++k) {                // and every interior tap is given
            const int m = base + k - ax.first[R];
            if (ax.d[k] != 0.0 && (m < 0 || m >= T)) same = false;
        }
        ax.special[R] = !same;
        nspecial += !same;
    }
    if (2 * nspecial > nout && nout > 64)      // every such row becomes an explicit sparse row
        return fail(h, MMC_ERR_UNSUPPORTED, "sampling taps: rows are not shifted copies of one another (non-integer zoom?)");
    return MMC_OK;
}

static void free_correction(mmc_handle *h) {
    for (void *p : h->eallocs) (void)hipFree(p);
    h->eallocs.clear();
    Arrays &a = h->a;
    a.ne = a.net = 0;
    a.e_row = a.e_ptr = a.e_col = a.et_row = a.et_ptr = a.et_col = nullptr;
    a.e_val = a.et_val = a.pcorr = a.ucorr = nullptr;
}

template <typename T>
static int upload_vec(mmc_handle *h, T **dst, const std::vector<T> &v) {
    void *p = nullptr;
    HIPCHK(h, hipMalloc(&p, std::max<size_t>(v.size(), 1) * sizeof(T)));
    h->eallocs.push_back(p);
    HIPCHK(h, hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, h->stream));
    *dst = (T *)p;
    return MMC_OK;
}

// A = S*B per frame (MultiframeMotionCoupling.m:368-375): composite kernel of (interior sampling
// taps) x (flipped blur) for the kernels, exact abs row / column sums of A for the step sizes with
// the replicate border of RepConvMtx.m:23-32 merged, and, for the low-res pixels whose sampling
// rows deviate from the interior shape, the sparse difference between the exact row of A and the
// composite kernel applied with a single clamp.
static int build_data_operator(mmc_handle *h) {
    if (h->dataop_built) return MMC_OK;
    Geom &g = h->g;
    const int kh = h->kh, kw = h->kw, f = g.f;
    AxisOp ay, ax;
    if (h->prm.sampling == MMC_SAMPLING_TAPS) {
        if (h->sty <= 0) return fail(h, MMC_ERR_STATE, "mmc_init: no sampling taps (mmc_set_sampling_taps)");
        ay.first = h->sfy; ay.w = h->swy; ay.T = h->sty;
        ax.first = h->sfx; ax.w = h->swx; ax.T = h->stx;
    } else {
        // 'average' at factor 4 = rows [1/4 1/4 1/4 1/4] (samplingOperator.m:70-73); 'stride' = one tap (:137-141)
        const int td = h->prm.sampling == MMC_SAMPLING_AVERAGE ? f : 1;
        for (int axn = 0; axn < 2; ++axn) {
            AxisOp &o = axn ? ax : ay;
            const int nout = axn ? g.nx : g.ny;
            o.T = td;
            o.first.resize((size_t)nout);
            o.w.assign((size_t)nout * td, 1.0 / td);
            for (int R = 0; R < nout; ++R) o.first[R] = f * R;
        }
    }
    int rc;
    if ((rc = analyse_axis(h, ay, f, g.ny, g.Ny))) return rc;
    if ((rc = analyse_axis(h, ax, f, g.nx, g.Nx))) return rc;
    const int tdy = (int)ay.d.size(), tdx = (int)ax.d.size();
    // composite kernel; a first tap below / right of f*R is moved back onto it with zero rows
    const int py = std::max(ay.od - kh / 2, 0), px = std::max(ax.od - kw / 2, 0);
    g.CH = tdy + kh - 1 + py; g.CW = tdx + kw - 1 + px;
    g.oy = ay.od - kh / 2 - py; g.ox = ax.od - kw / 2 - px;
    if (g.CH > MMC_MAX_CK || g.CW > MMC_MAX_CK)
        return fail(h, MMC_ERR_UNSUPPORTED, "sampling taps x blur kernel too large");
    auto kflip = [&](int i, int j) {      // flipped kernel: true convolution, RepConvMtx.m:33
        return h->blur.empty() ? 1.0 : (double)h->blur[(size_t)(kh - 1 - i) + (size_t)kh * (kw - 1 - j)];
    };
    std::vector<double> ck((size_t)g.CH * g.CW, 0.0);
    for (int b = 0; b < tdx; ++b)
        for (int aa = 0; aa < tdy; ++aa)
            for (int j = 0; j < kw; ++j)
                for (int i = 0; i < kh; ++i)
                    ck[(size_t)(py + aa + i) + (size_t)g.CH * (px + b + j)] += ay.d[aa] * ax.d[b] * kflip(i, j);
    std::vector<float> ckf(ck.begin(), ck.end());
    g.Jr = (g.CH + g.f - 1) / g.f; g.Jc = (g.CW + g.f - 1) / g.f;
    g.MP = g.f * g.Jr; g.NP = g.f * g.Jc;
    // Low-res window of p staged per tile by the primal sweep.  Border tiles also evaluate the virtual
    // rows / columns outside the frame that the replicate border folds onto the first and last pixel.
    {
        const int er = std::max(g.f * (g.ny - 1) + g.oy + g.CH - 1 - (g.Ny - 1), 0);
        const int ec = std::max(g.f * (g.nx - 1) + g.ox + g.CW - 1 - (g.Nx - 1), 0);
        g.PR = (TILE_R - 1) / g.f + g.Jr + 1 + (-g.oy + g.f - 1) / g.f + (er + g.f - 1) / g.f;
        g.PC = (TILE_C - 1) / g.f + g.Jc + 1 + (-g.ox + g.f - 1) / g.f + (ec + g.f - 1) / g.f;
    }
    g.dv_PR = make_fastdiv((unsigned)g.PR);
    std::vector<float> ckpad((size_t)g.MP * g.NP, 0.f);
    for (int nn = 0; nn < g.CW; ++nn)
        for (int mm = 0; mm < g.CH; ++mm) ckpad[(size_t)mm + (size_t)g.MP * nn] = ckf[(size_t)mm + (size_t)g.CH * nn];
    HIPCHK(h, hipMemcpyAsync(h->a.ckp, ckpad.data(), ckpad.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
    g.wpad = (4 - ((-g.oy) & 3)) & 3;
    g.MPq = (g.wpad + g.CH + 3) & ~3;
    {
        // Window height: the rows the tile's low-res pixels touch (with a zoom factor that does not divide the
        // tile they reach up to f - 1 rows further), and room for whole rows of the padded kernel table from
        // the last low-res pixel's first row on (the even-zoom path reads them in pairs; the padding is zero).
        const int fe = (TILE_R % g.f == 0 && TILE_C % g.f == 0) ? g.f : 1;
        const int rows = std::max(g.wpad + TILE_R + g.CH - fe, TILE_R - fe + g.MPq);
        g.SRp = (rows + 3) & ~3;
    }
    std::vector<float> ckq((size_t)g.MPq * g.CW, 0.f);
    for (int nn = 0; nn < g.CW; ++nn)
        for (int mm = 0; mm < g.CH; ++mm) ckq[(size_t)(g.wpad + mm) + (size_t)g.MPq * nn] = ckf[(size_t)mm + (size_t)g.CH * nn];
    HIPCHK(h, hipMemcpyAsync(h->a.ckq, ckq.data(), ckq.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
    // Separable composite kernel (the samplers are products of two 1-D factors, samplingOperator.m:146; a
    // Gaussian blur is one as well): ck = ky kx' up to float32 rounding.  Worth two passes only beyond 3 x 3
    // low-res taps at zoom 4, whose one-pass form has its taps unrolled; from MMC_SEP_MIN_TAPS low-res taps on
    // at the other zoom factors.
    g.sep = 0;
    std::vector<float> ksep((size_t)3 * MMC_KSEP_PITCH, 0.f);
    if (MMC_SEPARABLE && g.Jr * g.Jc >= (f == 4 ? MMC_SEP_MIN_TAPS4 : MMC_SEP_MIN_TAPS) && g.SRp <= 128 && g.MP <= MMC_KSEP_PITCH && g.NP <= MMC_KSEP_PITCH && g.MPq <= MMC_KSEP_PITCH) {
        size_t piv = 0;
        for (size_t i = 1; i < ck.size(); ++i) if (std::fabs(ck[i]) > std::fabs(ck[piv])) piv = i;
        const int pm = (int)(piv % (size_t)g.CH), pn = (int)(piv / (size_t)g.CH);
        if (ck[piv] != 0.0) {
            double err = 0.0;
            for (int nn = 0; nn < g.CW; ++nn)
                for (int mm = 0; mm < g.CH; ++mm)
                    err = std::max(err, std::fabs(ck[(size_t)mm + (size_t)g.CH * nn] -
                                                  ck[(size_t)mm + (size_t)g.CH * pn] * ck[(size_t)pm + (size_t)g.CH * nn] / ck[piv]));
            // The blur arrives in float32: an outer product rounded entry by entry deviates from rank one by a few
            // 2^-24 of its entries.  Within 2^-22 of the largest entry the two forms differ by less than the
            // rounding of the sums they feed.
            if (err <= 2.5e-7 * std::fabs(ck[piv])) {
                g.sep = 1;
                for (int mm = 0; mm < g.CH; ++mm) {
                    const float v = (float)ck[(size_t)mm + (size_t)g.CH * pn];
                    ksep[(size_t)mm] = v;
                    ksep[(size_t)2 * MMC_KSEP_PITCH + g.wpad + mm] = v;
                }
                for (int nn = 0; nn < g.CW; ++nn) ksep[(size_t)MMC_KSEP_PITCH + nn] = (float)(ck[(size_t)pm + (size_t)g.CH * nn] / ck[piv]);
            }
        }
    }
    HIPCHK(h, hipMemcpyAsync(h->a.ksep, ksep.data(), ksep.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));

    auto clampi = [](int v, int hi) { return std::min(std::max(v, 0), hi); };
    std::vector<double> colsum((size_t)g.N, 0.0);
    std::vector<float> sig((size_t)g.n);
    std::vector<double> merged((size_t)g.CH * g.CW);
    // correction rows: (low-res pixel, [(high-res pixel, value)])
    std::vector<int> e_row, e_ptr(1, 0), e_col;
    std::vector<float> e_val;
    std::vector<double> patch;
    for (int C = 0; C < g.nx; ++C)
        for (int R = 0; R < g.ny; ++R) {
            const int rbase = g.f * R + g.oy, cbase = g.f * C + g.ox;
            const bool special = ay.special[R] || ax.special[C];
            const bool border = rbase < 0 || cbase < 0 || rbase + g.CH > g.Ny || cbase + g.CW > g.Nx;
            double rowsum = 0.0;
            if (!special && !border) {
                for (int nn = 0; nn < g.CW; ++nn)
                    for (int mm = 0; mm < g.CH; ++mm) {
                        const double v = std::fabs((double)ckf[(size_t)mm + (size_t)g.CH * nn]);
                        rowsum += v;
                        colsum[(size_t)(rbase + mm) + (size_t)g.Ny * (cbase + nn)] += v;
                    }
            } else if (!special) {
                // interior sampling row, blur clamps at the frame border: merge taps that land on one pixel
                std::fill(merged.begin(), merged.end(), 0.0);
                const int rlo = clampi(rbase, g.Ny - 1), clo = clampi(cbase, g.Nx - 1);
                for (int nn = 0; nn < g.CW; ++nn)
                    for (int mm = 0; mm < g.CH; ++mm)
                        merged[(size_t)(clampi(rbase + mm, g.Ny - 1) - rlo) + (size_t)g.CH * (clampi(cbase + nn, g.Nx - 1) - clo)] +=
                            (double)ckf[(size_t)mm + (size_t)g.CH * nn];
                for (int nn = 0; nn < g.CW; ++nn)
                    for (int mm = 0; mm < g.CH; ++mm) {
                        const double v = std::fabs(merged[(size_t)mm + (size_t)g.CH * nn]);
                        if (v == 0.0) continue;
                        rowsum += v;
                        colsum[(size_t)(rlo + mm) + (size_t)g.Ny * (clo + nn)] += v;
                    }
            } else {
                // exact row of S*B for this pixel, and what the kernels compute for it
                const int fy0 = ay.first[R], fx0 = ax.first[C];
                const int r_lo = std::min(clampi(fy0 - kh / 2, g.Ny - 1), clampi(rbase, g.Ny - 1));
                const int r_hi = std::max(clampi(fy0 + ay.T - 1 + kh / 2, g.Ny - 1), clampi(rbase + g.CH - 1, g.Ny - 1));
                const int c_lo = std::min(clampi(fx0 - kw / 2, g.Nx - 1), clampi(cbase, g.Nx - 1));
                const int c_hi = std::max(clampi(fx0 + ax.T - 1 + kw / 2, g.Nx - 1), clampi(cbase + g.CW - 1, g.Nx - 1));
                const int PH = r_hi - r_lo + 1, PW = c_hi - c_lo + 1;
                patch.assign((size_t)2 * PH * PW, 0.0);          // [0]: exact, [1]: composite
                double *ex = patch.data(), *co = patch.data() + (size_t)PH * PW;
                for (int b = 0; b < ax.T; ++b) {
                    const double wx = ax.w[(size_t)C * ax.T + b];
                    if (wx == 0.0) continue;
                    for (int aa = 0; aa < ay.T; ++aa) {
                        const double wy = ay.w[(size_t)R * ay.T + aa];
                        if (wy == 0.0) continue;
                        for (int j = 0; j < kw; ++j)
                            for (int i = 0; i < kh; ++i) {
                                const int rr = clampi(fy0 + aa + i - kh / 2, g.Ny - 1), cc = clampi(fx0 + b + j - kw / 2, g.Nx - 1);
                                ex[(size_t)(rr - r_lo) + (size_t)PH * (cc - c_lo)] += wy * wx * kflip(i, j);
                            }
                    }
                }
                for (int nn = 0; nn < g.CW; ++nn)
                    for (int mm = 0; mm < g.CH; ++mm)
                        co[(size_t)(clampi(rbase + mm, g.Ny - 1) - r_lo) + (size_t)PH * (clampi(cbase + nn, g.Nx - 1) - c_lo)] +=
                            (double)ckf[(size_t)mm + (size_t)g.CH * nn];
                for (int cc = 0; cc < PW; ++cc)
                    for (int rr = 0; rr < PH; ++rr) {
                        const double v = ex[(size_t)rr + (size_t)PH * cc];
                        const double e = v - co[(size_t)rr + (size_t)PH * cc];
                        const size_t hr = (size_t)(r_lo + rr) + (size_t)g.Ny * (c_lo + cc);
                        if (v != 0.0) { rowsum += std::fabs(v); colsum[hr] += std::fabs(v); }
                        if (std::fabs(e) > 1e-9) { e_col.push_back((int)hr); e_val.push_back((float)e); }
                    }
                if ((int)e_col.size() > e_ptr.back()) { e_row.push_back(R + g.ny * C); e_ptr.push_back((int)e_col.size()); }
            }
            sig[(size_t)R + (size_t)g.ny * C] = rowsum > 0.0 ? (float)(1.0 / rowsum) : 1.f;
        }
    std::vector<float> colf(colsum.begin(), colsum.end());
    g.cA_margin = 0;
    if (f == 4 && g.ny >= 2 && g.nx >= 2) {
        const int rc = 4 * (g.ny / 2), cc = 4 * (g.nx / 2);
        float vmax = 0.f;
        for (int k = 0; k < 4; ++k)
            for (int i = 0; i < 4; ++i) {
                g.cA_tab[i + 4 * k] = colf[(size_t)(rc + i) + (size_t)g.Ny * (cc + k)];
                vmax = std::max(vmax, g.cA_tab[i + 4 * k]);
            }
        int margin = 0;
        for (int c = 0; c < g.Nx; ++c)
            for (int r = 0; r < g.Ny; ++r)
                if (std::fabs(colf[(size_t)r + (size_t)g.Ny * c] - g.cA_tab[(r & 3) + 4 * (c & 3)]) > 1e-7f * vmax)
                    margin = std::max(margin, std::min(std::min(r, g.Ny - 1 - r), std::min(c, g.Nx - 1 - c)) + 1);
        margin = (margin + 3) & ~3;
        if (2 * margin + 4 <= g.Ny && 2 * margin + 1 <= g.Nx) g.cA_margin = std::max(margin, 4);
    }
    HIPCHK(h, hipMemcpyAsync(h->a.cker, ckf.data(), ckf.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->a.cA, colf.data(), colf.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->a.sigp, sig.data(), sig.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));

    free_correction(h);
    g.uc_margin = 0;
    if (!e_row.empty()) {
        // transpose: rows = high-res pixels
        std::vector<int> hr_ids(e_col);
        std::sort(hr_ids.begin(), hr_ids.end());
        hr_ids.erase(std::unique(hr_ids.begin(), hr_ids.end()), hr_ids.end());
        std::vector<int> et_ptr(hr_ids.size() + 1, 0), et_col(e_col.size());
        std::vector<float> et_val(e_col.size());
        auto slot = [&](int hr) { return (int)(std::lower_bound(hr_ids.begin(), hr_ids.end(), hr) - hr_ids.begin()); };
        for (int c : e_col) et_ptr[(size_t)slot(c) + 1]++;
        for (size_t i = 1; i < et_ptr.size(); ++i) et_ptr[i] += et_ptr[i - 1];
        std::vector<int> fill(et_ptr.begin(), et_ptr.end() - 1);
        for (size_t r = 0; r < e_row.size(); ++r)
            for (int k = e_ptr[r]; k < e_ptr[r + 1]; ++k) {
                const int dst = fill[(size_t)slot(e_col[k])]++;
                et_col[dst] = e_row[r];
                et_val[dst] = e_val[k];
            }
        Arrays &a = h->a;
        if ((rc = upload_vec(h, &a.e_row, e_row)) || (rc = upload_vec(h, &a.e_ptr, e_ptr)) ||
            (rc = upload_vec(h, &a.e_col, e_col)) || (rc = upload_vec(h, &a.e_val, e_val)) ||
            (rc = upload_vec(h, &a.et_row, hr_ids)) || (rc = upload_vec(h, &a.et_ptr, et_ptr)) ||
            (rc = upload_vec(h, &a.et_col, et_col)) || (rc = upload_vec(h, &a.et_val, et_val)))
            return rc;
        a.ne = (int)e_row.size();
        a.net = (int)hr_ids.size();
        int margin = 0;
        for (int hr : hr_ids) {
            const int c = hr / g.Ny, r = hr - c * g.Ny;
            margin = std::max(margin, std::min(std::min(r, g.Ny - 1 - r), std::min(c, g.Nx - 1 - c)) + 1);
        }
        g.uc_margin = (margin + 3) & ~3;
        void *pc = nullptr, *uc = nullptr;
        HIPCHK(h, hipMalloc(&pc, (size_t)h->L * g.n * sizeof(float)));
        h->eallocs.push_back(pc);
        HIPCHK(h, hipMalloc(&uc, (size_t)h->L * g.N * sizeof(float) + 16));      // + 16: see dev_alloc
        h->eallocs.push_back(uc);
        HIPCHK(h, hipMemsetAsync(pc, 0, (size_t)h->L * g.n * sizeof(float), h->stream));
        HIPCHK(h, hipMemsetAsync(uc, 0, (size_t)h->L * g.N * sizeof(float), h->stream));
        a.pcorr = (float *)pc;
        a.ucorr = (float *)uc;
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    h->dataop_built = true;
    return MMC_OK;
}

const Geom *mmc_internal_geom(const mmc_handle *h) { return h ? &h->g : nullptr; }
const Arrays *mmc_internal_arrays(const mmc_handle *h) { return h ? &h->a : nullptr; }
int mmc_internal_build_data_operator(mmc_handle *h) {
    REQUIRE_HANDLE(h);
    int rc = use_device(h); if (rc) return rc;
    return build_data_operator(h);
}

static int read_sums(mmc_handle *h, int first, int count, double *out) {
    std::vector<double> host((size_t)MMC_NUM_SUMS_INT * MMC_SUM_SLOTS);
    HIPCHK(h, hipMemcpyAsync(host.data(), h->a.sums, host.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < count; ++i) {
        double s = 0.0;
        for (int k = 0; k < MMC_SUM_SLOTS; ++k) s += host[(size_t)(first + i) * MMC_SUM_SLOTS + k];
        out[i] = s;
    }
    return MMC_OK;
}

static int zero_sums(mmc_handle *h) {
    HIPCHK(h, hipMemsetAsync(h->a.sums, 0, (size_t)MMC_NUM_SUMS_INT * MMC_SUM_SLOTS * sizeof(double), h->stream));
    return MMC_OK;
}

extern "C" int mmc_h_partials(mmc_handle *h, double out[2]) {
    REQUIRE_HANDLE(h);
    if (!out) return fail(h, MMC_ERR_INVALID_ARG, "mmc_h_partials: null output");
    if (!h->have_state || !h->have_flow) return fail(h, MMC_ERR_STATE, "mmc_h_partials: set the flow and the start point first");
    int rc = use_device(h); if (rc) return rc;
    if ((rc = zero_sums(h))) return rc;
    HIPCHK(h, mmc_launch_h_partials(h->g, h->a, h->stream));
    return read_sums(h, SUM_H, 2, out);
}

extern "C" int mmc_set_h(mmc_handle *h, float value) {
    REQUIRE_HANDLE(h);
    if (!(value > 0.f) || !std::isfinite(value)) return fail(h, MMC_ERR_INVALID_ARG, "mmc_set_h: h must be positive and finite");
    h->h = value; h->have_h = true; h->g.inv_h = 1.f / value;
    return MMC_OK;
}

extern "C" int mmc_get_h(const mmc_handle *h, float *value) {
    if (!h || !value) return MMC_ERR_INVALID_ARG;
    *value = h->h;
    return MMC_OK;
}

extern "C" int mmc_init(mmc_handle *h) {
    REQUIRE_HANDLE(h);
    if (!h->have_f) return fail(h, MMC_ERR_STATE, "mmc_init: no low-resolution data (mmc_set_lowres)");
    if (!h->have_flow && h->npairs > 0) return fail(h, MMC_ERR_STATE, "mmc_init: no flow field (mmc_set_flow)");
    if (!h->have_state) return fail(h, MMC_ERR_STATE, "mmc_init: no start point (mmc_set_state)");
    int rc = use_device(h); if (rc) return rc;
    if ((rc = build_data_operator(h))) return rc;
    const Geom &g = h->g;
    Arrays &a = h->a;
    {
        std::vector<int> bb((size_t)(h->L + 1) * g.ntiles * 4);
        for (size_t i = 0; i < bb.size(); i += 4) { bb[i] = INT_MAX; bb[i + 1] = -1; bb[i + 2] = INT_MAX; bb[i + 3] = -1; }
        HIPCHK(h, hipMemcpyAsync(a.bbox, bb.data(), bb.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    const size_t nslots = (size_t)(h->L + 1) * g.ntiles;
    HIPCHK(h, hipMemsetAsync(a.bstat, 0, nslots * 3 * sizeof(unsigned long long), h->stream));
    HIPCHK(h, mmc_launch_init_warp(g, a, h->scratch, h->stream));
    {
        // sources outside their tile's rectangle: count, prefix sum on the host, then place
        std::vector<int> cnt(nslots), ptr(nslots + 1, 0);
        HIPCHK(h, hipMemsetAsync(h->ovf_cursor, 0, nslots * sizeof(int), h->stream));
        HIPCHK(h, mmc_launch_init_overflow(g, a, false, h->ovf_cursor, h->stream));
        HIPCHK(h, hipMemcpyAsync(cnt.data(), h->ovf_cursor, nslots * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        long long total = 0;
        for (size_t i = 0; i < nslots; ++i) { ptr[i] = (int)total; total += cnt[i]; }
        if (total > (long long)INT_MAX) return fail(h, MMC_ERR_UNSUPPORTED, "mmc_init: flow too irregular (overflow lists exceed 2^31 entries)");
        ptr[nslots] = (int)total;
        if ((size_t)total > h->ovf_capacity || !a.ovf_idx) {
            if (a.ovf_idx) { HIPCHK(h, hipFree(a.ovf_idx)); a.ovf_idx = nullptr; h->ovf_capacity = 0; }
            const size_t cap = std::max<size_t>((size_t)total + (size_t)total / 4, 1024);
            HIPCHK(h, hipMalloc((void **)&a.ovf_idx, cap * sizeof(int)));
            h->ovf_capacity = cap;
        }
        HIPCHK(h, hipMemcpyAsync(a.ovf_ptr, ptr.data(), (nslots + 1) * sizeof(int), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemsetAsync(h->ovf_cursor, 0, nslots * sizeof(int), h->stream));
        HIPCHK(h, mmc_launch_init_overflow(g, a, true, h->ovf_cursor, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        h->ovf_total = (size_t)total;
    }
    HIPCHK(h, mmc_launch_sanitize(g, a, h->stream));
    if (!h->have_h && !h->has_prev && !h->has_next) {
        // MultiframeMotionCoupling.m:431-437.  A block with neighbours holds only part of the two
        // sums (a trailing one-frame block owns no pair at all): there h stays unset until the
        // caller has added mmc_h_partials up over all blocks and called mmc_set_h.
        double s[2];
        if ((rc = mmc_h_partials(h, s))) return rc;
        if (g.nc_global == 1) {
            h->h = 1.f;
        } else {
            if (!(s[0] > 0.0) || !(s[1] > 0.0))
                return fail(h, MMC_ERR_INVALID_ARG,
                            "mmc_init: warp or gradient energy of the start point is zero, h = warpPix/gradPix is undefined; pass h explicitly");
            h->h = (float)(s[0] / s[1]);
        }
        h->have_h = true;
        h->g.inv_h = 1.f / h->h;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->inited = true;
    return MMC_OK;
}

static int check_range(mmc_handle *h, int t0, int t1) {
    if (!h->inited) return fail(h, MMC_ERR_STATE, "No initialization found, run mmc_init first.");
    if (!h->have_h)
        return fail(h, MMC_ERR_STATE, "sweep: h is not set; add mmc_h_partials up over all blocks of the sequence and call mmc_set_h");
    if (t0 < 0 || t1 > h->L || t1 < t0) return fail(h, MMC_ERR_INVALID_ARG, "sweep: frame range outside the block");
    return MMC_OK;
}

enum { SWEEP_PROF_MAX = 4096 };

// Event pair around one sweep call when profiling is on (bounded pool, first calls only).
static int sweep_prof_begin(mmc_handle *h, int kind, hipEvent_t **ev) {
    *ev = nullptr;
    if (!h->profiling || h->sweep_used >= SWEEP_PROF_MAX) return MMC_OK;
    if (h->sweep_events.size() < 2 * (h->sweep_used + 1)) {
        hipEvent_t e0 = mmc_pool_event_acquire(h->prm.device, true), e1 = mmc_pool_event_acquire(h->prm.device, true);
        if (!e0 || !e1) {
            mmc_pool_event_release(h->prm.device, true, e0);
            mmc_pool_event_release(h->prm.device, true, e1);
            return fail(h, MMC_ERR_HIP, "sweep profiling: hipEventCreate failed");
        }
        h->sweep_events.push_back(e0);
        h->sweep_events.push_back(e1);
        h->sweep_kinds.push_back(kind);
    }
    h->sweep_kinds[h->sweep_used] = kind;
    *ev = &h->sweep_events[2 * h->sweep_used];
    HIPCHK(h, hipEventRecord((*ev)[0], h->stream));
    return MMC_OK;
}

static int sweep_prof_end(mmc_handle *h, hipEvent_t *ev) {
    if (!ev) return MMC_OK;
    HIPCHK(h, hipEventRecord(ev[1], h->stream));
    h->sweep_used++;
    return MMC_OK;
}

extern "C" int mmc_sweep_primal(mmc_handle *h, int t0, int t1, int reduce) {
    REQUIRE_HANDLE(h);
    int rc = check_range(h, t0, t1); if (rc) return rc;
    if ((rc = use_device(h))) return rc;
    if (t1 == t0) return MMC_OK;
    hipEvent_t *ev;
    if ((rc = sweep_prof_begin(h, 0, &ev))) return rc;
    HIPCHK(h, mmc_launch_primal(h->g, h->a, t0, t1, MODE_UPDATE, reduce != 0, h->stream));
    return sweep_prof_end(h, ev);
}

extern "C" int mmc_sweep_dual(mmc_handle *h, int t0, int t1, int reduce) {
    REQUIRE_HANDLE(h);
    int rc = check_range(h, t0, t1); if (rc) return rc;
    if ((rc = use_device(h))) return rc;
    if (t1 == t0) return MMC_OK;
    hipEvent_t *ev;
    if ((rc = sweep_prof_begin(h, 1, &ev))) return rc;
    HIPCHK(h, mmc_launch_dual(h->g, h->a, t0, t1, MODE_UPDATE, reduce != 0, h->stream));
    return sweep_prof_end(h, ev);
}

extern "C" int mmc_get_sweep_profile(mmc_handle *h, double total_ms[2], int calls[2]) {
    REQUIRE_HANDLE(h);
    if (!total_ms || !calls) return fail(h, MMC_ERR_INVALID_ARG, "mmc_get_sweep_profile: null output");
    int rc = use_device(h); if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    total_ms[0] = total_ms[1] = 0.0;
    calls[0] = calls[1] = 0;
    for (size_t i = 0; i < h->sweep_used; ++i) {
        float ms = 0.f;
        HIPCHK(h, hipEventElapsedTime(&ms, h->sweep_events[2 * i], h->sweep_events[2 * i + 1]));
        total_ms[h->sweep_kinds[i]] += ms;
        calls[h->sweep_kinds[i]]++;
    }
    return MMC_OK;
}

extern "C" int mmc_sums_reset(mmc_handle *h) {
    REQUIRE_HANDLE(h);
    int rc = use_device(h); if (rc) return rc;
    return zero_sums(h);
}

extern "C" int mmc_sums_read(mmc_handle *h, double out[MMC_NUM_SUMS]) {
    REQUIRE_HANDLE(h);
    if (!out) return fail(h, MMC_ERR_INVALID_ARG, "mmc_sums_read: null output");
    int rc = use_device(h); if (rc) return rc;
    return read_sums(h, 0, MMC_NUM_SUMS, out);
}

extern "C" int mmc_sync(mmc_handle *h) {
    REQUIRE_HANDLE(h);
    int rc = use_device(h); if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MMC_OK;
}

extern "C" int mmc_halo_info(mmc_handle *h, mmc_halo *out) {
    REQUIRE_HANDLE(h);
    if (!out) return fail(h, MMC_ERR_INVALID_ARG, "mmc_halo_info: null output");
    const Geom &g = h->g;
    const Arrays &a = h->a;
    memset(out, 0, sizeof(*out));
    out->frame_bytes = (uint64_t)g.N * sizeof(float);
    const bool ic = g.model == MODEL_IC;
    const size_t N = (size_t)g.N;
    if (h->has_prev) {
        out->send_prev[0] = a.ub + N;                       // first local frame
        out->send_prev[1] = ic ? a.wb + N : nullptr;
        out->recv_prev[0] = a.g1t;                          // ghost slot of pair -1
        out->recv_prev[1] = ic ? a.g2t : nullptr;
    }
    if (h->has_next) {
        out->recv_next[0] = a.ub + N * (size_t)(h->L + 1);  // ghost slot of frame L
        out->recv_next[1] = ic ? a.wb + N * (size_t)(h->L + 1) : nullptr;
        out->send_next[0] = a.g1t + N * (size_t)h->L;       // last local pair
        out->send_next[1] = ic ? a.g2t + N * (size_t)h->L : nullptr;
    }
    return MMC_OK;
}

// ---- step balance -------------------------------------------------------------------------------
enum { STEP_ADAPT_EVERY = 50 };

extern "C" int mmc_set_step_balance(mmc_handle *h, float s) {
    REQUIRE_HANDLE(h);
    if (!(s > 0.f) || !std::isfinite(s)) return fail(h, MMC_ERR_INVALID_ARG, "mmc_set_step_balance: the factor must be positive and finite");
    h->balance = (double)s;
    h->g.tau_s = s;
    h->g.sigma_s = 1.f / s;
    return MMC_OK;
}

extern "C" int mmc_get_step_balance(const mmc_handle *h, float *s) {
    if (!h || !s) return MMC_ERR_INVALID_ARG;
    *s = (float)h->balance;
    return MMC_OK;
}

extern "C" int mmc_set_step_adaptation(mmc_handle *h, int enable) {
    REQUIRE_HANDLE(h);
    h->adapt = enable != 0;
    h->adapt_alpha = 0.5;
    return MMC_OK;
}

extern "C" int mmc_step_balance_rule(const double sums[2], double *balance, double *alpha) {
    if (!sums || !balance || !alpha) return 0;
    const double Delta = 1.5, eta = 0.8;
    const double P = std::sqrt(std::max(sums[0], 0.0)), D = std::sqrt(std::max(sums[1], 0.0));
    if (!(P > 0.0) || !(D > 0.0) || !std::isfinite(P) || !std::isfinite(D)) return 0;
    double s = *balance;
    if (P > Delta * D) s /= (1.0 - *alpha);
    else if (P < D / Delta) s *= (1.0 - *alpha);
    else return 0;
    *balance = std::min(std::max(s, 1e-4), 1e4);
    *alpha *= eta;
    return 1;
}

// ---- diagnostics: primal and dual objective, gap, optimality residuals -------------------------

extern "C" int mmc_set_diagnostics(mmc_handle *h, int mode) {
    REQUIRE_HANDLE(h);
    if (mode < MMC_DIAG_OFF || mode > MMC_DIAG_ALWAYS) return fail(h, MMC_ERR_INVALID_ARG, "mmc_set_diagnostics: unknown mode");
    h->diagnostics = mode;
    return MMC_OK;
}

extern "C" int mmc_problem_size(const mmc_handle *h, double *n_primal, double *n_dual) {
    if (!h) return MMC_ERR_INVALID_ARG;
    const Geom &g = h->g;
    const double M = (double)h->L * g.N, m = (double)h->L * g.n;
    if (n_primal) *n_primal = g.model == MODEL_IC ? 2.0 * M : M;
    if (n_dual) *n_dual = m + (g.model == MODEL_IC ? 6.0 * M : 3.0 * M) +
                          ((g.model == MODEL_IC && g.fixed && g.frame_begin == 0) ? 2.0 * g.N : 0.0);
    return MMC_OK;
}

extern "C" int mmc_state_halo_info(mmc_handle *h, mmc_halo *out) {
    REQUIRE_HANDLE(h);
    if (!out) return fail(h, MMC_ERR_INVALID_ARG, "mmc_state_halo_info: null output");
    const Geom &g = h->g;
    const Arrays &a = h->a;
    memset(out, 0, sizeof(*out));
    out->frame_bytes = (uint64_t)g.N * sizeof(float);
    const bool ic = g.model == MODEL_IC;
    const size_t N = (size_t)g.N;
    if (h->has_prev) {
        out->send_prev[0] = a.u + N;
        out->send_prev[1] = ic ? a.w + N : nullptr;
    }
    if (h->has_next) {
        out->recv_next[0] = a.u + N * (size_t)(h->L + 1);
        out->recv_next[1] = ic ? a.w + N * (size_t)(h->L + 1) : nullptr;
    }
    return MMC_OK;
}

extern "C" int mmc_diag_partials(mmc_handle *h, double out[MMC_NUM_DIAG]) {
    REQUIRE_HANDLE(h);
    if (!out) return fail(h, MMC_ERR_INVALID_ARG, "mmc_diag_partials: null output");
    int rc = check_range(h, 0, h->L); if (rc) return rc;
    if ((rc = use_device(h))) return rc;
    if ((rc = zero_sums(h))) return rc;
    Arrays e = h->a;
    e.ub = h->a.u; e.wb = h->a.w;     // K x at the iterate, not at xbar
    HIPCHK(h, mmc_launch_dual(h->g, e, 0, h->L, MODE_ENERGY, false, h->stream));
    HIPCHK(h, mmc_launch_primal(h->g, h->a, 0, h->L, MODE_DUALOBJ, false, h->stream));
    HIPCHK(h, mmc_launch_dual_linear(h->g, h->a, h->stream));
    if ((rc = read_sums(h, SUM_ENERGY, 4, out))) return rc;
    if ((rc = read_sums(h, SUM_DUALOBJ, 5, out + 4))) return rc;
    return MMC_OK;
}

static void clear_diag(mmc_stats *st) {
    const double nan = std::nan("");
    st->primal_energy = st->dual_energy = st->gap = st->dual_infeasibility = nan;
    st->opt_res_primal = st->opt_res_dual = st->norm_KTy = st->norm_Kx = nan;
}

extern "C" int mmc_stats_from_diag(const double d[MMC_NUM_DIAG], double n_primal, double n_dual, float tol, mmc_stats *st) {
    if (!d || !st) return 0;
    st->primal_energy = d[0] + d[1];
    st->dual_energy = -d[8] - d[4];
    st->gap = st->primal_energy - st->dual_energy;
    st->dual_infeasibility = d[5];
    st->norm_Kx = std::sqrt(std::max(d[2], 0.0));
    st->opt_res_dual = std::sqrt(std::max(d[3], 0.0));
    st->norm_KTy = std::sqrt(std::max(d[6], 0.0));
    st->opt_res_primal = std::sqrt(std::max(d[7], 0.0));
    if (!(tol > 0.f)) return 0;
    const double t = (double)tol;
    return st->opt_res_primal <= t * std::sqrt(std::max(n_primal, 0.0)) + t * st->norm_KTy &&
           st->opt_res_dual <= t * std::sqrt(std::max(n_dual, 0.0)) + t * st->norm_Kx;
}

static bool diag_wanted(const mmc_handle *h, float tol, mmc_progress_cb cb) {
    return h->diagnostics == MMC_DIAG_ALWAYS || (h->diagnostics == MMC_DIAG_AUTO && (tol > 0.f || cb));
}

extern "C" int mmc_step_sums_read(mmc_handle *h, double out[2]) {
    REQUIRE_HANDLE(h);
    if (!out) return fail(h, MMC_ERR_INVALID_ARG, "mmc_step_sums_read: null output");
    int rc = use_device(h); if (rc) return rc;
    return read_sums(h, SUM_WRES, 2, out);
}

static void fill_stats(mmc_stats *st, const double s[MMC_NUM_SUMS]) {
    const double tiny = 1e-30;
    st->res_primal = std::sqrt(s[0]) / std::max(std::sqrt(s[1]), tiny);
    st->res_dual = std::sqrt(s[2]) / std::max(std::sqrt(s[3]), tiny);
}

extern "C" int mmc_solve(mmc_handle *h, int max_iters, float tol, int check_every,
                         mmc_stats *stats, mmc_progress_cb cb, void *user) {
    REQUIRE_HANDLE(h);
    if (!h->inited) return fail(h, MMC_ERR_STATE, "No initialization found, run mmc_init first.");
    if (h->has_prev || h->has_next)
        return fail(h, MMC_ERR_STATE, "mmc_solve: this block has neighbours; drive it with mmc_sweep_* and the halo exchange");
    if (max_iters < 0) return fail(h, MMC_ERR_INVALID_ARG, "mmc_solve: negative iteration count");
    if (check_every <= 0) check_every = max_iters > 0 ? max_iters : 1;
    int rc = use_device(h); if (rc) return rc;
    mmc_stats st;
    memset(&st, 0, sizeof(st));
    clear_diag(&st);
    const bool diag = diag_wanted(h, tol, cb);
    double np = 0.0, nd = 0.0;
    mmc_problem_size(h, &np, &nd);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const auto t_start = std::chrono::steady_clock::now();
    const Geom &g = h->g;
    int prof_n = 0;
    if (h->profiling) {
        // one event between consecutive sweeps: e[0] P e[1] D e[2] P e[3] D ...
        const size_t need = (size_t)2 * std::min(max_iters, (int)PROF_MAX_ITERS) + 1;
        while (h->events.size() < need) {
            hipEvent_t e = mmc_pool_event_acquire(h->prm.device, true);
            if (!e) return fail(h, MMC_ERR_HIP, "mmc_solve: hipEventCreate failed");
            h->events.push_back(e);
        }
    }
    st.step_balance = h->balance;
    for (int it = 0; it < max_iters; ++it) {
        const bool check = ((it + 1) % check_every == 0) || it == max_iters - 1;
        const bool adapt_now = h->adapt && (it + 1) % STEP_ADAPT_EVERY == 0;
        const bool reduce = check || adapt_now;
        if (reduce && (rc = zero_sums(h))) return rc;
        const bool prof = h->profiling && it < PROF_MAX_ITERS;
        hipEvent_t *ev = prof ? &h->events[(size_t)2 * it] : nullptr;
        if (prof && it == 0) HIPCHK(h, hipEventRecord(ev[0], h->stream));
        HIPCHK(h, mmc_launch_primal(g, h->a, 0, h->L, MODE_UPDATE, reduce, h->stream));
        if (prof) HIPCHK(h, hipEventRecord(ev[1], h->stream));
        HIPCHK(h, mmc_launch_dual(g, h->a, 0, h->L, MODE_UPDATE, reduce, h->stream));
        if (prof) { HIPCHK(h, hipEventRecord(ev[2], h->stream)); prof_n = it + 1; }
        st.iterations = it + 1;
        if (adapt_now) {
            double ws[2];
            if ((rc = read_sums(h, SUM_WRES, 2, ws))) return rc;
            if (mmc_step_balance_rule(ws, &h->balance, &h->adapt_alpha)) {
                h->g.tau_s = (float)h->balance;
                h->g.sigma_s = (float)(1.0 / h->balance);
            }
            st.step_balance = h->balance;
        }
        if (check) {
            double s[MMC_NUM_SUMS];
            if ((rc = read_sums(h, 0, MMC_NUM_SUMS, s))) return rc;
            fill_stats(&st, s);
            bool stop = tol > 0.f && st.res_primal < tol && st.res_dual < tol;
            if (diag) {
                double d[MMC_NUM_DIAG];
                if ((rc = mmc_diag_partials(h, d))) return rc;
                stop = mmc_stats_from_diag(d, np, nd, tol, &st) != 0;
            }
            st.solve_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
            if (cb && cb(st.iterations, &st, user)) break;
            if (stop) { st.converged = 1; break; }
        }
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    st.solve_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    if (h->profiling) {
        prof_n = std::min(prof_n, st.iterations);
        double acc[3] = {0, 0, 0};      // primal sweep, dual sweep, (unused)
        for (int it = 0; it < prof_n; ++it) {
            hipEvent_t *ev = &h->events[(size_t)2 * it];
            for (int k = 0; k < 2; ++k) {
                float ms = 0.f;
                HIPCHK(h, hipEventElapsedTime(&ms, ev[k], ev[k + 1]));
                acc[k] += ms;
            }
        }
        for (int k = 0; k < 3; ++k) h->prof_ms[k] = prof_n ? acc[k] / prof_n : 0.0;
        h->prof_iters = prof_n;
    }
    if (stats) *stats = st;
    return MMC_OK;
}

extern "C" int mmc_set_profiling(mmc_handle *h, int enable) {
    REQUIRE_HANDLE(h);
    h->profiling = enable != 0;
    h->sweep_used = 0;
    return MMC_OK;
}

extern "C" int mmc_get_profile(mmc_handle *h, double out_ms[3], int *iterations) {
    REQUIRE_HANDLE(h);
    if (!out_ms) return fail(h, MMC_ERR_INVALID_ARG, "mmc_get_profile: null output");
    for (int k = 0; k < 3; ++k) out_ms[k] = h->prof_ms[k];
    if (iterations) *iterations = h->prof_iters;
    return MMC_OK;
}

extern "C" int mmc_get_state(mmc_handle *h, float *u, float *w) {
    REQUIRE_HANDLE(h);
    if (!h->have_state) return fail(h, MMC_ERR_STATE, "mmc_get_state: no state");
    int rc = use_device(h); if (rc) return rc;
    const Geom &g = h->g;
    const size_t nloc = (size_t)h->L * g.N * sizeof(float);
    if (u) HIPCHK(h, hipMemcpyAsync(u, h->a.u + g.N, nloc, hipMemcpyDeviceToHost, h->stream));
    if (w) {
        if (g.model != MODEL_IC) return fail(h, MMC_ERR_INVALID_ARG, "mmc_get_state: this model has no w");
        HIPCHK(h, hipMemcpyAsync(w, h->a.w + g.N, nloc, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MMC_OK;
}

extern "C" int mmc_energy(mmc_handle *h, double *value) {
    REQUIRE_HANDLE(h);
    if (!value) return fail(h, MMC_ERR_INVALID_ARG, "mmc_energy: null output");
    if (!h->inited) return fail(h, MMC_ERR_STATE, "No initialization found, run mmc_init first.");
    if (h->has_next) return fail(h, MMC_ERR_UNSUPPORTED, "mmc_energy: only for a block without a next neighbour");
    int rc = use_device(h); if (rc) return rc;
    if ((rc = zero_sums(h))) return rc;
    Arrays e = h->a;
    e.ub = h->a.u; e.wb = h->a.w;     // evaluate at the iterate, not at xbar
    HIPCHK(h, mmc_launch_dual(h->g, e, 0, h->L, MODE_ENERGY, false, h->stream));
    double s[2];
    if ((rc = read_sums(h, SUM_ENERGY, 2, s))) return rc;
    *value = s[0] + s[1];
    return MMC_OK;
}

// ---------------------------------------------------------------------------
// test hooks
// ---------------------------------------------------------------------------

static int zero_duals(mmc_handle *h) {
    const Geom &g = h->g;
    Arrays &a = h->a;
    const size_t fs = frame_slots(h) * (size_t)g.N * sizeof(float);
    float *duals[] = {a.g1t, a.g1x, a.g1y, a.g2t, a.g2x, a.g2y};
    for (float *d : duals) HIPCHK(h, hipMemsetAsync(d, 0, fs, h->stream));
    HIPCHK(h, hipMemsetAsync(a.p, 0, (size_t)h->L * g.n * sizeof(float), h->stream));
    HIPCHK(h, hipMemsetAsync(a.q1, 0, (size_t)g.N * sizeof(float), h->stream));
    HIPCHK(h, hipMemsetAsync(a.q2, 0, (size_t)g.N * sizeof(float), h->stream));
    return MMC_OK;
}

static int restore_xbar(mmc_handle *h) {
    const Geom &g = h->g;
    Arrays &a = h->a;
    const size_t nloc = (size_t)h->L * g.N * sizeof(float);
    HIPCHK(h, hipMemcpyAsync(a.ub + g.N, a.u + g.N, nloc, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(a.wb + g.N, a.w + g.N, nloc, hipMemcpyDeviceToDevice, h->stream));
    return MMC_OK;
}

static int download_dual(mmc_handle *h, float *out_g1, float *out_g2) {
    const Geom &g = h->g;
    const Arrays &a = h->a;
    const size_t M = (size_t)h->L * g.N, B = M * sizeof(float);
    const size_t o = (size_t)g.N;
    if (g.model == MODEL_ADD) {
        if (out_g1) HIPCHK(h, hipMemcpyAsync(out_g1, a.g1t + o, B, hipMemcpyDeviceToHost, h->stream));
        if (out_g2) {
            HIPCHK(h, hipMemcpyAsync(out_g2, a.g1x + o, B, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipMemcpyAsync(out_g2 + M, a.g1y + o, B, hipMemcpyDeviceToHost, h->stream));
        }
    } else {
        if (out_g1) {
            HIPCHK(h, hipMemcpyAsync(out_g1, a.g1t + o, B, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipMemcpyAsync(out_g1 + M, a.g1x + o, B, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipMemcpyAsync(out_g1 + 2 * M, a.g1y + o, B, hipMemcpyDeviceToHost, h->stream));
        }
        if (out_g2 && g.model == MODEL_IC) {
            HIPCHK(h, hipMemcpyAsync(out_g2, a.g2t + o, B, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipMemcpyAsync(out_g2 + M, a.g2x + o, B, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipMemcpyAsync(out_g2 + 2 * M, a.g2y + o, B, hipMemcpyDeviceToHost, h->stream));
        }
    }
    return MMC_OK;
}

extern "C" int mmc_debug_apply_K(mmc_handle *h, const float *u, const float *w,
                                 float *out_p, float *out_g1, float *out_g2) {
    REQUIRE_HANDLE(h);
    if (!h->inited) return fail(h, MMC_ERR_STATE, "No initialization found, run mmc_init first.");
    if (!u) return fail(h, MMC_ERR_INVALID_ARG, "mmc_debug_apply_K: null u");
    int rc = use_device(h); if (rc) return rc;
    const Geom &g = h->g;
    Arrays &a = h->a;
    const size_t nall = (size_t)(h->L + h->has_next) * g.N * sizeof(float);
    HIPCHK(h, hipMemcpyAsync(a.ub + g.N, u, nall, hipMemcpyHostToDevice, h->stream));
    if (g.model == MODEL_IC) {
        if (!w) return fail(h, MMC_ERR_INVALID_ARG, "mmc_debug_apply_K: this model needs w");
        HIPCHK(h, hipMemcpyAsync(a.wb + g.N, w, nall, hipMemcpyHostToDevice, h->stream));
    }
    HIPCHK(h, mmc_launch_dual(g, a, 0, h->L, MODE_RAW, false, h->stream));
    if (out_p) HIPCHK(h, hipMemcpyAsync(out_p, a.p, (size_t)h->L * g.n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    if ((rc = download_dual(h, out_g1, out_g2))) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if ((rc = zero_duals(h))) return rc;
    if ((rc = restore_xbar(h))) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MMC_OK;
}

extern "C" int mmc_debug_apply_KT(mmc_handle *h, const float *p, const float *g1, const float *g2,
                                  const float *ghost_g1t, const float *ghost_g2t,
                                  float *out_u, float *out_w) {
    REQUIRE_HANDLE(h);
    if (!h->inited) return fail(h, MMC_ERR_STATE, "No initialization found, run mmc_init first.");
    if (!p || !g1) return fail(h, MMC_ERR_INVALID_ARG, "mmc_debug_apply_KT: null input");
    int rc = use_device(h); if (rc) return rc;
    const Geom &g = h->g;
    Arrays &a = h->a;
    const size_t M = (size_t)h->L * g.N, B = M * sizeof(float), o = (size_t)g.N;
    if ((rc = zero_duals(h))) return rc;
    HIPCHK(h, hipMemcpyAsync(a.p, p, (size_t)h->L * g.n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(a.g1t + o, g1, B, hipMemcpyHostToDevice, h->stream));
    if (g.model == MODEL_ADD) {
        if (!g2) return fail(h, MMC_ERR_INVALID_ARG, "mmc_debug_apply_KT: null g2");
        HIPCHK(h, hipMemcpyAsync(a.g1x + o, g2, B, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(a.g1y + o, g2 + M, B, hipMemcpyHostToDevice, h->stream));
    } else {
        HIPCHK(h, hipMemcpyAsync(a.g1x + o, g1 + M, B, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(a.g1y + o, g1 + 2 * M, B, hipMemcpyHostToDevice, h->stream));
        if (g.model == MODEL_IC) {
            if (!g2) return fail(h, MMC_ERR_INVALID_ARG, "mmc_debug_apply_KT: null g2");
            HIPCHK(h, hipMemcpyAsync(a.g2t + o, g2, B, hipMemcpyHostToDevice, h->stream));
            HIPCHK(h, hipMemcpyAsync(a.g2x + o, g2 + M, B, hipMemcpyHostToDevice, h->stream));
            HIPCHK(h, hipMemcpyAsync(a.g2y + o, g2 + 2 * M, B, hipMemcpyHostToDevice, h->stream));
        }
    }
    if (h->has_prev && ghost_g1t) HIPCHK(h, hipMemcpyAsync(a.g1t, ghost_g1t, o * sizeof(float), hipMemcpyHostToDevice, h->stream));
    if (h->has_prev && ghost_g2t && g.model == MODEL_IC)
        HIPCHK(h, hipMemcpyAsync(a.g2t, ghost_g2t, o * sizeof(float), hipMemcpyHostToDevice, h->stream));
    // the sweeps keep temporal duals of masked rows at zero; K' relies on it
    HIPCHK(h, mmc_launch_sanitize(g, a, h->stream));
    HIPCHK(h, mmc_launch_primal(g, a, 0, h->L, MODE_RAW, false, h->stream));
    if (out_u) HIPCHK(h, hipMemcpyAsync(out_u, a.ub + o, B, hipMemcpyDeviceToHost, h->stream));
    if (out_w && g.model == MODEL_IC) HIPCHK(h, hipMemcpyAsync(out_w, a.wb + o, B, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if ((rc = zero_duals(h))) return rc;
    if ((rc = restore_xbar(h))) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MMC_OK;
}

static int debug_steps(mmc_handle *h, int primal_mode, float *tau_u, float *tau_w,
                       float *sigma_p, float *sigma_g1, float *sigma_g2);

extern "C" int mmc_debug_steps(mmc_handle *h, float *tau_u, float *tau_w,
                               float *sigma_p, float *sigma_g1, float *sigma_g2) {
    return debug_steps(h, MODE_STEPS, tau_u, tau_w, sigma_p, sigma_g1, sigma_g2);
}

extern "C" int mmc_debug_steps_update(mmc_handle *h, float *tau_u, float *tau_w) {
    return debug_steps(h, MODE_STEPS_UPD, tau_u, tau_w, nullptr, nullptr, nullptr);
}

static int debug_steps(mmc_handle *h, int primal_mode, float *tau_u, float *tau_w,
                       float *sigma_p, float *sigma_g1, float *sigma_g2) {
    REQUIRE_HANDLE(h);
    if (!h->inited) return fail(h, MMC_ERR_STATE, "No initialization found, run mmc_init first.");
    int rc = use_device(h); if (rc) return rc;
    const Geom &g = h->g;
    Arrays &a = h->a;
    const size_t B = (size_t)h->L * g.N * sizeof(float), o = (size_t)g.N;
    HIPCHK(h, mmc_launch_primal(g, a, 0, h->L, primal_mode, false, h->stream));
    HIPCHK(h, mmc_launch_dual(g, a, 0, h->L, MODE_STEPS, false, h->stream));
    if (tau_u) HIPCHK(h, hipMemcpyAsync(tau_u, a.ub + o, B, hipMemcpyDeviceToHost, h->stream));
    if (tau_w && g.model == MODEL_IC) HIPCHK(h, hipMemcpyAsync(tau_w, a.wb + o, B, hipMemcpyDeviceToHost, h->stream));
    if (sigma_p) HIPCHK(h, hipMemcpyAsync(sigma_p, a.p, (size_t)h->L * g.n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    if (sigma_g1) HIPCHK(h, hipMemcpyAsync(sigma_g1, a.g1t + o, B, hipMemcpyDeviceToHost, h->stream));
    if (sigma_g2) HIPCHK(h, hipMemcpyAsync(sigma_g2, a.g2t + o, B, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if ((rc = zero_duals(h))) return rc;
    if ((rc = restore_xbar(h))) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MMC_OK;
}
