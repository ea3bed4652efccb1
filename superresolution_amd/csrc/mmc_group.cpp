// mmc_group_*: one sequence solved by several frame blocks from ONE host thread, each block on a
// device of its own (or several on one device).  This is what a single-process caller -- the MEX
// gateway inside a MATLAB session -- uses to reach all GPUs of a node; the call sequence stays the
// one of MultiframeMotionCoupling.m:977-986 (one solve call in, u and w out).
//
// The motion-coupling operator couples frame t with t+1 only (helpers/constructWarpFF.m:40-58),
// so the blocks exchange two frame-sized messages per boundary and iteration:
//   after the primal sweep  xbar of a block's FIRST frame       -> ghost slot of the previous block
//   after the dual sweep    temporal duals of its LAST frame    -> ghost slot of the next block
// Each block runs its sweeps on its own stream; a copy stream per block carries the outgoing
// frames (hipMemcpyPeerAsync) and events order producer, copy and consumer, so a copy overlaps
// with the neighbour's sweep over the frames that do not need it.
#include "../../include/mmc_hip.h"
#include "mmc_internal.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

struct GroupBlock {
    mmc_handle *h = nullptr;
    int device = 0, begin = 0, end = 0, L = 0;
    bool has_prev = false, has_next = false;
    hipStream_t sweep = nullptr;     // the block's own stream (owned by the handle)
    hipStream_t copy = nullptr;      // carries this block's outgoing frames
    mmc_halo halo;
    mmc_halo state_halo;             // first frame of u, w -> ghost slot of the previous block (diagnostics only)
    // recorded on `sweep`: primal / dual sweep of the iteration finished.
    // recorded on `copy` : this block's xbar has arrived at the previous block / its duals at the next.
    hipEvent_t primal_done = nullptr, dual_done = nullptr, xbar_sent = nullptr, duals_sent = nullptr;
};

struct mmc_group {
    mmc_params prm;
    std::vector<GroupBlock> blocks;
    int Ny = 0, Nx = 0, nc = 0;
    bool has_w = false, inited = false, have_state = false;
    float h = 0.f;
    bool have_h = false;
    int diagnostics = MMC_DIAG_AUTO;
    double balance = 1.0;         // step balance of every block (mmc_set_step_balance)
    bool adapt = false;           // mmc_group_solve chooses it by residual balancing
    double adapt_alpha = 0.5;
    std::string err;
};

static thread_local std::string g_group_create_err;


static int gfail(mmc_group *g, int code, const std::string &msg) {
    if (g) g->err = msg; else g_group_create_err = msg;
    return code;
}

// failure of a block call: take over its message
static int bfail(mmc_group *g, const GroupBlock &b, int code) {
    g->err = std::string("block [") + std::to_string(b.begin) + ", " + std::to_string(b.end) + "): " + mmc_last_error(b.h);
    return code;
}

#define GHIP(g, call)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) return gfail((g), MMC_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

#define GBLK(g, b, call)                          \
    do {                                          \
        int rc_ = (call);                         \
        if (rc_ != MMC_OK) return bfail((g), (b), rc_); \
    } while (0)

extern "C" const char *mmc_group_last_error(const mmc_group *g) { return g ? g->err.c_str() : g_group_create_err.c_str(); }

extern "C" int mmc_group_num_blocks(const mmc_group *g) { return g ? (int)g->blocks.size() : MMC_ERR_INVALID_ARG; }

extern "C" int mmc_group_block_range(const mmc_group *g, int block, int32_t *frame_begin, int32_t *frame_end, int32_t *device) {
    if (!g || block < 0 || block >= (int)g->blocks.size()) return MMC_ERR_INVALID_ARG;
    if (frame_begin) *frame_begin = g->blocks[block].begin;
    if (frame_end) *frame_end = g->blocks[block].end;
    if (device) *device = g->blocks[block].device;
    return MMC_OK;
}

extern "C" mmc_handle *mmc_group_block_handle(mmc_group *g, int block) {
    if (!g || block < 0 || block >= (int)g->blocks.size()) return nullptr;
    return g->blocks[block].h;
}

extern "C" int mmc_group_destroy(mmc_group *g) {
    if (!g) return MMC_OK;
#ifdef MMC_WATCH
    mmc_canary_free();
#endif
    for (GroupBlock &b : g->blocks) {
        if (!b.h) continue;                 // never created: no stream, no events
        if (hipSetDevice(b.device) != hipSuccess) (void)hipGetLastError();
        // streams and events go back to the per-device pool (they live as long as the process)
        if (b.copy) { (void)hipStreamSynchronize(b.copy); mmc_pool_stream_release(b.device, b.copy); }
        for (hipEvent_t e : {b.primal_done, b.dual_done, b.xbar_sent, b.duals_sent}) mmc_pool_event_release(b.device, false, e);
        if (b.h) mmc_destroy(b.h);
    }
    delete g;
    return MMC_OK;
}

extern "C" int mmc_group_create(const mmc_params *p, int nblocks, const int32_t *device_ids, mmc_group **out) {
    if (!p || !out) return gfail(nullptr, MMC_ERR_INVALID_ARG, "mmc_group_create: null argument");
    *out = nullptr;
    if (p->size != sizeof(mmc_params) || p->version != MMC_ABI_VERSION)
        return gfail(nullptr, MMC_ERR_INVALID_ARG, "mmc_group_create: params size/version mismatch");
    if (p->nc_global <= 0 || p->frame_begin != 0 || p->frame_end != p->nc_global)
        return gfail(nullptr, MMC_ERR_INVALID_ARG, "mmc_group_create: params must describe the whole sequence (frame_begin 0, frame_end nc_global)");
    if (nblocks <= 0 || nblocks > p->nc_global)
        return gfail(nullptr, MMC_ERR_INVALID_ARG, "mmc_group_create: need between 1 and nc_global blocks (one frame per block at least)");
    if (p->stream) return gfail(nullptr, MMC_ERR_INVALID_ARG, "mmc_group_create: a group makes its own streams; params.stream must be NULL");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return gfail(nullptr, MMC_ERR_NO_DEVICE, "mmc_group_create: no HIP device available");
    }
    for (int i = 0; i < nblocks; ++i) {
        const int d = device_ids ? device_ids[i] : p->device;
        if (d < 0 || d >= ndev) return gfail(nullptr, MMC_ERR_INVALID_ARG, "mmc_group_create: device ordinal out of range");
    }
    mmc_group *g = new mmc_group();
    g->prm = *p;
    g->nc = p->nc_global;
    g->Ny = p->ny * p->factor;
    g->Nx = p->nx * p->factor;
    g->has_w = !(std::isnan(p->kappa) || p->kappa == 1.f);
    if (p->h > 0.f) { g->h = p->h; g->have_h = true; }
    auto bail = [&](int code, const std::string &msg) {
        g_group_create_err = msg;
        mmc_group_destroy(g);
        return code;
    };
    // contiguous blocks, remainders to the first ones
    const int base = g->nc / nblocks, rem = g->nc % nblocks;
    int at = 0;
    g->blocks.resize((size_t)nblocks);
#ifdef MMC_WATCH
    mmc_canary_new();       // a chunk of the size class that was seen corrupted, allocated where the block table is
#endif
    for (int i = 0; i < nblocks; ++i) {
        GroupBlock &b = g->blocks[(size_t)i];
        b.begin = at;
        b.end = at + base + (i < rem ? 1 : 0);
        at = b.end;
        b.L = b.end - b.begin;
        b.has_prev = b.begin > 0;
        b.has_next = b.end < g->nc;
        b.device = device_ids ? device_ids[i] : p->device;
    }
    for (GroupBlock &b : g->blocks) {
        mmc_params bp = *p;
        bp.frame_begin = b.begin;
        bp.frame_end = b.end;
        bp.device = b.device;
        bp.stream = nullptr;
        int rc = mmc_create(&bp, &b.h);
        if (rc != MMC_OK) return bail(rc, std::string("mmc_group_create: ") + mmc_last_error(nullptr));
        b.sweep = mmc_internal_stream(b.h);
        WCHK("group: after mmc_create of a block");
        if (hipSetDevice(b.device) != hipSuccess) return bail(MMC_ERR_HIP, "mmc_group_create: hipSetDevice failed");
        if (!(b.copy = mmc_pool_stream_acquire(b.device))) return bail(MMC_ERR_HIP, "mmc_group_create: hipStreamCreate failed");
        WCHK("group: after copy stream of a block");
        for (hipEvent_t *e : {&b.primal_done, &b.dual_done, &b.xbar_sent, &b.duals_sent})
            if (!(*e = mmc_pool_event_acquire(b.device, false))) return bail(MMC_ERR_HIP, "mmc_group_create: hipEventCreate failed");
        WCHK("group: after copy stream and events of a block");
        if (mmc_halo_info(b.h, &b.halo) != MMC_OK || mmc_state_halo_info(b.h, &b.state_halo) != MMC_OK)
            return bail(MMC_ERR_HIP, "mmc_group_create: mmc_halo_info failed");
    }
    // direct copies between neighbouring devices where the hardware allows it (xGMI); without
    // peer access hipMemcpyPeerAsync stages through the host and stays correct
    for (size_t i = 0; i + 1 < g->blocks.size(); ++i) {
        const int da = g->blocks[i].device, db = g->blocks[i + 1].device;
        if (da == db) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, da, db) == hipSuccess && can) { (void)hipSetDevice(da); (void)hipDeviceEnablePeerAccess(db, 0); }
        if (hipDeviceCanAccessPeer(&can, db, da) == hipSuccess && can) { (void)hipSetDevice(db); (void)hipDeviceEnablePeerAccess(da, 0); }
        (void)hipGetLastError();      // "already enabled" is fine
    }
    *out = g;
    return MMC_OK;
}

#define REQUIRE_GROUP(g) do { if (!(g)) return MMC_ERR_INVALID_ARG; } while (0)

extern "C" int mmc_group_set_lowres(mmc_group *g, const float *f) {
    REQUIRE_GROUP(g);
    if (!f) return gfail(g, MMC_ERR_INVALID_ARG, "mmc_group_set_lowres: null data");
    const size_t n = (size_t)g->prm.ny * g->prm.nx;
    WCHK("group: entry of set_lowres");
    for (GroupBlock &b : g->blocks) { GBLK(g, b, mmc_set_lowres(b.h, f + n * (size_t)b.begin)); WCHK("group: after set_lowres of a block"); }
    return MMC_OK;
}

extern "C" int mmc_group_set_blur(mmc_group *g, const float *k, int kh, int kw) {
    REQUIRE_GROUP(g);
    for (GroupBlock &b : g->blocks) GBLK(g, b, mmc_set_blur(b.h, k, kh, kw));
    g->inited = false;
    return MMC_OK;
}

extern "C" int mmc_group_set_sampling_taps(mmc_group *g, const int32_t *first_y, const float *w_y, int taps_y,
                                           const int32_t *first_x, const float *w_x, int taps_x) {
    REQUIRE_GROUP(g);
    for (GroupBlock &b : g->blocks) GBLK(g, b, mmc_set_sampling_taps(b.h, first_y, w_y, taps_y, first_x, w_x, taps_x));
    g->inited = false;
    return MMC_OK;
}

extern "C" int mmc_group_set_flow(mmc_group *g, const float *v, const uint8_t *occ) {
    REQUIRE_GROUP(g);
    const size_t N = (size_t)g->Ny * g->Nx, npairs = g->nc > 1 ? (size_t)g->nc - 1 : 0;
    if (npairs > 0 && !v) return gfail(g, MMC_ERR_INVALID_ARG, "mmc_group_set_flow: null flow");
    for (GroupBlock &b : g->blocks) {
        // pairs this block touches: its own and the previous block's last one
        const int lo = b.begin > 0 ? b.begin - 1 : 0;
        if (mmc_num_flow_pairs(b.h) == 0) { GBLK(g, b, mmc_set_flow(b.h, nullptr, nullptr)); continue; }
        GBLK(g, b, mmc_internal_set_flow_planes(b.h, v + N * (size_t)lo, v + N * (npairs + (size_t)lo), occ ? occ + N * (size_t)lo : nullptr));
    }
    g->inited = false;
    return MMC_OK;
}

// xbar of every block's first frame -> ghost slot of the previous block, outside the iteration
// (after a start point was set), synchronously.
static int exchange_xbar_now(mmc_group *g) {
    for (GroupBlock &b : g->blocks) GBLK(g, b, mmc_sync(b.h));
    for (size_t i = 1; i < g->blocks.size(); ++i) {
        GroupBlock &b = g->blocks[i], &pv = g->blocks[i - 1];
        GHIP(g, hipSetDevice(b.device));
        for (int k = 0; k < 2; ++k)
            if (b.halo.send_prev[k])
                GHIP(g, hipMemcpyPeerAsync(pv.halo.recv_next[k], pv.device, b.halo.send_prev[k], b.device, b.halo.frame_bytes, b.copy));
        GHIP(g, hipStreamSynchronize(b.copy));
    }
    return MMC_OK;
}

// The diagnostics of mmc_stats for the whole sequence: every block's first frame of the iterate goes to
// the previous block's ghost slot, then the blocks' partial sums are added up.  Synchronous.
static int group_diagnostics(mmc_group *g, float tol, mmc_stats *st, bool *stop) {
    for (GroupBlock &b : g->blocks) {
        GHIP(g, hipSetDevice(b.device));
        GHIP(g, hipStreamSynchronize(b.copy));
        GBLK(g, b, mmc_sync(b.h));
    }
    for (size_t i = 1; i < g->blocks.size(); ++i) {
        GroupBlock &b = g->blocks[i], &pv = g->blocks[i - 1];
        GHIP(g, hipSetDevice(b.device));
        for (int k = 0; k < 2; ++k)
            if (b.state_halo.send_prev[k])
                GHIP(g, hipMemcpyPeerAsync(pv.state_halo.recv_next[k], pv.device, b.state_halo.send_prev[k], b.device,
                                           b.state_halo.frame_bytes, b.copy));
        GHIP(g, hipStreamSynchronize(b.copy));
    }
    double d[MMC_NUM_DIAG], np = 0.0, nd = 0.0;
    for (double &v : d) v = 0.0;
    for (GroupBlock &b : g->blocks) {
        double part[MMC_NUM_DIAG], p1 = 0.0, d1 = 0.0;
        GBLK(g, b, mmc_diag_partials(b.h, part));
        for (int k = 0; k < MMC_NUM_DIAG; ++k) d[k] += part[k];
        mmc_problem_size(b.h, &p1, &d1);
        np += p1;
        nd += d1;
    }
    *stop = mmc_stats_from_diag(d, np, nd, tol, st) != 0;
    return MMC_OK;
}

extern "C" int mmc_group_set_step_balance(mmc_group *g, float s) {
    REQUIRE_GROUP(g);
    for (GroupBlock &b : g->blocks) GBLK(g, b, mmc_set_step_balance(b.h, s));
    g->balance = (double)s;
    return MMC_OK;
}

extern "C" int mmc_group_set_step_adaptation(mmc_group *g, int enable) {
    REQUIRE_GROUP(g);
    g->adapt = enable != 0;
    g->adapt_alpha = 0.5;
    // a one-block group hands the whole solve to mmc_solve
    if (g->blocks.size() == 1) GBLK(g, g->blocks[0], mmc_set_step_adaptation(g->blocks[0].h, enable));
    return MMC_OK;
}

extern "C" int mmc_group_set_diagnostics(mmc_group *g, int mode) {
    REQUIRE_GROUP(g);
    for (GroupBlock &b : g->blocks) GBLK(g, b, mmc_set_diagnostics(b.h, mode));
    g->diagnostics = mode;
    return MMC_OK;
}

extern "C" int mmc_group_set_state(mmc_group *g, const float *u, const float *w) {
    REQUIRE_GROUP(g);
    if (u && g->has_w && !w) return gfail(g, MMC_ERR_INVALID_ARG, "mmc_group_set_state: this model needs w");
    const size_t N = (size_t)g->Ny * g->Nx;
    for (GroupBlock &b : g->blocks) {
        // a block takes its own frames followed by the next block's first one (contiguous in u)
        if (u) GBLK(g, b, mmc_set_state(b.h, u + N * (size_t)b.begin, g->has_w ? w + N * (size_t)b.begin : nullptr));
        else GBLK(g, b, mmc_set_state(b.h, nullptr, nullptr));
    }
    if (!u) { int rc = exchange_xbar_now(g); if (rc) return rc; }      // the bicubic estimate of the neighbour's first frame
    g->have_state = true;
    return MMC_OK;
}

extern "C" int mmc_group_set_fixed_frame(mmc_group *g, const float *u0, const float *w0) {
    REQUIRE_GROUP(g);
    for (GroupBlock &b : g->blocks) GBLK(g, b, mmc_set_fixed_frame(b.h, u0, w0));
    return MMC_OK;
}

extern "C" int mmc_group_set_flexbox_backward_terms(mmc_group *g, int enable) {
    REQUIRE_GROUP(g);
    for (GroupBlock &b : g->blocks) GBLK(g, b, mmc_set_flexbox_backward_terms(b.h, enable));
    return MMC_OK;
}

extern "C" int mmc_group_init(mmc_group *g) {
    REQUIRE_GROUP(g);
    WCHK("group: entry of init");
    for (GroupBlock &b : g->blocks) GBLK(g, b, mmc_init(b.h));
    WCHK("group: after mmc_init of the blocks");
    if (!g->have_h) {
        // h = (sum |W u0|) / (sum |grad u0|) over the whole sequence, MultiframeMotionCoupling.m:431-437
        double s[2] = {0.0, 0.0};
        for (GroupBlock &b : g->blocks) {
            double part[2];
            GBLK(g, b, mmc_h_partials(b.h, part));
            s[0] += part[0];
            s[1] += part[1];
        }
        if (g->nc == 1) {
            g->h = 1.f;
        } else {
            if (!(s[0] > 0.0) || !(s[1] > 0.0))
                return gfail(g, MMC_ERR_INVALID_ARG, "mmc_group_init: warp or gradient energy of the start point is zero, h = warpPix/gradPix is undefined; pass h explicitly");
            g->h = (float)(s[0] / s[1]);
        }
        g->have_h = true;
    }
    for (GroupBlock &b : g->blocks) GBLK(g, b, mmc_set_h(b.h, g->h));
    g->inited = true;
    return MMC_OK;
}

extern "C" int mmc_group_get_h(const mmc_group *g, float *value) {
    if (!g || !value) return MMC_ERR_INVALID_ARG;
    *value = g->h;
    return MMC_OK;
}

extern "C" int mmc_group_set_h(mmc_group *g, float value) {
    REQUIRE_GROUP(g);
    for (GroupBlock &b : g->blocks) GBLK(g, b, mmc_set_h(b.h, value));
    g->h = value;
    g->have_h = true;
    return MMC_OK;
}

extern "C" int mmc_group_solve(mmc_group *g, int max_iters, float tol, int check_every,
                               mmc_stats *stats, mmc_progress_cb cb, void *user) {
    REQUIRE_GROUP(g);
    WCHK("group: entry of solve");
    if (!g->inited) return gfail(g, MMC_ERR_STATE, "No initialization found, run mmc_group_init first.");
    if (max_iters < 0) return gfail(g, MMC_ERR_INVALID_ARG, "mmc_group_solve: negative iteration count");
    if (check_every <= 0) check_every = max_iters > 0 ? max_iters : 1;
    mmc_stats st;
    memset(&st, 0, sizeof(st));
    st.primal_energy = st.dual_energy = st.gap = st.dual_infeasibility = std::nan("");
    st.opt_res_primal = st.opt_res_dual = st.norm_KTy = st.norm_Kx = std::nan("");
    const bool diag = g->diagnostics == MMC_DIAG_ALWAYS || (g->diagnostics == MMC_DIAG_AUTO && (tol > 0.f || cb));
    std::vector<GroupBlock> &B = g->blocks;
    const size_t nb = B.size();
    if (nb == 1) {
        GBLK(g, B[0], mmc_solve(B[0].h, max_iters, tol, check_every, stats, cb, user));
        return MMC_OK;
    }
    for (GroupBlock &b : B) GBLK(g, b, mmc_sync(b.h));
    const auto t_start = std::chrono::steady_clock::now();
    st.step_balance = g->balance;
    for (int it = 0; it < max_iters; ++it) {
        const bool want_check = ((it + 1) % check_every == 0) || it == max_iters - 1;
        const bool adapt_now = g->adapt && (it + 1) % 50 == 0;
        const bool check = want_check || adapt_now;          // sweeps with reduced sums
        if (check) for (GroupBlock &b : B) GBLK(g, b, mmc_sums_reset(b.h));
        // primal sweep: a block's first frame needs the previous block's temporal duals, and its
        // xbar may only be overwritten once the copy of the last iteration has read it
        for (size_t i = 0; i < nb; ++i) {
            GroupBlock &b = B[i];
            GHIP(g, hipSetDevice(b.device));
            if (b.has_prev) {
                GBLK(g, b, mmc_sweep_primal(b.h, 1, b.L, check));
                GHIP(g, hipStreamWaitEvent(b.sweep, B[i - 1].duals_sent, 0));
                GHIP(g, hipStreamWaitEvent(b.sweep, b.xbar_sent, 0));
                GBLK(g, b, mmc_sweep_primal(b.h, 0, 1, check));
            } else {
                GBLK(g, b, mmc_sweep_primal(b.h, 0, b.L, check));
            }
            GHIP(g, hipEventRecord(b.primal_done, b.sweep));
        }
        // xbar of the first frame -> previous block (whose last dual sweep has finished reading its ghost)
        for (size_t i = 1; i < nb; ++i) {
            GroupBlock &b = B[i], &pv = B[i - 1];
            GHIP(g, hipSetDevice(b.device));
            GHIP(g, hipStreamWaitEvent(b.copy, b.primal_done, 0));
            GHIP(g, hipStreamWaitEvent(b.copy, pv.dual_done, 0));
            for (int k = 0; k < 2; ++k)
                if (b.halo.send_prev[k])
                    GHIP(g, hipMemcpyPeerAsync(pv.halo.recv_next[k], pv.device, b.halo.send_prev[k], b.device, b.halo.frame_bytes, b.copy));
            GHIP(g, hipEventRecord(b.xbar_sent, b.copy));
        }
        // dual sweep: a block's last frame needs the next block's xbar, and its temporal duals may
        // only be overwritten once the copy of the last iteration has read them
        for (size_t i = 0; i < nb; ++i) {
            GroupBlock &b = B[i];
            GHIP(g, hipSetDevice(b.device));
            if (b.has_next) {
                GBLK(g, b, mmc_sweep_dual(b.h, 0, b.L - 1, check));
                GHIP(g, hipStreamWaitEvent(b.sweep, B[i + 1].xbar_sent, 0));
                GHIP(g, hipStreamWaitEvent(b.sweep, b.duals_sent, 0));
                GBLK(g, b, mmc_sweep_dual(b.h, b.L - 1, b.L, check));
            } else {
                GBLK(g, b, mmc_sweep_dual(b.h, 0, b.L, check));
            }
            GHIP(g, hipEventRecord(b.dual_done, b.sweep));
        }
        // temporal duals of the last frame -> next block (whose primal sweep has finished reading its ghost)
        for (size_t i = 0; i + 1 < nb; ++i) {
            GroupBlock &b = B[i], &nx = B[i + 1];
            GHIP(g, hipSetDevice(b.device));
            GHIP(g, hipStreamWaitEvent(b.copy, b.dual_done, 0));
            GHIP(g, hipStreamWaitEvent(b.copy, nx.primal_done, 0));
            for (int k = 0; k < 2; ++k)
                if (b.halo.send_next[k])
                    GHIP(g, hipMemcpyPeerAsync(nx.halo.recv_prev[k], nx.device, b.halo.send_next[k], b.device, b.halo.frame_bytes, b.copy));
            GHIP(g, hipEventRecord(b.duals_sent, b.copy));
        }
        st.iterations = it + 1;
        if (adapt_now) {
            double ws[2] = {0, 0};
            for (GroupBlock &b : B) {
                double part[2];
                GBLK(g, b, mmc_step_sums_read(b.h, part));
                ws[0] += part[0]; ws[1] += part[1];
            }
            if (mmc_step_balance_rule(ws, &g->balance, &g->adapt_alpha))
                for (GroupBlock &b : B) GBLK(g, b, mmc_set_step_balance(b.h, (float)g->balance));
            st.step_balance = g->balance;
        }
        if (want_check) {
            double s[MMC_NUM_SUMS] = {0, 0, 0, 0};
            for (GroupBlock &b : B) {
                double part[MMC_NUM_SUMS];
                GBLK(g, b, mmc_sums_read(b.h, part));
                for (int k = 0; k < MMC_NUM_SUMS; ++k) s[k] += part[k];
            }
            const double tiny = 1e-30;
            st.res_primal = std::sqrt(s[0]) / std::max(std::sqrt(s[1]), tiny);
            st.res_dual = std::sqrt(s[2]) / std::max(std::sqrt(s[3]), tiny);
            bool stop = tol > 0.f && st.res_primal < tol && st.res_dual < tol;
            if (diag) { int rc = group_diagnostics(g, tol, &st, &stop); if (rc) return rc; }
            st.solve_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
            if (cb && cb(st.iterations, &st, user)) break;
            if (stop) { st.converged = 1; break; }
        }
    }
    for (GroupBlock &b : B) {
        GHIP(g, hipSetDevice(b.device));
        GHIP(g, hipStreamSynchronize(b.copy));
        GBLK(g, b, mmc_sync(b.h));
    }
    st.solve_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    if (stats) *stats = st;
    return MMC_OK;
}

extern "C" int mmc_group_get_state(mmc_group *g, float *u, float *w) {
    REQUIRE_GROUP(g);
    if (w && !g->has_w) return gfail(g, MMC_ERR_INVALID_ARG, "mmc_group_get_state: this model has no w");
    const size_t N = (size_t)g->Ny * g->Nx;
    WCHK("group: entry of get_state");
    for (GroupBlock &b : g->blocks)
        GBLK(g, b, mmc_get_state(b.h, u ? u + N * (size_t)b.begin : nullptr, w ? w + N * (size_t)b.begin : nullptr));
    return MMC_OK;
}
