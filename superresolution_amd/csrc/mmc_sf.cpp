// mmc_sf_*: C ABI of the classical single-frame super-resolution solve
// (classical methods/singleframeMotionSR_mitzel_prost.m:82-172 and its two siblings): one
// high-resolution frame, one motion-compensated data term per low-resolution frame, TV.
// The kernels are in mmc_sf_kernels.hip.  Geometry, the composite kernel of D*B and its column
// sums come from a one-frame mmc_handle owned by this handle, so both paths share that code.
#include "../../include/mmc_hip.h"
#include "mmc_internal.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

struct mmc_sf_handle {
    mmc_sf_params prm;
    mmc_handle *geo = nullptr;           // geometry + data operator (owns the stream)
    hipStream_t stream = nullptr;
    SfArrays a;
    int N = 0, n = 0, nf = 0, central = 0;
    bool have_f = false, have_flow = false, inited = false;
    std::vector<void *> allocs;
    std::string err;
};

static thread_local std::string g_sf_create_err;

static int sfail(mmc_sf_handle *h, int code, const std::string &msg) {
    if (h) h->err = msg; else g_sf_create_err = msg;
    return code;
}

#define SFHIP(h, call)                                                                         \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return sfail((h), MMC_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

#define SF_REQUIRE(h) do { if (!(h)) return MMC_ERR_INVALID_ARG; } while (0)

static int geo_fail(mmc_sf_handle *h, int code) { return sfail(h, code, mmc_last_error(h->geo)); }

static int sf_use_device(mmc_sf_handle *h) {
    SFHIP(h, hipSetDevice(h->prm.device));
    (void)hipGetLastError();
    return MMC_OK;
}

template <typename T>
static int sf_alloc(mmc_sf_handle *h, T **ptr, size_t count) {
    void *p = nullptr;
    const size_t bytes = (count ? count : 1) * sizeof(T);
    SFHIP(h, hipMalloc(&p, bytes));
    h->allocs.push_back(p);
    SFHIP(h, hipMemsetAsync(p, 0, bytes, h->stream));
    *ptr = (T *)p;
    return MMC_OK;
}

extern "C" const char *mmc_sf_last_error(const mmc_sf_handle *h) { return h ? h->err.c_str() : g_sf_create_err.c_str(); }

extern "C" int mmc_sf_destroy(mmc_sf_handle *h) {
    if (!h) return MMC_OK;
    (void)hipSetDevice(h->prm.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (void *p : h->allocs) (void)hipFree(p);
    if (h->geo) mmc_destroy(h->geo);
    delete h;
    return MMC_OK;
}

extern "C" int mmc_sf_create(const mmc_sf_params *p, mmc_sf_handle **out) {
    if (!p || !out) return sfail(nullptr, MMC_ERR_INVALID_ARG, "mmc_sf_create: null argument");
    *out = nullptr;
    if (p->size != sizeof(mmc_sf_params) || p->version != MMC_ABI_VERSION)
        return sfail(nullptr, MMC_ERR_INVALID_ARG, "mmc_sf_create: params size/version mismatch");
    if (p->num_frames <= 0) return sfail(nullptr, MMC_ERR_INVALID_ARG, "mmc_sf_create: need at least one low-resolution frame");
    if (p->num_frames > 65535) return sfail(nullptr, MMC_ERR_INVALID_ARG, "mmc_sf_create: more than 65535 frames (a grid dimension of the per-frame kernels)");
    if (!(p->huber_eps >= 0.f)) return sfail(nullptr, MMC_ERR_INVALID_ARG, "mmc_sf_create: huber_eps must be >= 0");
    if (p->sampling != MMC_SAMPLING_AVERAGE && p->sampling != MMC_SAMPLING_STRIDE)
        return sfail(nullptr, MMC_ERR_UNSUPPORTED, "mmc_sf_create: the single-frame solve takes the built-in samplers only ('average' at factor 4, 'stride')");
    // geometry handle: one frame, validates ny, nx, factor, alpha, sampling, device the way mmc_create does
    mmc_params gp;
    memset(&gp, 0, sizeof(gp));
    gp.size = sizeof(gp); gp.version = MMC_ABI_VERSION;
    gp.ny = p->ny; gp.nx = p->nx; gp.factor = p->factor;
    gp.nc_global = 1; gp.frame_begin = 0; gp.frame_end = 1;
    gp.alpha = p->alpha; gp.kappa = 1.f; gp.h = 1.f;
    gp.flow_direction = MMC_FLOW_BACKWARD; gp.sampling = p->sampling;
    gp.device = p->device; gp.stream = p->stream;
    mmc_handle *geo = nullptr;
    int rc = mmc_internal_create_geometry(&gp, &geo);
    if (rc != MMC_OK) return sfail(nullptr, rc, mmc_last_error(nullptr));
    mmc_sf_handle *h = new mmc_sf_handle();
    h->prm = *p;
    h->geo = geo;
    h->stream = mmc_internal_stream(geo);
    const Geom &g = *mmc_internal_geom(geo);
    h->N = g.N; h->n = g.n; h->nf = p->num_frames;
    h->central = (p->num_frames + 1) / 2 - 1;                 // ceil(numFrames/2), 1-based (:21)
    auto bail = [&](int code) { g_sf_create_err = h->err; mmc_sf_destroy(h); return code; };
    if ((long long)h->N * h->nf > (long long)INT32_MAX * 2)
        return bail(sfail(h, MMC_ERR_INVALID_ARG, "mmc_sf_create: num_frames x frame too large"));
    memset(&h->a, 0, sizeof(h->a));
    SfArrays &a = h->a;
    a.nf = h->nf; a.alpha = p->alpha; a.delta = p->huber_eps;
    const size_t N = (size_t)h->N, n = (size_t)h->n, nf = (size_t)h->nf;
    if ((rc = sf_alloc(h, &a.u, N)) || (rc = sf_alloc(h, &a.ub, N)) || (rc = sf_alloc(h, &a.qx, N)) || (rc = sf_alloc(h, &a.qy, N)) ||
        (rc = sf_alloc(h, &a.vx, nf * N)) || (rc = sf_alloc(h, &a.vy, nf * N)) || (rc = sf_alloc(h, &a.wimg, nf * N)) ||
        (rc = sf_alloc(h, &a.p, nf * n)) || (rc = sf_alloc(h, &a.f, nf * n)) || (rc = sf_alloc(h, &a.sigp, nf * n)) ||
        (rc = sf_alloc(h, &a.tau, N)) || (rc = sf_alloc(h, &a.acc, N)) ||
        (rc = sf_alloc(h, &a.ck12, (size_t)144)) || (rc = sf_alloc(h, &a.bbox, nf * (size_t)g.ntiles * 4)) || (rc = sf_alloc(h, &a.irregular, nf)) ||
        (rc = sf_alloc(h, &a.sums, (size_t)MMC_NUM_SUMS_INT * MMC_SUM_SLOTS)))
        return bail(rc);
    if (hipStreamSynchronize(h->stream) != hipSuccess) return bail(sfail(h, MMC_ERR_HIP, "hipStreamSynchronize failed"));
    *out = h;
    return MMC_OK;
}

extern "C" int mmc_sf_set_lowres(mmc_sf_handle *h, const float *f) {
    SF_REQUIRE(h);
    if (!f) return sfail(h, MMC_ERR_INVALID_ARG, "mmc_sf_set_lowres: null data");
    int rc = sf_use_device(h); if (rc) return rc;
    SFHIP(h, hipMemcpyAsync(h->a.f, f, (size_t)h->nf * h->n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    SFHIP(h, hipStreamSynchronize(h->stream));
    h->have_f = true;
    return MMC_OK;
}

extern "C" int mmc_sf_set_blur(mmc_sf_handle *h, const float *k, int kh, int kw) {
    SF_REQUIRE(h);
    if (k) {
        for (int i = 0; i < kh * kw && kh > 0 && kw > 0; ++i)
            if (!(k[i] >= 0.f)) return sfail(h, MMC_ERR_UNSUPPORTED, "mmc_sf_set_blur: kernel has a negative (or NaN) entry");
    }
    int rc = mmc_set_blur(h->geo, k, kh, kw);
    if (rc != MMC_OK) return geo_fail(h, rc);
    h->inited = false;
    return MMC_OK;
}

extern "C" int mmc_sf_set_flow(mmc_sf_handle *h, const float *v, const uint8_t *occ) {
    SF_REQUIRE(h);
    if (!v) return sfail(h, MMC_ERR_INVALID_ARG, "mmc_sf_set_flow: null flow");
    int rc = sf_use_device(h); if (rc) return rc;
    const size_t N = (size_t)h->N, cnt = N * (size_t)h->nf;
    const Geom &g = *mmc_internal_geom(h->geo);
    SFHIP(h, hipMemcpyAsync(h->a.vx, v, cnt * sizeof(float), hipMemcpyHostToDevice, h->stream));
    SFHIP(h, hipMemcpyAsync(h->a.vy, v + cnt, cnt * sizeof(float), hipMemcpyHostToDevice, h->stream));
    if (occ) {
        // the mask travels through the scratch plane of warped frames (nf*N floats >= nf*N bytes)
        uint8_t *docc = reinterpret_cast<uint8_t *>(h->a.wimg);
        SFHIP(h, hipMemcpyAsync(docc, occ, cnt, hipMemcpyHostToDevice, h->stream));
        SFHIP(h, mmc_launch_fold_occ(g, h->a.vx, docc, cnt, h->stream));
    }
    // the central frame is warped by the identity (:35-37, :66-69)
    SFHIP(h, hipMemsetAsync(h->a.vx + (size_t)h->central * N, 0, N * sizeof(float), h->stream));
    SFHIP(h, hipMemsetAsync(h->a.vy + (size_t)h->central * N, 0, N * sizeof(float), h->stream));
    SFHIP(h, hipStreamSynchronize(h->stream));
    h->have_flow = true;
    h->inited = false;
    return MMC_OK;
}

static int sf_zero_duals(mmc_sf_handle *h) {
    SfArrays &a = h->a;
    SFHIP(h, hipMemsetAsync(a.p, 0, (size_t)h->nf * h->n * sizeof(float), h->stream));
    SFHIP(h, hipMemsetAsync(a.qx, 0, (size_t)h->N * sizeof(float), h->stream));
    SFHIP(h, hipMemsetAsync(a.qy, 0, (size_t)h->N * sizeof(float), h->stream));
    SFHIP(h, hipMemsetAsync(a.acc, 0, (size_t)h->N * sizeof(unsigned long long), h->stream));
    return MMC_OK;
}

extern "C" int mmc_sf_set_state(mmc_sf_handle *h, const float *u) {
    SF_REQUIRE(h);
    int rc = sf_use_device(h); if (rc) return rc;
    const size_t bytes = (size_t)h->N * sizeof(float);
    if (u) {
        SFHIP(h, hipMemcpyAsync(h->a.u, u, bytes, hipMemcpyHostToDevice, h->stream));
    } else {
        SFHIP(h, hipMemsetAsync(h->a.u, 0, bytes, h->stream));
    }
    SFHIP(h, hipMemcpyAsync(h->a.ub, h->a.u, bytes, hipMemcpyDeviceToDevice, h->stream));
    if ((rc = sf_zero_duals(h))) return rc;
    SFHIP(h, hipStreamSynchronize(h->stream));
    return MMC_OK;
}

extern "C" int mmc_sf_init(mmc_sf_handle *h) {
    SF_REQUIRE(h);
    if (!h->have_f) return sfail(h, MMC_ERR_STATE, "mmc_sf_init: no low-resolution data (mmc_sf_set_lowres)");
    if (!h->have_flow) return sfail(h, MMC_ERR_STATE, "mmc_sf_init: no flow (mmc_sf_set_flow)");
    int rc = sf_use_device(h); if (rc) return rc;
    if ((rc = mmc_internal_build_data_operator(h->geo)) != MMC_OK) return geo_fail(h, rc);
    const Geom &g = *mmc_internal_geom(h->geo);
    const Arrays &ga = *mmc_internal_arrays(h->geo);
    h->a.cker = ga.cker;
    h->a.cA = ga.cA;
    h->a.ck12_ok = 0;
    if (g.f == 4 && g.CH <= 12 && g.CW <= 12) {
        std::vector<float> ck((size_t)g.CH * g.CW), pad(144, 0.f);
        SFHIP(h, hipMemcpyAsync(ck.data(), ga.cker, ck.size() * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        SFHIP(h, hipStreamSynchronize(h->stream));
        for (int nn = 0; nn < g.CW; ++nn)
            for (int mm = 0; mm < g.CH; ++mm) pad[(size_t)mm + 12 * (size_t)nn] = ck[(size_t)mm + (size_t)g.CH * nn];
        SFHIP(h, hipMemcpyAsync(h->a.ck12, pad.data(), pad.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
        SFHIP(h, hipStreamSynchronize(h->stream));
        h->a.ck12_ok = 1;
    }
    // source rectangles of the tile scatter; a frame with a rectangle too large to walk is scattered with global atomics
    {
        SFHIP(h, mmc_sf_launch_bbox(g, h->a, h->stream));
        std::vector<int> bb((size_t)h->nf * g.ntiles * 4), irr((size_t)h->nf, 0);
        SFHIP(h, hipMemcpyAsync(bb.data(), h->a.bbox, bb.size() * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        SFHIP(h, hipStreamSynchronize(h->stream));
        h->a.any_irregular = 0;
        for (int i = 0; i < h->nf; ++i)
            for (int t = 0; t < g.ntiles; ++t) {
                const int *b = &bb[((size_t)i * g.ntiles + t) * 4];
                if (b[1] >= b[0] && (long long)(b[1] - b[0] + 1) * (b[3] - b[2] + 1) > MMC_SF_BOX_MAX) { irr[(size_t)i] = 1; h->a.any_irregular = 1; }
            }
        SFHIP(h, hipMemcpyAsync(h->a.irregular, irr.data(), irr.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
        SFHIP(h, hipStreamSynchronize(h->stream));
    }
    // step sizes: the operators applied to constant input (no entry of K is negative).  These launches write
    // sigp and tau only and leave the accumulator at zero; the iterate and the duals are not touched.
    SFHIP(h, hipMemsetAsync(h->a.acc, 0, (size_t)h->N * sizeof(unsigned long long), h->stream));
    SFHIP(h, mmc_sf_launch_dual(g, h->a, MODE_STEPS, false, h->stream));
    SFHIP(h, mmc_sf_launch_primal(g, h->a, MODE_STEPS, false, h->stream));
    SFHIP(h, hipStreamSynchronize(h->stream));
    h->inited = true;
    return MMC_OK;
}

static int sf_read_sums(mmc_sf_handle *h, double out[MMC_NUM_SUMS]) {
    std::vector<double> host((size_t)MMC_NUM_SUMS_INT * MMC_SUM_SLOTS);
    SFHIP(h, hipMemcpyAsync(host.data(), h->a.sums, host.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    SFHIP(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < MMC_NUM_SUMS; ++i) {
        double s = 0.0;
        for (int k = 0; k < MMC_SUM_SLOTS; ++k) s += host[(size_t)i * MMC_SUM_SLOTS + k];
        out[i] = s;
    }
    return MMC_OK;
}

extern "C" int mmc_sf_solve(mmc_sf_handle *h, int max_iters, float tol, int check_every,
                            mmc_stats *stats, mmc_progress_cb cb, void *user) {
    SF_REQUIRE(h);
    if (!h->inited) return sfail(h, MMC_ERR_STATE, "mmc_sf_solve: call mmc_sf_init first");
    if (max_iters < 0) return sfail(h, MMC_ERR_INVALID_ARG, "mmc_sf_solve: negative iteration count");
    int rc = sf_use_device(h); if (rc) return rc;
    const Geom &g = *mmc_internal_geom(h->geo);
    const auto t0 = std::chrono::steady_clock::now();
    mmc_stats st;
    memset(&st, 0, sizeof(st));
    st.step_balance = 1.0;
    st.primal_energy = st.dual_energy = st.gap = st.dual_infeasibility = std::nan("");      // not evaluated by this solve
    st.opt_res_primal = st.opt_res_dual = st.norm_KTy = st.norm_Kx = std::nan("");
    if (check_every <= 0) check_every = std::max(max_iters, 1);
    for (int it = 0; it < max_iters; ++it) {
        const bool check = ((it + 1) % check_every == 0) || it == max_iters - 1;
        if (check) SFHIP(h, hipMemsetAsync(h->a.sums, 0, (size_t)MMC_NUM_SUMS_INT * MMC_SUM_SLOTS * sizeof(double), h->stream));
        SFHIP(h, mmc_sf_launch_primal(g, h->a, MODE_UPDATE, check, h->stream));
        SFHIP(h, mmc_sf_launch_dual(g, h->a, MODE_UPDATE, check, h->stream));
        st.iterations = it + 1;
        if (check) {
            double s[MMC_NUM_SUMS];
            if ((rc = sf_read_sums(h, s))) return rc;
            const double tiny = 1e-30;
            st.res_primal = std::sqrt(s[0]) / std::max(std::sqrt(s[1]), tiny);
            st.res_dual = std::sqrt(s[2]) / std::max(std::sqrt(s[3]), tiny);
            st.solve_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (cb && cb(it + 1, &st, user)) break;
            if (tol > 0.f && st.res_primal < tol && st.res_dual < tol) { st.converged = 1; break; }
        }
    }
    SFHIP(h, hipStreamSynchronize(h->stream));
    st.solve_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (stats) *stats = st;
    return MMC_OK;
}

extern "C" int mmc_sf_get_state(mmc_sf_handle *h, float *u) {
    SF_REQUIRE(h);
    if (!u) return sfail(h, MMC_ERR_INVALID_ARG, "mmc_sf_get_state: null output");
    int rc = sf_use_device(h); if (rc) return rc;
    SFHIP(h, hipMemcpyAsync(u, h->a.u, (size_t)h->N * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    SFHIP(h, hipStreamSynchronize(h->stream));
    return MMC_OK;
}

// ---- test hooks: state (u, ub, duals) is saved and restored around them --------------------------

struct SfSaved { std::vector<float> ub, p, qx, qy; };

static int sf_save(mmc_sf_handle *h, SfSaved &s) {
    s.ub.resize((size_t)h->N); s.qx.resize((size_t)h->N); s.qy.resize((size_t)h->N); s.p.resize((size_t)h->nf * h->n);
    SFHIP(h, hipMemcpyAsync(s.ub.data(), h->a.ub, s.ub.size() * 4, hipMemcpyDeviceToHost, h->stream));
    SFHIP(h, hipMemcpyAsync(s.qx.data(), h->a.qx, s.qx.size() * 4, hipMemcpyDeviceToHost, h->stream));
    SFHIP(h, hipMemcpyAsync(s.qy.data(), h->a.qy, s.qy.size() * 4, hipMemcpyDeviceToHost, h->stream));
    SFHIP(h, hipMemcpyAsync(s.p.data(), h->a.p, s.p.size() * 4, hipMemcpyDeviceToHost, h->stream));
    SFHIP(h, hipStreamSynchronize(h->stream));
    return MMC_OK;
}

static int sf_restore(mmc_sf_handle *h, const SfSaved &s) {
    SFHIP(h, hipMemcpyAsync(h->a.ub, s.ub.data(), s.ub.size() * 4, hipMemcpyHostToDevice, h->stream));
    SFHIP(h, hipMemcpyAsync(h->a.qx, s.qx.data(), s.qx.size() * 4, hipMemcpyHostToDevice, h->stream));
    SFHIP(h, hipMemcpyAsync(h->a.qy, s.qy.data(), s.qy.size() * 4, hipMemcpyHostToDevice, h->stream));
    SFHIP(h, hipMemcpyAsync(h->a.p, s.p.data(), s.p.size() * 4, hipMemcpyHostToDevice, h->stream));
    SFHIP(h, hipStreamSynchronize(h->stream));
    return MMC_OK;
}

extern "C" int mmc_sf_debug_apply_K(mmc_sf_handle *h, const float *u, float *out_p, float *out_q) {
    SF_REQUIRE(h);
    if (!h->inited) return sfail(h, MMC_ERR_STATE, "mmc_sf_debug_apply_K: call mmc_sf_init first");
    if (!u || !out_p || !out_q) return sfail(h, MMC_ERR_INVALID_ARG, "mmc_sf_debug_apply_K: null argument");
    int rc = sf_use_device(h); if (rc) return rc;
    const Geom &g = *mmc_internal_geom(h->geo);
    SfSaved s;
    if ((rc = sf_save(h, s))) return rc;
    const size_t N = (size_t)h->N;
    SFHIP(h, hipMemcpyAsync(h->a.ub, u, N * 4, hipMemcpyHostToDevice, h->stream));
    SFHIP(h, mmc_sf_launch_dual(g, h->a, MODE_RAW, false, h->stream));
    SFHIP(h, hipMemcpyAsync(out_p, h->a.p, (size_t)h->nf * h->n * 4, hipMemcpyDeviceToHost, h->stream));
    SFHIP(h, hipMemcpyAsync(out_q, h->a.qx, N * 4, hipMemcpyDeviceToHost, h->stream));
    SFHIP(h, hipMemcpyAsync(out_q + N, h->a.qy, N * 4, hipMemcpyDeviceToHost, h->stream));
    SFHIP(h, hipStreamSynchronize(h->stream));
    return sf_restore(h, s);
}

extern "C" int mmc_sf_debug_apply_KT(mmc_sf_handle *h, const float *p, const float *q, float *out_u) {
    SF_REQUIRE(h);
    if (!h->inited) return sfail(h, MMC_ERR_STATE, "mmc_sf_debug_apply_KT: call mmc_sf_init first");
    if (!p || !q || !out_u) return sfail(h, MMC_ERR_INVALID_ARG, "mmc_sf_debug_apply_KT: null argument");
    int rc = sf_use_device(h); if (rc) return rc;
    const Geom &g = *mmc_internal_geom(h->geo);
    SfSaved s;
    if ((rc = sf_save(h, s))) return rc;
    const size_t N = (size_t)h->N;
    SFHIP(h, hipMemcpyAsync(h->a.p, p, (size_t)h->nf * h->n * 4, hipMemcpyHostToDevice, h->stream));
    SFHIP(h, hipMemcpyAsync(h->a.qx, q, N * 4, hipMemcpyHostToDevice, h->stream));
    SFHIP(h, hipMemcpyAsync(h->a.qy, q + N, N * 4, hipMemcpyHostToDevice, h->stream));
    SFHIP(h, mmc_sf_launch_primal(g, h->a, MODE_RAW, false, h->stream));
    SFHIP(h, hipMemcpyAsync(out_u, h->a.ub, N * 4, hipMemcpyDeviceToHost, h->stream));
    SFHIP(h, hipStreamSynchronize(h->stream));
    return sf_restore(h, s);
}

extern "C" int mmc_sf_debug_steps(mmc_sf_handle *h, float *tau, float *sigma_p, float *sigma_q) {
    SF_REQUIRE(h);
    if (!h->inited) return sfail(h, MMC_ERR_STATE, "mmc_sf_debug_steps: call mmc_sf_init first");
    int rc = sf_use_device(h); if (rc) return rc;
    const Geom &g = *mmc_internal_geom(h->geo);
    const size_t N = (size_t)h->N;
    if (tau) SFHIP(h, hipMemcpyAsync(tau, h->a.tau, N * 4, hipMemcpyDeviceToHost, h->stream));
    if (sigma_p) SFHIP(h, hipMemcpyAsync(sigma_p, h->a.sigp, (size_t)h->nf * h->n * 4, hipMemcpyDeviceToHost, h->stream));
    SFHIP(h, hipStreamSynchronize(h->stream));
    if (sigma_q) {
        // the gradient rows' step depends on the pixel position only; the dual sweep recomputes it
        for (int c = 0; c < g.Nx; ++c)
            for (int r = 0; r < g.Ny; ++r) {
                const float rs = std::max(c < g.Nx - 1 ? 2.f : 0.f, r < g.Ny - 1 ? 2.f : 0.f);
                sigma_q[(size_t)r + (size_t)g.Ny * c] = rs > 0.f ? 1.f / rs : 1.f;
            }
    }
    return MMC_OK;
}
