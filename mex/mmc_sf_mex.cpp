// mmc_sf_mex.cpp -- MEX gateway between the reference's classical single-frame methods
// (classical methods/singleframeMotionSR_mitzel_prost.m, ..._mitzel_prost_fast.m, ..._unger_prost.m)
// and the mmc_sf_* calls of libmmc_hip.so.  It replaces :64-172 of those files (operators, prost
// problem, solve); their flow estimation (:28-62) stays.
//
// Build (on a machine with MATLAB and ROCm; not built by this repository's tests):
//   mex -R2018a mex/mmc_sf_mex.cpp -Iinclude -Lsuperresolution_amd/lib -lmmc_hip
//
// No logic here: doubles to float32, one mmc_sf_* call per command, status to MATLAB error.
//   h = mmc_sf_mex('create', S)     S: ny, nx, factor, num_frames, alpha, huber_eps, sampling, device
//   mmc_sf_mex('lowres', h, J0)     ny x nx x num_frames double
//   mmc_sf_mex('blur',   h, k)      odd x odd double without negative entries, or []
//   mmc_sf_mex('flow',   h, v, occ) v: Ny x Nx x num_frames x 2 double; occ: logical / double or []
//   mmc_sf_mex('state',  h, u)      start point Ny x Nx, [] = zero
//   mmc_sf_mex('init',   h)
//   [I, stats] = mmc_sf_mex('solve', h, max_iters, tol, check_every)
//   mmc_sf_mex('destroy', h)
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "mex.h"
#include "mmc_hip.h"

namespace {

struct Entry { mmc_sf_handle *h; mmc_sf_params p; };
std::map<uint64_t, Entry> g_handles;
uint64_t g_next = 1;

void at_exit() {
    for (auto &kv : g_handles) mmc_sf_destroy(kv.second.h);
    g_handles.clear();
}

void check(int rc, const mmc_sf_handle *h) {
    if (rc != MMC_OK) mexErrMsgIdAndTxt("MMC:sf", "%s", mmc_sf_last_error(h));
}

void need(int nrhs, int n, const char *cmd) {
    if (nrhs < n) mexErrMsgIdAndTxt("MMC:sf", "'%s' needs %d arguments", cmd, n);
}

Entry &lookup(const mxArray *a) {
    auto it = g_handles.find((uint64_t)mxGetScalar(a));
    if (it == g_handles.end()) mexErrMsgIdAndTxt("MMC:sf", "invalid handle");
    return it->second;
}

std::vector<float> to_float(const mxArray *a, size_t numel, const char *what) {
    if (!mxIsDouble(a) || mxIsComplex(a) || mxIsSparse(a)) mexErrMsgIdAndTxt("MMC:sf", "%s: expected a full real double array", what);
    if (mxGetNumberOfElements(a) != numel)
        mexErrMsgIdAndTxt("MMC:sf", "%s has %zu elements, expected %zu", what, mxGetNumberOfElements(a), numel);
    const double *d = mxGetDoubles(a);
    return std::vector<float>(d, d + numel);
}

double field(const mxArray *s, const char *name, double dflt) {
    const mxArray *f = mxGetField(s, 0, name);
    return (f && !mxIsEmpty(f)) ? mxGetScalar(f) : dflt;
}

int progress(int it, const mmc_stats *st, void *) {
    mexPrintf("  it %6d  res_primal %.3e  res_dual %.3e  (%.2f s)\n", it, st->res_primal, st->res_dual, st->solve_seconds);
    return 0;
}

}  // namespace

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    if (nrhs < 1 || !mxIsChar(prhs[0])) mexErrMsgIdAndTxt("MMC:sf", "first argument must be a command string");
    char cmdbuf[32];
    mxGetString(prhs[0], cmdbuf, sizeof(cmdbuf));
    const std::string cmd(cmdbuf);
    if (!mexIsLocked()) mexAtExit(at_exit);

    if (cmd == "create") {
        need(nrhs, 2, "create");
        if (!mxIsStruct(prhs[1])) mexErrMsgIdAndTxt("MMC:sf", "'create' takes a struct");
        mmc_sf_params p;
        p.size = sizeof(p); p.version = MMC_ABI_VERSION;
        p.ny = (int32_t)field(prhs[1], "ny", 0); p.nx = (int32_t)field(prhs[1], "nx", 0);
        p.factor = (int32_t)field(prhs[1], "factor", 4);
        p.num_frames = (int32_t)field(prhs[1], "num_frames", 0);
        p.alpha = (float)field(prhs[1], "alpha", 0.1);
        p.huber_eps = (float)field(prhs[1], "huber_eps", 0.0);
        p.sampling = (int32_t)field(prhs[1], "sampling", MMC_SAMPLING_AVERAGE);
        p.device = (int32_t)field(prhs[1], "device", 0);
        p.stream = nullptr;
        mmc_sf_handle *h = nullptr;
        check(mmc_sf_create(&p, &h), nullptr);
        if (!mexIsLocked()) mexLock();
        g_handles[g_next] = Entry{h, p};
        plhs[0] = mxCreateDoubleScalar((double)g_next++);
        return;
    }
    need(nrhs, 2, cmd.c_str());
    if (cmd == "destroy") {
        auto it = g_handles.find((uint64_t)mxGetScalar(prhs[1]));
        if (it != g_handles.end()) { mmc_sf_destroy(it->second.h); g_handles.erase(it); }
        if (g_handles.empty() && mexIsLocked()) mexUnlock();
        return;
    }
    Entry &e = lookup(prhs[1]);
    const size_t n = (size_t)e.p.ny * e.p.nx, N = n * e.p.factor * e.p.factor, nf = (size_t)e.p.num_frames;
    if (cmd == "lowres") {
        need(nrhs, 3, "lowres");
        std::vector<float> f = to_float(prhs[2], n * nf, "low-resolution sequence");
        check(mmc_sf_set_lowres(e.h, f.data()), e.h);
    } else if (cmd == "blur") {
        need(nrhs, 3, "blur");
        if (mxIsEmpty(prhs[2])) { check(mmc_sf_set_blur(e.h, nullptr, 0, 0), e.h); return; }
        std::vector<float> k = to_float(prhs[2], mxGetNumberOfElements(prhs[2]), "blur kernel");
        check(mmc_sf_set_blur(e.h, k.data(), (int)mxGetM(prhs[2]), (int)mxGetN(prhs[2])), e.h);
    } else if (cmd == "flow") {
        need(nrhs, 3, "flow");
        std::vector<float> v = to_float(prhs[2], N * nf * 2, "flow field");
        std::vector<uint8_t> occ;
        if (nrhs > 3 && !mxIsEmpty(prhs[3])) {
            const mxArray *o = prhs[3];
            if (mxGetNumberOfElements(o) != N * nf) mexErrMsgIdAndTxt("MMC:sf", "occlusion map has the wrong size");
            occ.resize(N * nf);
            if (mxIsLogical(o)) {
                const mxLogical *m = mxGetLogicals(o);
                for (size_t i = 0; i < occ.size(); ++i) occ[i] = m[i] ? 1 : 0;
            } else if (mxIsDouble(o) && !mxIsSparse(o) && !mxIsComplex(o)) {
                const double *m = mxGetDoubles(o);
                for (size_t i = 0; i < occ.size(); ++i) occ[i] = m[i] != 0.0 ? 1 : 0;
            } else {
                mexErrMsgIdAndTxt("MMC:sf", "occlusion map must be a full logical or double array");
            }
        }
        check(mmc_sf_set_flow(e.h, v.data(), occ.empty() ? nullptr : occ.data()), e.h);
    } else if (cmd == "state") {
        need(nrhs, 3, "state");
        if (mxIsEmpty(prhs[2])) { check(mmc_sf_set_state(e.h, nullptr), e.h); return; }
        std::vector<float> u = to_float(prhs[2], N, "start point");
        check(mmc_sf_set_state(e.h, u.data()), e.h);
    } else if (cmd == "init") {
        check(mmc_sf_init(e.h), e.h);
    } else if (cmd == "solve") {
        need(nrhs, 4, "solve");
        mmc_stats st;
        check(mmc_sf_solve(e.h, (int)mxGetScalar(prhs[2]), (float)mxGetScalar(prhs[3]),
                           nrhs > 4 ? (int)mxGetScalar(prhs[4]) : 0, &st, progress, nullptr), e.h);
        std::vector<float> u(N);
        check(mmc_sf_get_state(e.h, u.data()), e.h);
        plhs[0] = mxCreateDoubleMatrix((mwSize)e.p.ny * e.p.factor, (mwSize)e.p.nx * e.p.factor, mxREAL);
        double *d = mxGetDoubles(plhs[0]);
        for (size_t i = 0; i < N; ++i) d[i] = u[i];
        if (nlhs > 1) {
            const char *names[] = {"iterations", "converged", "res_primal", "res_dual", "seconds"};
            plhs[1] = mxCreateStructMatrix(1, 1, 5, names);
            mxSetField(plhs[1], 0, "iterations", mxCreateDoubleScalar(st.iterations));
            mxSetField(plhs[1], 0, "converged", mxCreateDoubleScalar(st.converged));
            mxSetField(plhs[1], 0, "res_primal", mxCreateDoubleScalar(st.res_primal));
            mxSetField(plhs[1], 0, "res_dual", mxCreateDoubleScalar(st.res_dual));
            mxSetField(plhs[1], 0, "seconds", mxCreateDoubleScalar(st.solve_seconds));
        }
    } else {
        mexErrMsgIdAndTxt("MMC:sf", "unknown command '%s'", cmd.c_str());
    }
}
