// mmc_hip_mex.cpp -- MEX gateway between MultiframeMotionCoupling.m and libmmc_hip.so.
//
// Build (on a machine with MATLAB and ROCm; not built by this repository's tests):
//   mex -R2018a mex/mmc_hip_mex.cpp -Iinclude -Lsuperresolution_amd/lib -lmmc_hip
//
// The gateway holds no logic: it converts MATLAB doubles (column-major) to the
// float32 buffers the C ABI takes, forwards to one mmc_* call per command and
// turns a non-zero status into a MATLAB error.  Commands (first argument):
//   h = mmc_hip_mex('create', S)      S: struct with the fields of mmc_params
//   mmc_hip_mex('lowres', h, f)       f: ny x nx x nc double
//   mmc_hip_mex('blur',   h, k)       k: odd x odd double, or []
//   mmc_hip_mex('flow',   h, v, occ)  v: Ny x Nx x (nc-1) x 2 double, occ: logical or []
//   mmc_hip_mex('state',  h, u, w)    start point, Ny x Nx x nc double each
//   hval = mmc_hip_mex('init', h)     returns h (MultiframeMotionCoupling.m:437)
//   [u, w, stats] = mmc_hip_mex('solve', h, max_iters, tol, check_every)
//   mmc_hip_mex('destroy', h)
#include <cstdint>
#include <cstring>
#include <map>
#include <algorithm>
#include <string>
#include <vector>

#include "mex.h"
#include "mmc_hip.h"

namespace {

struct Entry { mmc_handle *h; mmc_params p; };
std::map<uint64_t, Entry> g_handles;
uint64_t g_next = 1;

void at_exit() {
    for (auto &kv : g_handles) mmc_destroy(kv.second.h);
    g_handles.clear();
}

void check(int rc, const mmc_handle *h) {
    if (rc != MMC_OK) mexErrMsgIdAndTxt("MMC:hip", "%s", mmc_last_error(h));
}

Entry &lookup(const mxArray *a) {
    auto it = g_handles.find((uint64_t)mxGetScalar(a));
    if (it == g_handles.end()) mexErrMsgIdAndTxt("MMC:hip", "invalid handle");
    return it->second;
}

std::vector<float> to_float(const mxArray *a) {
    if (!mxIsDouble(a)) mexErrMsgIdAndTxt("MMC:hip", "expected a double array");
    const double *d = mxGetDoubles(a);
    const size_t n = mxGetNumberOfElements(a);
    return std::vector<float>(d, d + n);
}

double field(const mxArray *s, const char *name, double dflt) {
    const mxArray *f = mxGetField(s, 0, name);
    return f ? mxGetScalar(f) : dflt;
}

mxArray *to_double(const std::vector<float> &v, mwSize Ny, mwSize Nx, mwSize nc) {
    const mwSize dims[3] = {Ny, Nx, nc};
    mxArray *out = mxCreateNumericArray(3, dims, mxDOUBLE_CLASS, mxREAL);
    double *d = mxGetDoubles(out);
    for (size_t i = 0; i < v.size(); ++i) d[i] = v[i];
    return out;
}

int progress(int it, const mmc_stats *st, void *) {
    mexPrintf("  it %6d  res_primal %.3e  res_dual %.3e  (%.2f s)\n", it, st->res_primal, st->res_dual, st->solve_seconds);
    return 0;
}

}  // namespace

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    if (nrhs < 1 || !mxIsChar(prhs[0])) mexErrMsgIdAndTxt("MMC:hip", "first argument must be a command string");
    char cmdbuf[32];
    mxGetString(prhs[0], cmdbuf, sizeof(cmdbuf));
    const std::string cmd(cmdbuf);
    mexAtExit(at_exit);
    mexLock();

    if (cmd == "create") {
        const mxArray *s = prhs[1];
        mmc_params p;
        std::memset(&p, 0, sizeof(p));
        p.size = sizeof(p);
        p.version = MMC_ABI_VERSION;
        p.ny = (int)field(s, "ny", 0);
        p.nx = (int)field(s, "nx", 0);
        p.factor = (int)field(s, "factor", 4);
        p.nc_global = (int)field(s, "nc", 0);
        p.frame_begin = 0;
        p.frame_end = p.nc_global;
        p.alpha = (float)field(s, "alpha", 0.01);
        p.kappa = (float)field(s, "kappa", 0.25);          // NaN selects the additive model
        p.h = (float)field(s, "h", 0.0);
        p.flow_direction = (int)field(s, "flow_direction", MMC_FLOW_BACKWARD);
        p.sampling = (int)field(s, "sampling", MMC_SAMPLING_AVERAGE);
        p.device = (int)field(s, "device", 0);
        Entry e{nullptr, p};
        check(mmc_create(&p, &e.h), nullptr);
        g_handles[g_next] = e;
        plhs[0] = mxCreateDoubleScalar((double)g_next++);
    } else if (cmd == "lowres") {
        Entry &e = lookup(prhs[1]);
        std::vector<float> f = to_float(prhs[2]);
        check(mmc_set_lowres(e.h, f.data()), e.h);
    } else if (cmd == "blur") {
        Entry &e = lookup(prhs[1]);
        if (mxIsEmpty(prhs[2])) { check(mmc_set_blur(e.h, nullptr, 0, 0), e.h); return; }
        std::vector<float> k = to_float(prhs[2]);
        check(mmc_set_blur(e.h, k.data(), (int)mxGetM(prhs[2]), (int)mxGetN(prhs[2])), e.h);
    } else if (cmd == "sampling") {
        // mmc_hip_mex('sampling', handle, Sy, Sx): the two 1-D factors of samplingOperator
        // (full or sparse doubles, ny x Ny and nx x Nx); rows are handed over in banded form.
        Entry &e = lookup(prhs[1]);
        std::vector<int32_t> first[2];
        std::vector<float> w[2];
        int T[2] = {0, 0};
        for (int ax = 0; ax < 2; ++ax) {
            mxArray *full = nullptr;
            const mxArray *S = prhs[2 + ax];
            if (mxIsSparse(S)) {
                mxArray *in = const_cast<mxArray *>(S);
                mexCallMATLAB(1, &full, 1, &in, "full");
                S = full;
            }
            const size_t nout = mxGetM(S), nin = mxGetN(S);
            const double *d = mxGetDoubles(S);
            std::vector<int> lo(nout, (int)nin), hi(nout, -1);
            for (size_t c = 0; c < nin; ++c)
                for (size_t r = 0; r < nout; ++r)
                    if (d[r + nout * c] != 0.0) { lo[r] = std::min(lo[r], (int)c); hi[r] = std::max(hi[r], (int)c); }
            for (size_t r = 0; r < nout; ++r) if (hi[r] >= lo[r]) T[ax] = std::max(T[ax], hi[r] - lo[r] + 1);
            if (T[ax] == 0) mexErrMsgIdAndTxt("MMC:hip", "empty sampling matrix");
            first[ax].resize(nout);
            w[ax].assign(nout * (size_t)T[ax], 0.f);
            for (size_t r = 0; r < nout; ++r) {
                first[ax][r] = std::max(std::min(lo[r], (int)nin - T[ax]), 0);
                for (int m = 0; m < T[ax]; ++m) w[ax][r * T[ax] + m] = (float)d[r + nout * (size_t)(first[ax][r] + m)];
            }
            if (full) mxDestroyArray(full);
        }
        check(mmc_set_sampling_taps(e.h, first[0].data(), w[0].data(), T[0], first[1].data(), w[1].data(), T[1]), e.h);
    } else if (cmd == "flow") {
        Entry &e = lookup(prhs[1]);
        std::vector<float> v = to_float(prhs[2]);
        std::vector<uint8_t> occ;
        if (nrhs > 3 && !mxIsEmpty(prhs[3])) {
            const mxLogical *o = mxGetLogicals(prhs[3]);
            occ.assign(o, o + mxGetNumberOfElements(prhs[3]));
        }
        check(mmc_set_flow(e.h, v.data(), occ.empty() ? nullptr : occ.data()), e.h);
    } else if (cmd == "state") {
        Entry &e = lookup(prhs[1]);
        std::vector<float> u = to_float(prhs[2]), w;
        if (nrhs > 3 && !mxIsEmpty(prhs[3])) w = to_float(prhs[3]);
        check(mmc_set_state(e.h, u.data(), w.empty() ? nullptr : w.data()), e.h);
    } else if (cmd == "init") {
        Entry &e = lookup(prhs[1]);
        check(mmc_init(e.h), e.h);
        float hv = 0.f;
        check(mmc_get_h(e.h, &hv), e.h);
        plhs[0] = mxCreateDoubleScalar(hv);
    } else if (cmd == "solve") {
        Entry &e = lookup(prhs[1]);
        mmc_stats st;
        check(mmc_solve(e.h, (int)mxGetScalar(prhs[2]), (float)mxGetScalar(prhs[3]),
                        nrhs > 4 ? (int)mxGetScalar(prhs[4]) : 0, &st, progress, nullptr), e.h);
        const mwSize Ny = (mwSize)e.p.ny * e.p.factor, Nx = (mwSize)e.p.nx * e.p.factor, nc = e.p.nc_global;
        const bool has_w = !(mxIsNaN(e.p.kappa) || e.p.kappa == 1.f);
        std::vector<float> u((size_t)Ny * Nx * nc), w(has_w ? u.size() : 0);
        check(mmc_get_state(e.h, u.data(), has_w ? w.data() : nullptr), e.h);
        plhs[0] = to_double(u, Ny, Nx, nc);
        if (nlhs > 1) plhs[1] = has_w ? to_double(w, Ny, Nx, nc) : mxCreateDoubleMatrix(0, 0, mxREAL);
        if (nlhs > 2) {
            const char *names[] = {"iterations", "converged", "res_primal", "res_dual", "seconds"};
            plhs[2] = mxCreateStructMatrix(1, 1, 5, names);
            mxSetField(plhs[2], 0, "iterations", mxCreateDoubleScalar(st.iterations));
            mxSetField(plhs[2], 0, "converged", mxCreateDoubleScalar(st.converged));
            mxSetField(plhs[2], 0, "res_primal", mxCreateDoubleScalar(st.res_primal));
            mxSetField(plhs[2], 0, "res_dual", mxCreateDoubleScalar(st.res_dual));
            mxSetField(plhs[2], 0, "seconds", mxCreateDoubleScalar(st.solve_seconds));
        }
    } else if (cmd == "destroy") {
        auto it = g_handles.find((uint64_t)mxGetScalar(prhs[1]));
        if (it != g_handles.end()) { mmc_destroy(it->second.h); g_handles.erase(it); }
    } else {
        mexErrMsgIdAndTxt("MMC:hip", "unknown command '%s'", cmd.c_str());
    }
}
