// mmc_hip_mex.cpp -- MEX gateway between MultiframeMotionCoupling.m and libmmc_hip.so.
//
// Build (on a machine with MATLAB and ROCm; not built by this repository's tests):
//   mex -R2018a mex/mmc_hip_mex.cpp -Iinclude -Lsuperresolution_amd/lib -lmmc_hip
//
// The gateway holds no logic: it converts MATLAB doubles (column-major) to the
// float32 buffers the C ABI takes, forwards to one mmc_group_* call per command and
// turns a non-zero status into a MATLAB error.  A MATLAB session is one process, so
// the gateway always works on a group (include/mmc_hip.h): S.devices = [0 1 ... 7]
// spreads the sequence over the GPUs of the node, one frame block per entry; without
// that field it is one block on S.device.  Commands (first argument):
//   h = mmc_hip_mex('create', S)      S: struct with the fields of mmc_params (+ devices; + step_balance: a positive
//                                     factor, or adapt_steps = 1 for the library's residual balancing)
//   mmc_hip_mex('lowres', h, f)       f: ny x nx x nc double
//   mmc_hip_mex('blur',   h, k)       k: odd x odd double, or []
//   mmc_hip_mex('sampling', h, Sy, Sx) 1-D factors of samplingOperator, ny x Ny and nx x Nx
//   mmc_hip_mex('flow',   h, v, occ)  v: Ny x Nx x (nc-1) x 2 double, occ: logical / numeric or []
//   mmc_hip_mex('state',  h, u, w)    start point, Ny x Nx x nc double each; [] , []: bicubic
//   mmc_hip_mex('fixed',  h, u0, w0)  pinned first frame (u0_frame / w0_frame), [] , [] removes it
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

struct Entry { mmc_group *g; mmc_params p; };
std::map<uint64_t, Entry> g_handles;
uint64_t g_next = 1;

void at_exit() {
    for (auto &kv : g_handles) mmc_group_destroy(kv.second.g);
    g_handles.clear();
}

void check(int rc, const mmc_group *g) {
    if (rc != MMC_OK) mexErrMsgIdAndTxt("MMC:hip", "%s", mmc_group_last_error(g));
}

void need(int nrhs, int n, const char *cmd) {
    if (nrhs < n) mexErrMsgIdAndTxt("MMC:hip", "'%s' needs %d arguments", cmd, n);
}

Entry &lookup(const mxArray *a) {
    auto it = g_handles.find((uint64_t)mxGetScalar(a));
    if (it == g_handles.end()) mexErrMsgIdAndTxt("MMC:hip", "invalid handle");
    return it->second;
}

std::vector<float> to_float(const mxArray *a) {
    if (!mxIsDouble(a) || mxIsComplex(a) || mxIsSparse(a)) mexErrMsgIdAndTxt("MMC:hip", "expected a full real double array");
    const double *d = mxGetDoubles(a);
    const size_t n = mxGetNumberOfElements(a);
    return std::vector<float>(d, d + n);
}

void expect_numel(const mxArray *a, size_t n, const char *what) {
    if (mxGetNumberOfElements(a) != n) mexErrMsgIdAndTxt("MMC:hip", "%s has %zu elements, expected %zu", what, mxGetNumberOfElements(a), n);
}

double field(const mxArray *s, const char *name, double dflt) {
    const mxArray *f = mxGetField(s, 0, name);
    return (f && !mxIsEmpty(f)) ? mxGetScalar(f) : dflt;
}

mxArray *to_double(const std::vector<float> &v, mwSize Ny, mwSize Nx, mwSize nc) {
    const mwSize dims[3] = {Ny, Nx, nc};
    mxArray *out = mxCreateNumericArray(3, dims, mxDOUBLE_CLASS, mxREAL);
    double *d = mxGetDoubles(out);
    for (size_t i = 0; i < v.size(); ++i) d[i] = v[i];
    return out;
}

int progress(int it, const mmc_stats *st, void *) {
    // the progress lines prost prints 5 times per solve (MultiframeMotionCoupling.m:357-359)
    mexPrintf("  it %6d  energy %.6e  gap %.3e  res_primal %.3e  res_dual %.3e  (%.2f s)\n", it, st->primal_energy, st->gap,
              st->opt_res_primal, st->opt_res_dual, st->solve_seconds);
    return 0;
}

// occlusion map: logical, or any real numeric array (non-zero = occluded)
std::vector<uint8_t> to_mask(const mxArray *a) {
    const size_t n = mxGetNumberOfElements(a);
    std::vector<uint8_t> m(n);
    if (mxIsLogical(a)) {
        const mxLogical *o = mxGetLogicals(a);
        for (size_t i = 0; i < n; ++i) m[i] = o[i] ? 1 : 0;
    } else if (mxIsDouble(a) && !mxIsSparse(a) && !mxIsComplex(a)) {
        const double *o = mxGetDoubles(a);
        for (size_t i = 0; i < n; ++i) m[i] = o[i] != 0.0 ? 1 : 0;
    } else {
        mexErrMsgIdAndTxt("MMC:hip", "occlusion map must be a full logical or double array");
    }
    return m;
}

}  // namespace

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    if (nrhs < 1 || !mxIsChar(prhs[0])) mexErrMsgIdAndTxt("MMC:hip", "first argument must be a command string");
    char cmdbuf[32];
    mxGetString(prhs[0], cmdbuf, sizeof(cmdbuf));
    const std::string cmd(cmdbuf);
    // the handle table lives in this library: keep it loaded while handles exist, let go with the last one
    if (!mexIsLocked()) { mexAtExit(at_exit); }

    if (cmd == "create") {
        need(nrhs, 2, "create");
        const mxArray *s = prhs[1];
        if (!mxIsStruct(s)) mexErrMsgIdAndTxt("MMC:hip", "'create' takes a struct");
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
        // S.devices: one frame block per listed GPU (never more blocks than frames)
        std::vector<int32_t> devs;
        if (const mxArray *d = mxGetField(s, 0, "devices")) {
            if (!mxIsEmpty(d)) {
                if (!mxIsDouble(d)) mexErrMsgIdAndTxt("MMC:hip", "S.devices must be a double vector of device ordinals");
                const double *dv = mxGetDoubles(d);
                for (size_t i = 0; i < mxGetNumberOfElements(d); ++i) devs.push_back((int32_t)dv[i]);
            }
        }
        if (p.nc_global > 0 && (int)devs.size() > p.nc_global) devs.resize((size_t)p.nc_global);
        Entry e{nullptr, p};
        check(mmc_group_create(&p, devs.empty() ? 1 : (int)devs.size(), devs.empty() ? nullptr : devs.data(), &e.g), nullptr);
        const double bal = field(s, "step_balance", 1.0);
        if (bal != 1.0) check(mmc_group_set_step_balance(e.g, (float)bal), e.g);
        if (field(s, "adapt_steps", 0.0) != 0.0) check(mmc_group_set_step_adaptation(e.g, 1), e.g);
        g_handles[g_next] = e;
        if (!mexIsLocked()) mexLock();
        plhs[0] = mxCreateDoubleScalar((double)g_next++);
    } else if (cmd == "lowres") {
        need(nrhs, 3, "lowres");
        Entry &e = lookup(prhs[1]);
        expect_numel(prhs[2], (size_t)e.p.ny * e.p.nx * e.p.nc_global, "low-resolution sequence");
        std::vector<float> f = to_float(prhs[2]);
        check(mmc_group_set_lowres(e.g, f.data()), e.g);
    } else if (cmd == "blur") {
        need(nrhs, 3, "blur");
        Entry &e = lookup(prhs[1]);
        if (mxIsEmpty(prhs[2])) { check(mmc_group_set_blur(e.g, nullptr, 0, 0), e.g); return; }
        std::vector<float> k = to_float(prhs[2]);
        check(mmc_group_set_blur(e.g, k.data(), (int)mxGetM(prhs[2]), (int)mxGetN(prhs[2])), e.g);
    } else if (cmd == "sampling") {
        // mmc_hip_mex('sampling', handle, Sy, Sx): the two 1-D factors of samplingOperator
        // (full or sparse doubles, ny x Ny and nx x Nx); rows are handed over in banded form.
        need(nrhs, 4, "sampling");
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
            const size_t want_out = ax ? (size_t)e.p.nx : (size_t)e.p.ny;
            if (!mxIsDouble(S) || nout != want_out || nin != want_out * (size_t)e.p.factor)
                mexErrMsgIdAndTxt("MMC:hip", "sampling factor %d must be a %zu x %zu double matrix", ax + 1, want_out, want_out * (size_t)e.p.factor);
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
        check(mmc_group_set_sampling_taps(e.g, first[0].data(), w[0].data(), T[0], first[1].data(), w[1].data(), T[1]), e.g);
    } else if (cmd == "flow") {
        need(nrhs, 3, "flow");
        Entry &e = lookup(prhs[1]);
        const size_t N = (size_t)e.p.ny * e.p.factor * (size_t)e.p.nx * e.p.factor;
        const size_t npairs = e.p.nc_global > 1 ? (size_t)e.p.nc_global - 1 : 0;
        if (npairs == 0) { check(mmc_group_set_flow(e.g, nullptr, nullptr), e.g); return; }
        expect_numel(prhs[2], N * npairs * 2, "flow field");
        std::vector<float> v = to_float(prhs[2]);
        std::vector<uint8_t> occ;
        if (nrhs > 3 && !mxIsEmpty(prhs[3])) {
            expect_numel(prhs[3], N * npairs, "occlusion map");
            occ = to_mask(prhs[3]);
        }
        check(mmc_group_set_flow(e.g, v.data(), occ.empty() ? nullptr : occ.data()), e.g);
    } else if (cmd == "state") {
        need(nrhs, 3, "state");
        Entry &e = lookup(prhs[1]);
        if (mxIsEmpty(prhs[2])) { check(mmc_group_set_state(e.g, nullptr, nullptr), e.g); return; }   // bicubic, on the device
        const size_t M = (size_t)e.p.ny * e.p.factor * (size_t)e.p.nx * e.p.factor * (size_t)e.p.nc_global;
        expect_numel(prhs[2], M, "u");
        std::vector<float> u = to_float(prhs[2]), w;
        if (nrhs > 3 && !mxIsEmpty(prhs[3])) { expect_numel(prhs[3], M, "w"); w = to_float(prhs[3]); }
        check(mmc_group_set_state(e.g, u.data(), w.empty() ? nullptr : w.data()), e.g);
    } else if (cmd == "fixed") {
        need(nrhs, 4, "fixed");
        Entry &e = lookup(prhs[1]);
        if (mxIsEmpty(prhs[2]) && mxIsEmpty(prhs[3])) { check(mmc_group_set_fixed_frame(e.g, nullptr, nullptr), e.g); return; }
        const size_t N = (size_t)e.p.ny * e.p.factor * (size_t)e.p.nx * e.p.factor;
        std::vector<float> u0, w0;
        if (!mxIsEmpty(prhs[2])) { expect_numel(prhs[2], N, "u0_frame"); u0 = to_float(prhs[2]); }
        if (!mxIsEmpty(prhs[3])) { expect_numel(prhs[3], N, "w0_frame"); w0 = to_float(prhs[3]); }
        check(mmc_group_set_fixed_frame(e.g, u0.empty() ? nullptr : u0.data(), w0.empty() ? nullptr : w0.data()), e.g);
    } else if (cmd == "init") {
        need(nrhs, 2, "init");
        Entry &e = lookup(prhs[1]);
        check(mmc_group_init(e.g), e.g);
        float hv = 0.f;
        check(mmc_group_get_h(e.g, &hv), e.g);
        plhs[0] = mxCreateDoubleScalar(hv);
    } else if (cmd == "solve") {
        need(nrhs, 4, "solve");
        Entry &e = lookup(prhs[1]);
        mmc_stats st;
        check(mmc_group_solve(e.g, (int)mxGetScalar(prhs[2]), (float)mxGetScalar(prhs[3]),
                              nrhs > 4 ? (int)mxGetScalar(prhs[4]) : 0, &st, progress, nullptr), e.g);
        const mwSize Ny = (mwSize)e.p.ny * e.p.factor, Nx = (mwSize)e.p.nx * e.p.factor, nc = e.p.nc_global;
        const bool has_w = !(mxIsNaN(e.p.kappa) || e.p.kappa == 1.f);
        std::vector<float> u((size_t)Ny * Nx * nc), w(has_w ? u.size() : 0);
        check(mmc_group_get_state(e.g, u.data(), has_w ? w.data() : nullptr), e.g);
        plhs[0] = to_double(u, Ny, Nx, nc);
        if (nlhs > 1) plhs[1] = has_w ? to_double(w, Ny, Nx, nc) : mxCreateDoubleMatrix(0, 0, mxREAL);
        if (nlhs > 2) {
            const char *names[] = {"iterations", "converged", "res_primal", "res_dual", "seconds", "primal_energy", "dual_energy",
                                   "gap", "dual_infeasibility", "opt_res_primal", "opt_res_dual", "step_balance"};
            plhs[2] = mxCreateStructMatrix(1, 1, 12, names);
            mxSetField(plhs[2], 0, "step_balance", mxCreateDoubleScalar(st.step_balance));
            mxSetField(plhs[2], 0, "primal_energy", mxCreateDoubleScalar(st.primal_energy));
            mxSetField(plhs[2], 0, "dual_energy", mxCreateDoubleScalar(st.dual_energy));
            mxSetField(plhs[2], 0, "gap", mxCreateDoubleScalar(st.gap));
            mxSetField(plhs[2], 0, "dual_infeasibility", mxCreateDoubleScalar(st.dual_infeasibility));
            mxSetField(plhs[2], 0, "opt_res_primal", mxCreateDoubleScalar(st.opt_res_primal));
            mxSetField(plhs[2], 0, "opt_res_dual", mxCreateDoubleScalar(st.opt_res_dual));
            mxSetField(plhs[2], 0, "iterations", mxCreateDoubleScalar(st.iterations));
            mxSetField(plhs[2], 0, "converged", mxCreateDoubleScalar(st.converged));
            mxSetField(plhs[2], 0, "res_primal", mxCreateDoubleScalar(st.res_primal));
            mxSetField(plhs[2], 0, "res_dual", mxCreateDoubleScalar(st.res_dual));
            mxSetField(plhs[2], 0, "seconds", mxCreateDoubleScalar(st.solve_seconds));
        }
    } else if (cmd == "destroy") {
        need(nrhs, 2, "destroy");
        auto it = g_handles.find((uint64_t)mxGetScalar(prhs[1]));
        if (it != g_handles.end()) { mmc_group_destroy(it->second.g); g_handles.erase(it); }
        if (g_handles.empty() && mexIsLocked()) mexUnlock();     // `clear mex` may unload the gateway again
    } else {
        mexErrMsgIdAndTxt("MMC:hip", "unknown command '%s'", cmd.c_str());
    }
}
