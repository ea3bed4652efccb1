/* A plain C client of libmmc_hip.so: what a MEX gateway or any other foreign-function binding
 * does, without Python in between.  Reads a problem from a raw float32 file, runs the solve
 * through the calls of include/mmc_hip.h and writes u, w and the final residuals.
 *
 * usage: capi_client <in.bin> <out.bin>
 * in.bin : int32 header {ny, nx, nc, factor, kh, kw, iters, direction, have_occ}, float32 {alpha, kappa},
 *          then float32 f[ny*nx*nc], k[kh*kw], v[Ny*Nx*(nc-1)*2], u0[Ny*Nx*nc], w0[Ny*Nx*nc],
 *          then uint8 occ[Ny*Nx*(nc-1)] if have_occ
 * out.bin: float32 u[Ny*Nx*nc], w[Ny*Nx*nc], float64 {h, res_primal, res_dual, energy}
 */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "../../include/mmc_hip.h"

/* Last call entered, and a backtrace on a fatal signal: a crash inside the library or the HIP
 * runtime then names its place in the test log instead of a bare exit status. */
static const char *volatile g_phase = "start";
static void on_fatal(int sig) {
    void *bt[64];
    const char *msg = "capi_client: fatal signal in phase: ";
    if (write(2, msg, strlen(msg)) < 0) _exit(128 + sig);
    if (write(2, g_phase, strlen(g_phase)) < 0 || write(2, "\n", 1) < 0) _exit(128 + sig);
    backtrace_symbols_fd(bt, backtrace(bt, 64), 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

#define CHECK(call)                                                                   \
    do {                                                                              \
        g_phase = #call;                                                              \
        int rc_ = (call);                                                             \
        if (rc_ != MMC_OK) {                                                          \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, mmc_last_error(h));   \
            return 2;                                                                 \
        }                                                                             \
    } while (0)

static void *read_block(FILE *fp, size_t bytes) {
    void *p = malloc(bytes ? bytes : 1);
    if (!p || fread(p, 1, bytes, fp) != bytes) { fprintf(stderr, "short read\n"); exit(3); }
    return p;
}

int main(int argc, char **argv) {
    signal(SIGSEGV, on_fatal);
    signal(SIGBUS, on_fatal);
    signal(SIGABRT, on_fatal);
    if (argc != 3) { fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 1; }
    FILE *fp = fopen(argv[1], "rb");
    if (!fp) { perror(argv[1]); return 1; }
    int32_t hd[9];
    float sc[2];
    if (fread(hd, sizeof(int32_t), 9, fp) != 9 || fread(sc, sizeof(float), 2, fp) != 2) return 3;
    const int ny = hd[0], nx = hd[1], nc = hd[2], factor = hd[3], kh = hd[4], kw = hd[5], iters = hd[6];
    const size_t n = (size_t)ny * nx, N = n * factor * factor, np = nc > 1 ? (size_t)(nc - 1) : 0;
    float *f = read_block(fp, n * nc * sizeof(float));
    float *k = read_block(fp, (size_t)kh * kw * sizeof(float));
    float *v = read_block(fp, N * np * 2 * sizeof(float));
    float *u0 = read_block(fp, N * nc * sizeof(float));
    float *w0 = read_block(fp, N * nc * sizeof(float));
    uint8_t *occ = hd[8] ? read_block(fp, N * np) : NULL;
    fclose(fp);

    mmc_handle *h = NULL;
    mmc_params p;
    memset(&p, 0, sizeof(p));
    p.size = sizeof(p);
    p.version = MMC_ABI_VERSION;
    p.ny = ny; p.nx = nx; p.factor = factor;
    p.nc_global = nc; p.frame_begin = 0; p.frame_end = nc;
    p.alpha = sc[0]; p.kappa = sc[1]; p.h = 0.f;
    p.flow_direction = hd[7];
    p.sampling = MMC_SAMPLING_AVERAGE;
    p.device = 0;
    p.stream = NULL;
    CHECK(mmc_create(&p, &h));
    CHECK(mmc_set_lowres(h, f));
    CHECK(mmc_set_blur(h, kh > 0 ? k : NULL, kh, kw));
    CHECK(mmc_set_flow(h, v, occ));
    CHECK(mmc_set_state(h, u0, w0));   /* w0 is ignored by the models without w */
    CHECK(mmc_init(h));
    float hval = 0.f;
    CHECK(mmc_get_h(h, &hval));
    mmc_stats st;
    memset(&st, 0, sizeof(st));
    CHECK(mmc_solve(h, iters, 0.f, 0, &st, NULL, NULL));
    /* kappa = 1 and kappa = NaN are the models without w (MultiframeMotionCoupling.m:500-533) */
    const int has_w = !(sc[1] != sc[1] || sc[1] == 1.f);
    float *u = malloc(N * nc * sizeof(float)), *w = calloc(N * nc, sizeof(float));
    CHECK(mmc_get_state(h, u, has_w ? w : NULL));
    double energy = 0.0;
    CHECK(mmc_energy(h, &energy));
    CHECK(mmc_destroy(h));
    g_phase = "after mmc_destroy";

    FILE *out = fopen(argv[2], "wb");
    if (!out) { perror(argv[2]); return 1; }
    double tail[4] = {(double)hval, st.res_primal, st.res_dual, energy};
    fwrite(u, sizeof(float), N * nc, out);
    fwrite(w, sizeof(float), N * nc, out);
    fwrite(tail, sizeof(double), 4, out);
    fclose(out);
    printf("capi_client: %d iterations, h = %g, res_primal = %.3e, res_dual = %.3e\n", st.iterations, hval, st.res_primal, st.res_dual);
    return 0;
}
