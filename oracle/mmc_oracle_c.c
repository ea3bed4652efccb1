/*
 * mmc_oracle_c.c -- C/OpenMP restatement of the Multiframe Motion Coupling PDHG
 * solve.  TEST INFRASTRUCTURE ONLY: it is the second checker next to
 * oracle/mmc_oracle.py and the CPU baseline timed by bench.py; nothing in the
 * product package links or loads it.  PARITY UNPINNED against real
 * prost/flexBox output (the reference cannot run here and ships no vectors); it
 * is validated against oracle/mmc_oracle.py in tests/test_oracle_c.py.
 *
 * It follows what the reference hands to flexBox_CPP (explicit sparse operators,
 * MultiframeMotionCoupling.m:724-745): W and W' (helpers/constructWarpFF.m,
 * constructWarpFB.m) and A = D*B and A' (helpers/samplingOperator.m,
 * helpers/RepConvMtx.m) are stored as sparse rows, the gradient
 * (spmat_gradient2d) is applied matrix-free, the problem is the one of
 * init_u_prost (MultiframeMotionCoupling.m:480-533), the iteration and the step
 * sizes are those fixed in oracle/mmc_oracle.py (DECISION comments there).
 * float32 arithmetic throughout, double only for the reported sums.
 *
 * Two modes (mmco_params.matrix_free): 0 stores the operators as sparse rows as above; 1 keeps
 * nothing but the inputs: taps are recomputed from the flow at every use, W' is a scatter with
 * atomic float adds (so its sums depend on the thread schedule in the last bits), D*B and its
 * adjoint are loops over the composite kernel with the replicate border folded in.
 *
 * Build: see oracle/Makefile (gcc -O3 -fopenmp -shared).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct mmco_params {
    int ny, nx, factor, nc;
    float alpha, kappa, h;      /* kappa NaN: additive model; 1: single-dual model; h <= 0: compute */
    int flow_direction;         /* 0 forward 'F-I', 1 backward 'I-F', 2 forward-backward */
    int sampling;               /* 0 'average' (factor 4), 1 'stride' */
    int kh, kw;                 /* blur kernel size, 0 = no blur */
    int matrix_free;            /* 0: W', A, A' as stored sparse rows; 1: nothing stored, see the header */
} mmco_params;

typedef struct { int *ptr, *idx; float *val; int rows; } csr;

static void csr_free(csr *m) { free(m->ptr); free(m->idx); free(m->val); memset(m, 0, sizeof(*m)); }

/* transpose of a rows x cols CSR by counting sort */
static csr csr_transpose(const csr *a, int cols) {
    csr t;
    int nnz = a->ptr[a->rows], i, k;
    t.rows = cols;
    t.ptr = (int *)calloc((size_t)cols + 1, sizeof(int));
    t.idx = (int *)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(int));
    t.val = (float *)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(float));
    for (k = 0; k < nnz; ++k) t.ptr[a->idx[k] + 1]++;
    for (i = 0; i < cols; ++i) t.ptr[i + 1] += t.ptr[i];
    {
        int *fill = (int *)malloc((size_t)cols * sizeof(int));
        memcpy(fill, t.ptr, (size_t)cols * sizeof(int));
        for (i = 0; i < a->rows; ++i)
            for (k = a->ptr[i]; k < a->ptr[i + 1]; ++k) {
                int c = a->idx[k], pos = fill[c]++;
                t.idx[pos] = i;
                t.val[pos] = a->val[k];
            }
        free(fill);
    }
    return t;
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* A = D*B for one frame: row (R,C) of samplingOperator * RepConvMtx with the replicate border of
 * RepConvMtx.m:23-32; taps that clamp onto the same pixel are merged (sparse() sums duplicates, :38). */
static csr build_A(const mmco_params *P, const float *k) {
    const int f = P->factor, Ny = P->ny * f, Nx = P->nx * f;
    const int td = P->sampling == 0 ? f : 1;
    const int kh = P->kh > 0 ? P->kh : 1, kw = P->kw > 0 ? P->kw : 1;
    const int ker = kh / 2, kec = kw / 2;
    const int maxrow = td * td * kh * kw;
    const float wd = 1.0f / (float)(td * td);
    csr A;
    int R, C, n = P->ny * P->nx, nnz = 0;
    A.rows = n;
    A.ptr = (int *)calloc((size_t)n + 1, sizeof(int));
    A.idx = (int *)malloc((size_t)n * maxrow * sizeof(int));
    A.val = (float *)malloc((size_t)n * maxrow * sizeof(float));
    for (C = 0; C < P->nx; ++C)
        for (R = 0; R < P->ny; ++R) {
            const int row = R + P->ny * C, start = nnz;
            int a, b, i, j, e;
            for (b = 0; b < td; ++b)
                for (a = 0; a < td; ++a)
                    for (j = -kec; j <= kec; ++j)
                        for (i = -ker; i <= ker; ++i) {
                            /* (B u)(r,c) = sum k(kc-i, kc-j) u(clamp(r+i), clamp(c+j)), RepConvMtx.m:33 */
                            const float kv = P->kh > 0 ? k[(ker - i) + kh * (kec - j)] : 1.0f;
                            const int rr = clampi(f * R + a + i, 0, Ny - 1), cc = clampi(f * C + b + j, 0, Nx - 1);
                            const int col = rr + Ny * cc;
                            for (e = start; e < nnz; ++e)
                                if (A.idx[e] == col) break;
                            if (e == nnz) { A.idx[nnz] = col; A.val[nnz] = 0.0f; nnz++; }
                            A.val[e] += wd * kv;
                        }
            A.ptr[row + 1] = nnz;
            (void)row;
        }
    return A;
}

/* Composite kernel of sampling taps (td equal weights from f*R) and the flipped blur: A x (R,C) =
 * sum_{m,n} ck[m + CH*n] x(clamp(f*R + oy + m), clamp(f*C + ox + n)); one clamp is exact here because
 * the sampling taps themselves never leave the frame. */
typedef struct { int CH, CW, oy, ox; float *ck; } composite;

static composite build_composite(const mmco_params *P, const float *k) {
    const int f = P->factor, td = P->sampling == 0 ? f : 1;
    const int kh = P->kh > 0 ? P->kh : 1, kw = P->kw > 0 ? P->kw : 1;
    composite c;
    int a, b, i, j;
    c.CH = td + kh - 1; c.CW = td + kw - 1; c.oy = -(kh / 2); c.ox = -(kw / 2);
    c.ck = (float *)calloc((size_t)c.CH * c.CW, sizeof(float));
    for (b = 0; b < td; ++b)
        for (a = 0; a < td; ++a)
            for (j = 0; j < kw; ++j)
                for (i = 0; i < kh; ++i)
                    c.ck[(a + i) + c.CH * (b + j)] += (P->kh > 0 ? k[(kh - 1 - i) + kh * (kw - 1 - j)] : 1.0f) / (float)(td * td);
    return c;
}

static float apply_A_mf(const composite *c, int f, int Ny, int Nx, const float *x, int R, int C) {
    float acc = 0.0f;
    int m, n;
    for (n = 0; n < c->CW; ++n) {
        const int cc = clampi(f * C + c->ox + n, 0, Nx - 1);
        const float *col = x + (size_t)Ny * cc;
        for (m = 0; m < c->CH; ++m) acc += c->ck[m + c->CH * n] * col[clampi(f * R + c->oy + m, 0, Ny - 1)];
    }
    return acc;
}

/* (A' p)(r,c) in gather form; a border pixel also collects the taps of the virtual rows / columns
 * outside the frame that the replicate border folds onto it.  absolute != 0: sum of |weights| (p == NULL). */
static float apply_AT_mf(const composite *c, int f, int ny, int nx, int Ny, int Nx, const float *p, int r, int col, int absolute) {
    const int rlo = r == 0 ? (c->oy < 0 ? c->oy : 0) : r, clo = col == 0 ? (c->ox < 0 ? c->ox : 0) : col;
    int rhi = r, chi = col, rv, cv;
    float acc = 0.0f;
    if (r == Ny - 1) { rhi = f * (ny - 1) + c->oy + c->CH - 1; if (rhi < Ny - 1) rhi = Ny - 1; }
    if (col == Nx - 1) { chi = f * (nx - 1) + c->ox + c->CW - 1; if (chi < Nx - 1) chi = Nx - 1; }
    for (cv = clo; cv <= chi; ++cv) {
        const int numc = cv - c->ox - c->CW + 1;
        int Clo = numc > 0 ? (numc + f - 1) / f : 0, Chi = (cv - c->ox) / f, C;
        if (Chi > nx - 1) Chi = nx - 1;
        for (C = Clo; C <= Chi; ++C) {
            const int nn = cv - f * C - c->ox;
            for (rv = rlo; rv <= rhi; ++rv) {
                const int numr = rv - c->oy - c->CH + 1;
                int Rlo = numr > 0 ? (numr + f - 1) / f : 0, Rhi = (rv - c->oy) / f, R;
                if (Rhi > ny - 1) Rhi = ny - 1;
                for (R = Rlo; R <= Rhi; ++R) {
                    const float wgt = c->ck[(rv - f * R - c->oy) + c->CH * nn];
                    acc += absolute ? fabsf(wgt) : wgt * p[R + ny * C];
                }
            }
        }
    }
    return acc;
}

typedef struct { int idx[4]; float w[4]; int valid; } taps;

/* bilinear taps of warpingOperator as fixed in oracle/mmc_oracle.py: float32 position arithmetic,
 * row masked when the sample leaves the closed pixel domain or is NaN, +1 tap index clamped */
static taps make_taps(int Ny, int Nx, int r, int c, float vx, float vy) {
    taps t;
    float y = (float)r + vy, x = (float)c + vx, y0, x0, a, b, na, nb;
    int r0, c0, r1, c1;
    t.valid = (y >= 0.0f) && (y <= (float)(Ny - 1)) && (x >= 0.0f) && (x <= (float)(Nx - 1));
    if (!t.valid) { y = 0.0f; x = 0.0f; }
    y0 = floorf(y); x0 = floorf(x);
    a = y - y0; b = x - x0; na = 1.0f - a; nb = 1.0f - b;
    r0 = (int)y0; c0 = (int)x0;
    r1 = r0 + 1 < Ny ? r0 + 1 : Ny - 1;
    c1 = c0 + 1 < Nx ? c0 + 1 : Nx - 1;
    t.idx[0] = r0 + Ny * c0; t.idx[1] = r1 + Ny * c0; t.idx[2] = r0 + Ny * c1; t.idx[3] = r1 + Ny * c1;
    t.w[0] = na * nb; t.w[1] = a * nb; t.w[2] = na * b; t.w[3] = a * b;
    return t;
}

/* squared distance of d from the normal cone of the ball of radius R at g ({0} inside, {lambda g} on the sphere) */
static double ball_res2(float dt, float dx, float dy, float gt, float gx, float gy, float R) {
    const float n2 = gt * gt + gx * gx + gy * gy;
    float lam = 0.0f, rt, rx, ry;
    if (n2 > 0.0f && n2 >= R * R * (1.0f - 1e-5f)) { lam = (dt * gt + dx * gx + dy * gy) / n2; if (lam < 0.0f) lam = 0.0f; }
    rt = dt - lam * gt; rx = dx - lam * gx; ry = dy - lam * gy;
    return (double)rt * rt + (double)rx * rx + (double)ry * ry;
}
/* ... of the interval [-b, b] at y */
static double box_res2(float d, float y, float b) {
    const float r = y >= b ? (d < 0.0f ? d : 0.0f) : (y <= -b ? (d > 0.0f ? d : 0.0f) : d);
    return (double)r * r;
}

static int pair_is_IF(int dir, int q) { return dir == 1 ? 1 : (dir == 0 ? 0 : ((q & 1) == 0)); }

/* taps of pixel j of pair q: stored (tp != NULL) or recomputed from the flow */
static taps get_taps(const taps *tp, const float *v, const unsigned char *occ, int npairs, int Ny, int Nx, int q, int j) {
    const size_t N = (size_t)Ny * Nx;
    taps t;
    if (tp) return tp[(size_t)q * N + j];
    t = make_taps(Ny, Nx, j % Ny, j / Ny, v[(size_t)q * N + j], v[(size_t)(npairs + q) * N + j]);
    if (occ && occ[(size_t)q * N + j]) t.valid = 0;
    return t;
}

/* Returns 0 on success.  u, w: start point in, result out (w may be NULL for the models without w).
 * out[0..3] = sum(x+-x)^2, sum(x+)^2, sum(y+-y)^2, sum(y+)^2 of the last iteration, out[4] = h,
 * out[5] = seconds spent in the iteration loop, out[6] = threads used.
 * out[8..16]: the model's optimality measures at the final (x, y), evaluated in one extra read-only pass
 * (they depend on K, f and the radii only, not on the step sizes):
 *   [8] regulariser energy  [9] data energy ||Au-f||_1  [10] |K x|^2  [11] dist(K x, dF*(y))^2
 *   [12] sum max(0,-(K'y)_u)  [13] sum |(K'y)_w|  [14] |K'y|^2  [15] dist(-K'y, dG(x))^2  [16] <f,p>
 * primal energy = [8]+[9]; dual objective (without the condition (K'y)_w = 0) = -[16] - [12]. */
int mmco_solve(const mmco_params *P, const float *f_lr, const float *v, const unsigned char *occ,
               const float *k, float *u, float *w, int iters, int nthreads, double *out) {
    const int f = P->factor, ny = P->ny, nx = P->nx, nc = P->nc, Ny = ny * f, Nx = nx * f;
    const int N = Ny * Nx, n = ny * nx, npairs = nc > 1 ? nc - 1 : 0;
    const size_t M = (size_t)N * nc;
    const int model = isnan(P->kappa) ? 2 : (P->kappa == 1.0f ? 1 : 0);
    const float kap = model == 0 ? P->kappa : 1.0f;
    const float rad1 = model == 1 ? 2.0f * P->alpha : P->alpha, rad2 = P->alpha;
    csr A, AT, *WT;
    taps *tp;
    float *cW, *cA, *sigp, *ub, *wb, *p, *g1, *g2, h, inv_h;
    double s_dx = 0, s_x = 0, s_dy = 0, s_y = 0, t_loop = 0;
    double d_reg = 0, d_data = 0, d_kx2 = 0, d_rd2 = 0, d_box = 0, d_inf = 0, d_kty2 = 0, d_rp2 = 0, d_lin = 0;
    int q, it;
    long long jj;
    if ((P->sampling == 0 && f != 4) || (P->kh > 0 && ((P->kh & 1) == 0 || (P->kw & 1) == 0))) return -1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    const int mf = P->matrix_free != 0;
    composite ck = build_composite(P, k);
    float *wt1 = NULL, *wt2 = NULL;
    memset(&A, 0, sizeof(A)); memset(&AT, 0, sizeof(AT));
    tp = NULL; WT = NULL;
    cW = (float *)calloc(M, sizeof(float));
    cA = (float *)calloc((size_t)N, sizeof(float));
    sigp = (float *)malloc((size_t)n * sizeof(float));
    if (!mf) {
    A = build_A(P, k);
    AT = csr_transpose(&A, N);
    tp = (taps *)malloc((size_t)(npairs > 0 ? npairs : 1) * N * sizeof(taps));
    WT = (csr *)calloc((size_t)(npairs > 0 ? npairs : 1), sizeof(csr));
    /* constructWarpFF.m:22-38: taps, out-of-range / occlusion marker; W' as sparse rows per pair */
    for (q = 0; q < npairs; ++q) {
        csr Wq;
        int j, e, nnz = 0;
        Wq.rows = N;
        Wq.ptr = (int *)calloc((size_t)N + 1, sizeof(int));
        Wq.idx = (int *)malloc((size_t)4 * N * sizeof(int));
        Wq.val = (float *)malloc((size_t)4 * N * sizeof(float));
        for (j = 0; j < N; ++j) {
            taps t = make_taps(Ny, Nx, j % Ny, j / Ny, v[(size_t)q * N + j], v[(size_t)(npairs + q) * N + j]);
            if (occ && occ[(size_t)q * N + j]) t.valid = 0;
            tp[(size_t)q * N + j] = t;
            if (t.valid)
                for (e = 0; e < 4; ++e) { Wq.idx[nnz] = t.idx[e]; Wq.val[nnz] = t.w[e]; nnz++; }
            Wq.ptr[j + 1] = nnz;
        }
        WT[q] = csr_transpose(&Wq, N);
        {   /* column abs-sums of the coupling operator: warp part on the warped frame, -Id on the other */
            const int isIF = pair_is_IF(P->flow_direction, q);
            float *cw_w = cW + (size_t)(isIF ? q + 1 : q) * N, *cw_i = cW + (size_t)(isIF ? q : q + 1) * N;
            for (j = 0; j < N; ++j) {
                for (e = WT[q].ptr[j]; e < WT[q].ptr[j + 1]; ++e) cw_w[j] += WT[q].val[e];
                if (tp[(size_t)q * N + j].valid) cw_i[j] += 1.0f;
            }
        }
        csr_free(&Wq);
    }
    /* abs row / column sums of A */
    {
        int i, e;
        for (i = 0; i < n; ++i) {
            float s = 0.0f;
            for (e = A.ptr[i]; e < A.ptr[i + 1]; ++e) { s += fabsf(A.val[e]); cA[A.idx[e]] += fabsf(A.val[e]); }
            sigp[i] = s > 0.0f ? 1.0f / s : 1.0f;
        }
    }
    } else {
        /* the same sums without stored operators.  Taps that clamp onto one pixel are summed before the
         * absolute value is taken, as in the stored rows; with a non-negative composite kernel (block mean
         * times a non-negative blur) that is the plain sum, which is all this mode supports. */
        int j, e;
        for (j = 0; j < ck.CH * ck.CW; ++j) if (ck.ck[j] < 0.0f) { free(ck.ck); free(cW); free(cA); free(sigp); return -2; }
        wt1 = (float *)malloc(M * sizeof(float));
        wt2 = (float *)malloc(M * sizeof(float));
        for (q = 0; q < npairs; ++q) {
            const int isIF = pair_is_IF(P->flow_direction, q);
            float *cw_w = cW + (size_t)(isIF ? q + 1 : q) * N, *cw_i = cW + (size_t)(isIF ? q : q + 1) * N;
            for (j = 0; j < N; ++j) {
                const taps t = get_taps(NULL, v, occ, npairs, Ny, Nx, q, j);
                if (!t.valid) continue;
                for (e = 0; e < 4; ++e) cw_w[t.idx[e]] += t.w[e];
                cw_i[j] += 1.0f;
            }
        }
#pragma omp parallel for schedule(static)
        for (jj = 0; jj < (long long)N; ++jj) cA[jj] = apply_AT_mf(&ck, f, ny, nx, Ny, Nx, NULL, (int)(jj % Ny), (int)(jj / Ny), 1);
        {
            float s = 0.0f;
            for (j = 0; j < ck.CH * ck.CW; ++j) s += ck.ck[j];
            for (j = 0; j < n; ++j) sigp[j] = s > 0.0f ? 1.0f / s : 1.0f;
        }
    }
    ub = (float *)malloc(M * sizeof(float));
    wb = (float *)calloc(M, sizeof(float));
    memcpy(ub, u, M * sizeof(float));
    if (model == 0) memcpy(wb, w, M * sizeof(float));
    p = (float *)calloc((size_t)n * nc, sizeof(float));
    g1 = (float *)calloc(3 * M, sizeof(float));
    g2 = (float *)calloc(3 * M, sizeof(float));

    /* h = (sum|W u|/M) / (sum|G u|/M) on the start point, MultiframeMotionCoupling.m:431-437 */
    h = P->h;
    if (!(h > 0.0f)) {
        double sw = 0.0, sg = 0.0;
#pragma omp parallel for reduction(+ : sw, sg) schedule(static)
        for (jj = 0; jj < (long long)M; ++jj) {
            const int t = (int)(jj / N), j = (int)(jj % N), r = j % Ny, c = j / Ny;
            const float *ut = u + (size_t)t * N;
            if (c < Nx - 1) sg += fabs((double)(ut[j + Ny] - ut[j]));
            if (r < Ny - 1) sg += fabs((double)(ut[j + 1] - ut[j]));
            const taps th = t < npairs ? get_taps(tp, v, occ, npairs, Ny, Nx, t, j) : (taps){{0, 0, 0, 0}, {0, 0, 0, 0}, 0};
            if (t < npairs && th.valid) {
                const taps *tt = &th;
                const int isIF = pair_is_IF(P->flow_direction, t);
                const float *uw = u + (size_t)(isIF ? t + 1 : t) * N, *ui = u + (size_t)(isIF ? t : t + 1) * N;
                sw += fabs((double)(tt->w[0] * uw[tt->idx[0]] + tt->w[1] * uw[tt->idx[1]] + tt->w[2] * uw[tt->idx[2]] +
                                    tt->w[3] * uw[tt->idx[3]] - ui[j]));
            }
        }
        h = nc == 1 ? 1.0f : (float)(sw / sg);
    }
    inv_h = 1.0f / h;

    {
#ifdef _OPENMP
        double t0 = omp_get_wtime();
#endif
        for (it = 0; it <= iters; ++it) {
            const int last = it == iters - 1;
            const int eval = it == iters;       /* extra read-only pass: optimality measures at the final (x, y) */
            const float *xsrc = eval ? u : ub, *zsrc = (eval && model == 0) ? w : wb;
            double a_dx = 0, a_x = 0, a_dy = 0, a_y = 0;
            double e_reg = 0, e_data = 0, e_kx2 = 0, e_rd2 = 0, e_box = 0, e_inf = 0, e_kty2 = 0, e_rp2 = 0, e_lin = 0;
#ifdef _OPENMP
            if (eval) t_loop = omp_get_wtime() - t0;
#endif
            if (mf) {
                /* W' y as a scatter: every unmasked row adds its four taps into the warped frame */
                memset(wt1, 0, M * sizeof(float));
                if (model == 0) memset(wt2, 0, M * sizeof(float));
#pragma omp parallel for schedule(static)
                for (jj = 0; jj < (long long)npairs * N; ++jj) {
                    const int qq = (int)(jj / N), j = (int)(jj % N);
                    const taps t = get_taps(NULL, v, occ, npairs, Ny, Nx, qq, j);
                    const size_t ow = (size_t)(pair_is_IF(P->flow_direction, qq) ? qq + 1 : qq) * N;
                    int e;
                    if (!t.valid) continue;
                    for (e = 0; e < 4; ++e) {
                        const float a1 = g1[(size_t)qq * N + j];
#pragma omp atomic
                        wt1[ow + t.idx[e]] += t.w[e] * a1;
                        if (model == 0) {
                            const float a2 = kap * g2[(size_t)qq * N + j] - a1;
#pragma omp atomic
                            wt2[ow + t.idx[e]] += t.w[e] * a2;
                        }
                    }
                }
            }
            /* primal: x+ = prox_G(x - T K'y), xbar = 2x+ - x */
#pragma omp parallel for reduction(+ : a_dx, a_x, e_box, e_inf, e_kty2, e_rp2) schedule(static)
            for (jj = 0; jj < (long long)M; ++jj) {
                const int t = (int)(jj / N), j = (int)(jj % N), r = j % Ny, c = j / Ny;
                const size_t o = (size_t)t * N;
                float Wt1 = 0.0f, Wt2 = 0.0f, ATp = 0.0f, div1 = 0.0f, div2 = 0.0f, cgrad, cw, tau_u, KTu, un, uo;
                int e, qq;
                for (qq = t - 1; qq <= t; ++qq) {
                    int isIF;
                    if (qq < 0 || qq >= npairs) continue;
                    isIF = pair_is_IF(P->flow_direction, qq);
                    if ((qq == t - 1) == isIF) {            /* warp of pair qq samples frame t */
                        const csr *m;
                        if (mf) continue;                   /* already in wt1 / wt2 */
                        m = &WT[qq];
                        const float *a1 = g1 + (size_t)qq * N, *a2 = g2 + (size_t)qq * N;
                        for (e = m->ptr[j]; e < m->ptr[j + 1]; ++e) {
                            Wt1 += m->val[e] * a1[m->idx[e]];
                            if (model == 0) Wt2 += m->val[e] * (kap * a2[m->idx[e]] - a1[m->idx[e]]);
                        }
                    } else if (get_taps(tp, v, occ, npairs, Ny, Nx, qq, j).valid) { /* -Id block of pair qq sits on frame t */
                        Wt1 -= g1[(size_t)qq * N + j];
                        if (model == 0) Wt2 -= kap * g2[(size_t)qq * N + j] - g1[(size_t)qq * N + j];
                    }
                }
                if (mf) {
                    Wt1 += wt1[o + j];
                    Wt2 += wt2[o + j];
                    ATp = apply_AT_mf(&ck, f, ny, nx, Ny, Nx, p + (size_t)t * n, r, c, 0);
                } else {
                    for (e = AT.ptr[j]; e < AT.ptr[j + 1]; ++e) ATp += AT.val[e] * p[(size_t)t * n + AT.idx[e]];
                }
                {
                    const float *gx = g1 + M + o, *gy = g1 + 2 * M + o;
                    if (c > 0) div1 += gx[j - Ny];
                    if (c < Nx - 1) div1 -= gx[j];
                    if (r > 0) div1 += gy[j - 1];
                    if (r < Ny - 1) div1 -= gy[j];
                }
                cgrad = (float)((c < Nx - 1) + (c > 0) + (r < Ny - 1) + (r > 0));
                cw = cW[o + j] * inv_h;
                KTu = ATp + inv_h * Wt1 + kap * div1;
                tau_u = 1.0f / (cA[j] + cw + kap * cgrad);
                uo = u[o + j];
                if (eval) {
                    const float ru = uo <= 0.0f ? (KTu < 0.0f ? KTu : 0.0f) : (uo >= 1.0f ? (KTu > 0.0f ? KTu : 0.0f) : KTu);
                    e_box += KTu < 0.0f ? -(double)KTu : 0.0;
                    e_kty2 += (double)KTu * KTu;
                    e_rp2 += (double)ru * ru;
                    if (model == 0) {
                        const float *gx = g2 + M + o, *gy = g2 + 2 * M + o;
                        float KTw;
                        if (c > 0) div2 += gx[j - Ny];
                        if (c < Nx - 1) div2 -= gx[j];
                        if (r > 0) div2 += gy[j - 1];
                        if (r < Ny - 1) div2 -= gy[j];
                        KTw = inv_h * Wt2 - kap * div1 + div2;
                        e_inf += fabs((double)KTw);
                        e_kty2 += (double)KTw * KTw;
                        e_rp2 += (double)KTw * KTw;
                    }
                    continue;
                }
                un = uo - tau_u * KTu;
                un = un < 0.0f ? 0.0f : (un > 1.0f ? 1.0f : un);          /* G = ind[0,1], :495 */
                u[o + j] = un;
                ub[o + j] = 2.0f * un - uo;
                if (last) { a_dx += (double)(un - uo) * (un - uo); a_x += (double)un * un; }
                if (model == 0) {
                    const float *gx = g2 + M + o, *gy = g2 + 2 * M + o;
                    float KTw, tau_w, wo, wn;
                    if (c > 0) div2 += gx[j - Ny];
                    if (c < Nx - 1) div2 -= gx[j];
                    if (r > 0) div2 += gy[j - 1];
                    if (r < Ny - 1) div2 -= gy[j];
                    KTw = inv_h * Wt2 - kap * div1 + div2;
                    tau_w = 1.0f / ((1.0f + kap) * (cw + cgrad));
                    wo = w[o + j];
                    wn = wo - tau_w * KTw;
                    w[o + j] = wn;
                    wb[o + j] = 2.0f * wn - wo;
                    if (last) { a_dx += (double)(wn - wo) * (wn - wo); a_x += (double)wn * wn; }
                }
            }
            /* dual, data term: p <- clamp(p + s (A xbar - f), -1, 1), :496 */
#pragma omp parallel for reduction(+ : a_dy, a_y, e_data, e_kx2, e_rd2, e_lin) schedule(static)
            for (jj = 0; jj < (long long)n * nc; ++jj) {
                const int t = (int)(jj / n), i = (int)(jj % n);
                const float *x = xsrc + (size_t)t * N;
                float acc = 0.0f, po = p[jj], pn;
                int e;
                if (mf) acc = apply_A_mf(&ck, f, Ny, Nx, x, i % ny, i / ny);
                else for (e = A.ptr[i]; e < A.ptr[i + 1]; ++e) acc += A.val[e] * x[A.idx[e]];
                if (eval) {
                    e_data += fabs((double)(acc - f_lr[jj]));
                    e_kx2 += (double)acc * acc;
                    e_rd2 += box_res2(acc - f_lr[jj], po, 1.0f);
                    e_lin += (double)f_lr[jj] * po;
                    continue;
                }
                pn = po + sigp[i] * (acc - f_lr[jj]);
                pn = pn < -1.0f ? -1.0f : (pn > 1.0f ? 1.0f : pn);
                p[jj] = pn;
                if (last) { a_dy += (double)(pn - po) * (pn - po); a_y += (double)pn * pn; }
            }
            /* dual, regulariser: g <- proj(g + S K xbar), :497-498, :513, :532-533 */
#pragma omp parallel for reduction(+ : a_dy, a_y, e_reg, e_kx2, e_rd2) schedule(static)
            for (jj = 0; jj < (long long)M; ++jj) {
                const int t = (int)(jj / N), j = (int)(jj % N), r = j % Ny, c = j / Ny;
                const size_t o = (size_t)t * N;
                const int hasx = c < Nx - 1, hasy = r < Ny - 1;
                const float *x = xsrc + o, *z = zsrc + o;
                const float uc = x[j], ux = hasx ? x[j + Ny] : uc, uy = hasy ? x[j + 1] : uc;
                const float wc = z[j], wx = hasx ? z[j + Ny] : wc, wy = hasy ? z[j + 1] : wc;
                float tu = 0.0f, tw = 0.0f, k1t, k1x, k1y, k2t = 0, k2x = 0, k2y = 0, rs1, rs2 = 0.0f, sg1, sg2;
                float n1t, n1x, n1y, o1t = g1[o + j], o1x = g1[M + o + j], o1y = g1[2 * M + o + j], nrm;
                int valid = 0;
                if (t < npairs) {
                    const taps td_ = get_taps(tp, v, occ, npairs, Ny, Nx, t, j);
                    const taps *tt = &td_;
                    valid = tt->valid;
                    if (valid) {
                        const int isIF = pair_is_IF(P->flow_direction, t);
                        const size_t ow = (size_t)(isIF ? t + 1 : t) * N, oi = (size_t)(isIF ? t : t + 1) * N;
                        tu = tt->w[0] * xsrc[ow + tt->idx[0]] + tt->w[1] * xsrc[ow + tt->idx[1]] + tt->w[2] * xsrc[ow + tt->idx[2]] +
                             tt->w[3] * xsrc[ow + tt->idx[3]] - xsrc[oi + j];
                        if (model == 0)
                            tw = tt->w[0] * zsrc[ow + tt->idx[0]] + tt->w[1] * zsrc[ow + tt->idx[1]] +
                                 tt->w[2] * zsrc[ow + tt->idx[2]] + tt->w[3] * zsrc[ow + tt->idx[3]] - zsrc[oi + j];
                    }
                }
                if (model == 0) {
                    const float dc = uc - wc;
                    k1t = (tu - tw) * inv_h;
                    k1x = hasx ? kap * ((ux - wx) - dc) : 0.0f;
                    k1y = hasy ? kap * ((uy - wy) - dc) : 0.0f;
                    k2t = kap * tw * inv_h;
                    k2x = hasx ? wx - wc : 0.0f;
                    k2y = hasy ? wy - wc : 0.0f;
                    rs1 = fmaxf(valid ? 4.0f * inv_h : 0.0f, fmaxf(hasx ? 4.0f * kap : 0.0f, hasy ? 4.0f * kap : 0.0f));
                    rs2 = fmaxf(valid ? 2.0f * kap * inv_h : 0.0f, fmaxf(hasx ? 2.0f : 0.0f, hasy ? 2.0f : 0.0f));
                } else {
                    k1t = tu * inv_h;
                    k1x = hasx ? ux - uc : 0.0f;
                    k1y = hasy ? uy - uc : 0.0f;
                    if (model == 1) {
                        rs1 = fmaxf(valid ? 2.0f * inv_h : 0.0f, fmaxf(hasx ? 2.0f : 0.0f, hasy ? 2.0f : 0.0f));
                    } else {
                        rs1 = valid ? 2.0f * inv_h : 0.0f;
                        rs2 = fmaxf(hasx ? 2.0f : 0.0f, hasy ? 2.0f : 0.0f);
                    }
                }
                sg1 = rs1 > 0.0f ? 1.0f / rs1 : 1.0f;
                sg2 = rs2 > 0.0f ? 1.0f / rs2 : 1.0f;
                if (eval) {
                    e_kx2 += (double)k1t * k1t + (double)k1x * k1x + (double)k1y * k1y + (double)k2t * k2t + (double)k2x * k2x + (double)k2y * k2y;
                    if (model == 2) {
                        e_reg += rad1 * fabs((double)k1t) + rad2 * sqrt((double)k1x * k1x + (double)k1y * k1y);
                        e_rd2 += box_res2(k1t, o1t, rad1) + ball_res2(0.0f, k1x, k1y, 0.0f, o1x, o1y, rad2);
                    } else {
                        e_reg += rad1 * sqrt((double)k1t * k1t + (double)k1x * k1x + (double)k1y * k1y);
                        e_rd2 += ball_res2(k1t, k1x, k1y, o1t, o1x, o1y, rad1);
                        if (model == 0) {
                            e_reg += rad2 * sqrt((double)k2t * k2t + (double)k2x * k2x + (double)k2y * k2y);
                            e_rd2 += ball_res2(k2t, k2x, k2y, g2[o + j], g2[M + o + j], g2[2 * M + o + j], rad2);
                        }
                    }
                    continue;
                }
                if (model == 2) {
                    n1t = valid ? o1t + sg1 * k1t : 0.0f;
                    n1t = n1t < -rad1 ? -rad1 : (n1t > rad1 ? rad1 : n1t);
                    n1x = o1x + sg2 * k1x;
                    n1y = o1y + sg2 * k1y;
                    nrm = sqrtf(n1x * n1x + n1y * n1y);
                    if (nrm > rad2) { n1x *= rad2 / nrm; n1y *= rad2 / nrm; }
                } else {
                    n1t = valid ? o1t + sg1 * k1t : 0.0f;
                    n1x = o1x + sg1 * k1x;
                    n1y = o1y + sg1 * k1y;
                    nrm = sqrtf(n1t * n1t + n1x * n1x + n1y * n1y);
                    if (nrm > rad1) { const float sc = rad1 / nrm; n1t *= sc; n1x *= sc; n1y *= sc; }
                }
                g1[o + j] = n1t; g1[M + o + j] = n1x; g1[2 * M + o + j] = n1y;
                if (last) {
                    a_dy += (double)(n1t - o1t) * (n1t - o1t) + (double)(n1x - o1x) * (n1x - o1x) + (double)(n1y - o1y) * (n1y - o1y);
                    a_y += (double)n1t * n1t + (double)n1x * n1x + (double)n1y * n1y;
                }
                if (model == 0) {
                    const float o2t = g2[o + j], o2x = g2[M + o + j], o2y = g2[2 * M + o + j];
                    float n2t = valid ? o2t + sg2 * k2t : 0.0f, n2x = o2x + sg2 * k2x, n2y = o2y + sg2 * k2y;
                    nrm = sqrtf(n2t * n2t + n2x * n2x + n2y * n2y);
                    if (nrm > rad2) { const float sc = rad2 / nrm; n2t *= sc; n2x *= sc; n2y *= sc; }
                    g2[o + j] = n2t; g2[M + o + j] = n2x; g2[2 * M + o + j] = n2y;
                    if (last) {
                        a_dy += (double)(n2t - o2t) * (n2t - o2t) + (double)(n2x - o2x) * (n2x - o2x) + (double)(n2y - o2y) * (n2y - o2y);
                        a_y += (double)n2t * n2t + (double)n2x * n2x + (double)n2y * n2y;
                    }
                }
            }
            if (last) { s_dx = a_dx; s_x = a_x; s_dy = a_dy; s_y = a_y; }
            if (eval) {
                d_reg = e_reg; d_data = e_data; d_kx2 = e_kx2; d_rd2 = e_rd2; d_box = e_box; d_inf = e_inf;
                d_kty2 = e_kty2; d_rp2 = e_rp2; d_lin = e_lin;
            }
        }
    }
    if (out) {
        out[0] = s_dx; out[1] = s_x; out[2] = s_dy; out[3] = s_y; out[4] = h; out[5] = t_loop;
#ifdef _OPENMP
        out[6] = omp_get_max_threads();
#else
        out[6] = 1;
#endif
        out[7] = 0;
        out[8] = d_reg; out[9] = d_data; out[10] = d_kx2; out[11] = d_rd2; out[12] = d_box; out[13] = d_inf;
        out[14] = d_kty2; out[15] = d_rp2; out[16] = d_lin;
    }
    if (WT) for (q = 0; q < npairs; ++q) csr_free(&WT[q]);
    free(WT); free(tp); free(cW); free(cA); free(sigp); free(ub); free(wb); free(p); free(g1); free(g2);
    free(wt1); free(wt2); free(ck.ck);
    if (!mf) { csr_free(&A); csr_free(&AT); }
    return 0;
}
