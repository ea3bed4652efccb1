// gfx950 kernels of the Multiframe Motion Coupling PDHG solve.
//
// The saddle-point problem is the one MultiframeMotionCoupling.m:346-574
// (init_u_prost) assembles from sparse matrices; here every operator is
// matrix-free:
//   W   motion coupling   helpers/constructWarpFF.m, constructWarpFB.m
//   A   blur + sampling   helpers/samplingOperator.m, helpers/RepConvMtx.m
//   G   forward differences (spmat_gradient2d)
// One PDHG iteration is two sweeps over HBM:
//   primal sweep  x+ = prox_G(x - T K'y),  xbar = 2x+ - x
//   dual sweep    y+ = prox_F*(y + S K xbar)
// The adjoint of the bilinear warp is a scatter.  It is evaluated per target
// tile: a workgroup walks the box of source pixels whose taps can land in its
// tile (precomputed from the flow) and accumulates them in LDS, so no global
// atomics are issued in the iteration.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <string.h>
#include <mutex>
#include "../../include/mmc_hip.h"
#include "mmc_internal.h"

#include "mmc_device.h"

// The pieces of the dual objective and of the primal optimality residual that one entry of K'y
// contributes (G = indicator of [0,1] on u, MultiframeMotionCoupling.m:495; w is free):
//   d[0] += max(0, -KTu)            G*(-K'y) on u
//   d[1] += |KTw|                   the dual objective is finite only where this vanishes
//   d[2] += KTu^2 + KTw^2           |K'y|^2
//   d[3] += dist(-K'y, dG(x))^2     at u = 0 only KTu < 0 violates, at u = 1 only KTu > 0; all of KTw
__device__ __forceinline__ void dual_objective_terms(float (&d)[4], float KTu, float KTw, float u) {
    d[0] += fmaxf(-KTu, 0.f);
    d[1] += fabsf(KTw);
    d[2] += KTu * KTu + KTw * KTw;
    const float ru = u <= 0.f ? fminf(KTu, 0.f) : (u >= 1.f ? fmaxf(KTu, 0.f) : KTu);
    d[3] += ru * ru + KTw * KTw;
}

// --------------------------------------------------------------------------
// primal sweep
// --------------------------------------------------------------------------
// MODE_UPDATE   : u,w <- prox_G(x - T K'y); ub,wb <- 2x+ - x
// MODE_RAW      : ub,wb <- K'y                      (test hook)
// MODE_STEPS    : ub,wb <- tau_u, tau_w from the cW plane; MODE_STEPS_UPD: as the update derives them (test hooks)
// MODE_DUALOBJ  : the K'y terms of the dual objective and the primal optimality residual; writes nothing
// Each thread owns a row quad: four consecutive rows of one column, so that every
// per-pixel array moves as one 16-byte access per lane (the last quad of a column is
// short when Ny is not a multiple of 4).
// The low-res dual p of the tile and a zero-padded copy of the composite
// kernel sit in LDS, which makes the interior A'p a branch-free Jr x Jc sum:
// tap (jr,jc) of pixel row x = r - oy uses low-res row R = x/f - jr with
// kernel row (x mod f) + f*jr; rows of the table beyond CH and low-res pixels
// outside the frame hold zeros.
// 16-byte accesses to the planes.  A column starts at a multiple of Ny floats, which is 16-byte aligned
// only when Ny is a multiple of 4: the type says 4-byte alignment (global memory takes dwordx4
// accesses at any dword address; LDS accesses elsewhere in this file stay aligned).
typedef float rowquad_t __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ void ld4(float (&d)[4], const float *p) {
    const rowquad_t v = *reinterpret_cast<const rowquad_t *>(p);
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
}
__device__ __forceinline__ void st4(float *p, const float (&s)[4]) {
    rowquad_t v = {s[0], s[1], s[2], s[3]};
    *reinterpret_cast<rowquad_t *>(p) = v;
}
// The last quad of a column holds nv < 4 rows when Ny is not a multiple of 4; what follows it in memory
// is the top of the next column.
__device__ __forceinline__ void st4m(float *p, const float (&s)[4], int nv) {
    if (nv >= 4) { st4(p, s); return; }
#pragma unroll
    for (int i = 0; i < 3; ++i) if (i < nv) p[i] = s[i];
}

// Tuning switches of the row-quad sweeps (defaults are the measured best).
#ifndef MMC_TAP_ODD_PITCH
#define MMC_TAP_ODD_PITCH 1     // odd LDS pitch of the dual sweep's tap windows (conflict-free reads)
#endif
#ifndef MMC_DATA_QUAD
#define MMC_DATA_QUAD 1         // data term of the dual sweep summed by groups of four lanes, no second barrier
#endif
#ifndef MMC_SKIP_NOPAIR
#define MMC_SKIP_NOPAIR 0       // 1: frames without a pair skip their temporal dual planes (measured slower: a wave per SIMD less)
#endif
#ifndef MMC_DUAL_UX_LDS
#define MMC_DUAL_UX_LDS 1       // 1: right neighbour column and the row below of xbar_u come from the staged data-term window
#endif
#ifndef MMC_OCC_PRIMAL
#define MMC_OCC_PRIMAL 1
#endif
#ifndef MMC_PRIMAL_EARLY_X
#define MMC_PRIMAL_EARLY_X 0    // 1: the primal sweep loads u, w and the step-size inputs ahead of the scatter (one round trip less per workgroup)
#endif
#ifndef MMC_SCATTER_SCU
#define MMC_SCATTER_SCU 5       // source pixels per thread and pass of the primal sweep's scatter
#endif
#ifndef MMC_ATP_QUAD
#define MMC_ATP_QUAD 1          // zoom 4, kernels beyond 3 x 3 low-res taps: A'p of a row quad with 16-byte weight reads
#endif
#ifndef MMC_DUAL_UC_LDS
#define MMC_DUAL_UC_LDS 1         // 1: the quad's own xbar_u comes from the staged data-term window as well
#endif
#ifndef MMC_OCC_DUAL
#define MMC_OCC_DUAL 1
#endif
template <int MODEL, int MODE, bool REDUCE, int F, bool SEPK = false>
__global__ __launch_bounds__(MMC_THREADS, MMC_OCC_PRIMAL) void k_primal4(Geom g, Arrays a, int t0, int nf) {
    extern __shared__ __align__(16) float lds[];
    // In the update sweep the tap weights are scattered along with the duals, which yields the
    // scattered part of cW without reading the cW plane (4 bytes per pixel less).
    constexpr bool CWS = MMC_CW_SCATTER && (MODE == MODE_UPDATE || MODE == MODE_STEPS_UPD);
    fx64 *acc1 = reinterpret_cast<fx64 *>(lds);
    fx64 *acc2 = acc1 + TILE_R * TILE_C;
    fx64 *acc3 = acc2 + TILE_R * TILE_C;
    float *ckp = lds + 6 * TILE_R * TILE_C;
    float *pl = ckp + ((g.MP * g.NP + 3) & ~3);
    // Separable composite kernel (Geom::sep): column sums of the low-res window per tile column, and the two factors
    constexpr bool SEP = SEPK && MODE != MODE_STEPS && MODE != MODE_STEPS_UPD;    // the launcher picks SEPK from Geom::sep
    float *tl = pl + ((g.PR * g.PC + 3) & ~3);          // PR x TILE_C
    float *kys = tl + ((g.PR * TILE_C + 3) & ~3);       // ky, MP floats (16-byte aligned)
    float *kxs = kys + g.MP;                            // kx, NP floats
    const int f = F > 0 ? F : g.f;
    const int tid = threadIdx.x;
    int tile, trel;
    if (!decode_block(g, nf, tile, trel)) return;
    const int t = t0 + trel;
    const int tc = fdiv(tile, g.dv_tiles_r), tr = tile - tc * g.tiles_r;
    const int r0 = tr * TILE_R, c0 = tc * TILE_C;
    const float kap = (MODEL == MODEL_IC) ? g.kappa : 1.f;
    const float *p = a.p + (size_t)t * g.n;
    // First low-res row / column of the staged window of p.  In the first tile row (column) the
    // virtual rows above (left of) the frame start at kernel coordinate 0.
    const int Rbase = (r0 == 0 ? 0 : (r0 - g.oy) / f) - (g.Jr - 1);
    const int Cbase = (c0 == 0 ? 0 : (c0 - g.ox) / f) - (g.Jc - 1);

    // This thread's row quad.  Its own loads are issued first: they do not
    // depend on the scatter below and stay in flight across the barriers.
    // With zoom 4 the columns of one wave are taken 4 apart, so that the column phase of the
    // composite kernel (c mod 4) is the same in all 64 lanes and its weights can be scalar loads.
    constexpr int QPC = TILE_R / 4;                                   // quads per tile column
    constexpr int CPW = 64 / QPC;                                     // tile columns per wave
    constexpr bool PHASED = F == 4 && QPC <= 64 && CPW >= 1 && TILE_C == 4 * CPW && MMC_THREADS == 256;
    const int rq = tid & (QPC - 1);
    const int lc = PHASED ? ((tid / QPC) % CPW) * 4 + (tid >> 6) : tid / QPC;
    const int r = r0 + 4 * rq, c = c0 + lc;
    const bool active = r < g.Ny && c < g.Nx;
    const int j = active ? r + g.Ny * c : 0;
    const int nv = F == 4 ? 4 : min(g.Ny - r, 4);                       // rows of this quad inside the frame (zoom 4: Ny = 4 ny)
    const bool cl = c > 0, cr = c < g.Nx - 1;
    const bool idA = pair_exists(g, t) && pair_is_IF(g, t);            // -Id of pair t sits on this frame
    const bool idB = pair_exists(g, t - 1) && !pair_is_IF(g, t - 1);   // -Id of pair t-1
    float uo[4], wo[4], cWv[4], cAv[4];
    float gx1[4], gl1[4] = {0.f, 0.f, 0.f, 0.f}, gy1[4], up1 = 0.f;
    float gx2[4], gl2[4] = {0.f, 0.f, 0.f, 0.f}, gy2[4], up2 = 0.f;
    float sA1[4] = {0.f, 0.f, 0.f, 0.f}, sA2[4] = {0.f, 0.f, 0.f, 0.f};
    unsigned idb = 0;             // identity bits of this quad (CWS)
    // Loads of this thread's own quad, in groups.  The dual planes (grp1, grp2, grpA) are issued
    // here, ahead of the scatter, and are folded into eu / ew right after the staging barrier; the
    // state and step-size planes (grpX) are issued after the scatter, where they are first needed
    // (measured: fewer live registers across the scatter beats hiding their latency).
    auto own_loads = [&](bool grpX, bool grpA, bool grp2, bool grp1) {
        if (!active) return;
        if (grpX && MODE == MODE_DUALOBJ) ld4(uo, FR(a.u, t) + j);      // no step sizes in this mode
        if (grpX && MODE != MODE_DUALOBJ) {
            if (!CWS) ld4(cWv, FR(a.cW, t) + j);
            else idb = a.idq[(size_t)t * MMC_IDQ_PER_FRAME(g) + (size_t)(r >> 2) + (size_t)((g.Ny + 3) >> 2) * c];
            if (F == 4 && g.cA_margin > 0 && r >= g.cA_margin && r + 4 <= g.Ny - g.cA_margin &&
                c >= g.cA_margin && c < g.Nx - g.cA_margin && !ABL(128)) {
                // interior: period-4 pattern (row phase of quad row i is i)
                const int cph = PHASED ? __builtin_amdgcn_readfirstlane(c & 3) : (c & 3);
#pragma unroll
                for (int i = 0; i < 4; ++i) cAv[i] = g.cA_tab[i + 4 * cph];
            } else {
                ld4(cAv, a.cA + j);
            }
            if (MODE == MODE_UPDATE) {
                ld4(uo, FR(a.u, t) + j);
                if (MODEL == MODEL_IC) ld4(wo, FR(a.w, t) + j);
            }
        }
        if (MODE == MODE_STEPS) return;
        if (grp1) {
            const float *g1x = FR(a.g1x, t) + j, *g1y = FR(a.g1y, t) + j;
            ld4(gx1, g1x);
            if (cl) ld4(gl1, g1x - g.Ny);
            ld4(gy1, g1y);
            if (r > 0) up1 = g1y[-1];
        }
        if (grp2 && MODEL == MODEL_IC) {
            const float *g2x = FR(a.g2x, t) + j, *g2y = FR(a.g2y, t) + j;
            ld4(gx2, g2x);
            if (cl) ld4(gl2, g2x - g.Ny);
            ld4(gy2, g2y);
            if (r > 0) up2 = g2y[-1];
        }
        if (grpA && idA) {
            ld4(sA1, FR(a.g1t, t) + j);
            if (MODEL == MODEL_IC) ld4(sA2, FR(a.g2t, t) + j);
        }
    };
    own_loads(MMC_PRIMAL_EARLY_X != 0, true, true, true);

    // source boxes of the scatter below, fetched ahead of the barrier
    int4 sbox[2] = {make_int4(0, -1, 0, -1), make_int4(0, -1, 0, -1)};
    int2 sovf[2] = {make_int2(0, 0), make_int2(0, 0)};          // [begin, end) of the tile's overflow list
    if (MODE != MODE_STEPS) {
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            const int q = which == 0 ? t - 1 : t;
            if (pair_exists(g, q) && (which == 0) == pair_is_IF(g, q)) {
                const size_t slot = (size_t)(q + 1) * g.ntiles + tile;
                sbox[which] = *reinterpret_cast<const int4 *>(a.bbox + slot * 4);
                sovf[which] = make_int2(a.ovf_ptr[slot], a.ovf_ptr[slot + 1]);
            }
        }
    }

    for (int k = tid; k < TILE_R * TILE_C; k += MMC_THREADS) {
        acc1[k] = 0;
        if (MODEL == MODEL_IC) acc2[k] = 0;
        if (CWS) acc3[k] = 0;
    }
    if (MODE != MODE_STEPS) {
        for (int k = tid; k < g.MP * g.NP; k += MMC_THREADS) ckp[k] = a.ckp[k];
        if (SEP) {
            if (tid < g.MP) kys[tid] = a.ksep[tid];
            if (tid >= 64 && tid - 64 < g.NP) kxs[tid - 64] = a.ksep[MMC_KSEP_PITCH + tid - 64];
        }
        for (int k = tid; k < g.PR * g.PC; k += MMC_THREADS) {
            const int pc = fdiv(k, g.dv_PR), pr = k - pc * g.PR;
            const int R = Rbase + pr, C = Cbase + pc;
            pl[k] = (R >= 0 && R < g.ny && C >= 0 && C < g.nx) ? p[R + g.ny * C] : 0.f;
        }
    }
    __syncthreads();

    // The early loads have landed (the staging loads behind them have).  Fold them into the part
    // of K'y that does not depend on the scatter, so that 8 instead of 34 registers stay live
    // across it:  eu = kappa G'g1 - (1/h) Id'g1t,  ew = -kappa G'g1 + G'g2 - (1/h) Id'(kappa g2t - g1t).
    float eu[4] = {0.f, 0.f, 0.f, 0.f}, ew[4] = {0.f, 0.f, 0.f, 0.f};
    if (MODE != MODE_STEPS && active) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool below = r + i < g.Ny - 1;
            const float d1 = gl1[i] - (cr ? gx1[i] : 0.f) + (i > 0 ? gy1[i > 0 ? i - 1 : 0] : up1) - (below ? gy1[i] : 0.f);
            eu[i] = kap * d1;
            if (MODEL == MODEL_IC) {
                const float d2 = gl2[i] - (cr ? gx2[i] : 0.f) + (i > 0 ? gy2[i > 0 ? i - 1 : 0] : up2) - (below ? gy2[i] : 0.f);
                ew[i] = d2 - kap * d1;
            }
            if (idA) {
                eu[i] -= g.inv_h * sA1[i];
                if (MODEL == MODEL_IC) ew[i] -= g.inv_h * (kap * sA2[i] + g.wsgn * sA1[i]);
            }
        }
    }

    // --- scatter part of W': walk the source box of this tile ----------------
    // SCU source pixels per thread and pass, all their loads issued before the
    // first use so that one pass costs one memory round trip.
    if (MODE != MODE_STEPS) {
        constexpr int SCU = MMC_SCATTER_SCU;
        for (int which = 0; which < 2; ++which) {
            const int q = which == 0 ? t - 1 : t;
            if (!pair_exists(g, q)) continue;
            const bool isIF = pair_is_IF(g, q);
            if ((which == 0) != isIF) continue;
            const int rmin = sbox[which].x, rmax = sbox[which].y, cmin = sbox[which].z, cmax = sbox[which].w;
            if (ABL(1)) continue;
            const int nr = rmax - rmin + 1, total = (rmax < rmin || cmax < cmin) ? 0 : nr * (cmax - cmin + 1);
            const int ob = sovf[which].x, novf = sovf[which].y - ob;
            const float inv_nr = fast_rcp((float)nr);
            const float *vx = FR(a.vx, q), *vy = FR(a.vy, q);
            const float *s1p = FR(a.g1t, q), *s2p = FR(a.g2t, q);
            auto scatter_one = [&](int srow, int scol, float fx, float fy, float f1, float f2) {
                Taps tp = make_taps(g, srow, scol, fx, fy);
                if (!tp.valid) return;
                const float s1 = f1;
                const float s2 = (MODEL == MODEL_IC) ? kap * f2 + g.wsgn * s1 : 0.f;
                const double d1 = fx_scaled(s1), d2 = fx_scaled(s2);
                const int a0 = tp.r0 - r0, a1 = tp.r1 - r0, b0 = tp.c0 - c0, b1 = tp.c1 - c0;
                const bool ra0 = (unsigned)a0 < TILE_R, ra1 = (unsigned)a1 < TILE_R;
                const bool cb0 = (unsigned)b0 < TILE_C, cb1 = (unsigned)b1 < TILE_C;
                if (ABL(2)) { if (s1 == 12345.f && ra0 && cb1) acc1[0] = to_fx(1.0, d2); return; }
                if (ra0 && cb0) { fx_add(&acc1[a0 + TILE_R * b0], tp.w00, d1); if (MODEL == MODEL_IC) fx_add(&acc2[a0 + TILE_R * b0], tp.w00, d2); if (CWS) fx_add(&acc3[a0 + TILE_R * b0], tp.w00, 1099511627776.0); }
                if (ra1 && cb0) { fx_add(&acc1[a1 + TILE_R * b0], tp.w10, d1); if (MODEL == MODEL_IC) fx_add(&acc2[a1 + TILE_R * b0], tp.w10, d2); if (CWS) fx_add(&acc3[a1 + TILE_R * b0], tp.w10, 1099511627776.0); }
                if (ra0 && cb1) { fx_add(&acc1[a0 + TILE_R * b1], tp.w01, d1); if (MODEL == MODEL_IC) fx_add(&acc2[a0 + TILE_R * b1], tp.w01, d2); if (CWS) fx_add(&acc3[a0 + TILE_R * b1], tp.w01, 1099511627776.0); }
                if (ra1 && cb1) { fx_add(&acc1[a1 + TILE_R * b1], tp.w11, d1); if (MODEL == MODEL_IC) fx_add(&acc2[a1 + TILE_R * b1], tp.w11, d2); if (CWS) fx_add(&acc3[a1 + TILE_R * b1], tp.w11, 1099511627776.0); }
            };
            for (int base = tid; base < total; base += MMC_THREADS * SCU) {
                int sr[SCU], sc[SCU];
                float fvx[SCU], fvy[SCU], fs1[SCU], fs2[SCU];
#pragma unroll
                for (int k = 0; k < SCU; ++k) {
                    const int idx = min(base + k * MMC_THREADS, total - 1);
                    int cc = (int)(((float)idx + 0.5f) * inv_nr);
                    int rr = idx - cc * nr;
                    if (rr < 0) { --cc; rr += nr; }
                    if (rr >= nr) { ++cc; rr -= nr; }
                    sr[k] = rmin + rr;
                    sc[k] = cmin + cc;
                    const int js = sr[k] + g.Ny * sc[k];
                    fvx[k] = vx[js];
                    fvy[k] = vy[js];
                    fs1[k] = s1p[js];
                    fs2[k] = (MODEL == MODEL_IC) ? s2p[js] : 0.f;
                }
#pragma unroll
                for (int k = 0; k < SCU; ++k) {
                    if (base + k * MMC_THREADS >= total) continue;
                    scatter_one(sr[k], sc[k], fvx[k], fvy[k], fs1[k], fs2[k]);
                }
            }
            // sources outside the rectangle (stray vectors): one gather each
            for (int k = tid; k < novf; k += MMC_THREADS) {
                const int js = a.ovf_idx[ob + k];
                const int scol = fdiv(js, g.dv_Ny), srow = js - scol * g.Ny;
                scatter_one(srow, scol, vx[js], vy[js], s1p[js], (MODEL == MODEL_IC) ? s2p[js] : 0.f);
            }
        }
        __syncthreads();
    }

    // Separable kernel, first factor: for every low-res row of the window and every column of the tile the sum
    // over the low-res columns, tl[row + PR * column] = sum_jc kx[phase + 4 jc] p[row, Chi - jc]; the first and
    // the last column of the frame add the virtual columns that the replicate border of the blur folds onto
    // them (helpers/RepConvMtx.m:23-32).  Done here, after the scatter, where few registers are live.
    // low-res index and kernel phase of a kernel coordinate x >= 0 (x = 4 q + phase at zoom 4)
    const float inv_f = fast_rcp((float)f);
    auto lr_of = [&](int x, int &q, int &phase) {
        if (F == 4) { q = x >> 2; phase = x & 3; }
        else { q = (int)(((float)x + 0.5f) * inv_f); phase = x - f * q; }
    };
    if (SEP) {
        const int cvlast = max(f * (g.nx - 1) + g.ox + g.CW - 1, g.Nx - 1);
#pragma unroll 1
        for (int k = tid; k < g.PR * TILE_C; k += MMC_THREADS) {
            const int cl = fdiv(k, g.dv_PR), pr = k - cl * g.PR;
            const int cc = c0 + cl;
            const int cvA = cc == 0 ? min(g.ox, 0) : cc, cvB = cc == g.Nx - 1 ? cvlast : (cc < g.Nx ? cc : cc - 1);
            float sacc = 0.f;
#pragma unroll 1
            for (int cv = cvA; cv <= cvB; ++cv) {
                int Chv, bcv;
                lr_of(cv - g.ox, Chv, bcv);
                const float *pp = pl + pr + g.PR * (Chv - Cbase);
                const float *kw = kxs + bcv;
#pragma unroll 1
                for (int jc = 0; jc < g.Jc; ++jc) sacc += kw[f * jc] * pp[-g.PR * jc];
            }
            tl[k] = sacc;
        }
        __syncthreads();
    }

    if (!MMC_PRIMAL_EARLY_X) own_loads(true, false, false, false);      // u, w, cW, cA: only needed from here on
    float part[2] = {0.f, 0.f};
    float wres[1] = {0.f};        // sum ((x+ - x) / tau)^2: the step-weighted change the step balancing compares
    float dob[4] = {0.f, 0.f, 0.f, 0.f};
    if (active) {
        float *u = FR(a.u, t) + j, *w = FR(a.w, t) + j, *ub = FR(a.ub, t) + j, *wb = FR(a.wb, t) + j;
        float tau_u[4], tau_w[4];
        const bool isfix = MODEL == MODEL_IC && g.fixed && g.frame_begin + t == 0;   // identity rows of q1, q2 (:560-561)
        const float fixc = isfix ? 1.f : 0.f;
        if (CWS) {
#pragma unroll
            for (int i = 0; i < 4; ++i) cWv[i] = from_fx_plus(acc3[4 * rq + i + TILE_R * lc], (int)((idb >> (2 * i)) & 3u));
        }
        auto compute_tau = [&]() {
            // Every operation rounded on its own (no fused multiply-add picked by the compiler): the step
            // sizes must not depend on which instantiation of this kernel computes them.
#pragma clang fp contract(off)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float cgrad = (float)((int)cl + (int)cr + (int)(r + i > 0) + (int)(r + i < g.Ny - 1));
                const float cw = cWv[i] * g.inv_h;
                if (MODEL == MODEL_IC) {
                    tau_u[i] = g.tau_s * fast_rcp(cAv[i] + cw + kap * cgrad + fixc);
                    tau_w[i] = g.tau_s * fast_rcp((1.f + kap) * (cw + cgrad) + fixc);
                } else {
                    tau_u[i] = g.tau_s * fast_rcp(cAv[i] + cw + cgrad);
                    tau_w[i] = 0.f;
                }
            }
        };
        if (MODE != MODE_DUALOBJ) compute_tau();
        if (MODE == MODE_STEPS || MODE == MODE_STEPS_UPD) {
            st4m(ub, tau_u, nv);
            if (MODEL == MODEL_IC) st4m(wb, tau_w, nv);
        } else {
            float Wt1[4], Wt2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                Wt1[i] = from_fx(acc1[4 * rq + i + TILE_R * lc]);
                if (MODEL == MODEL_IC) Wt2[i] = from_fx(acc2[4 * rq + i + TILE_R * lc]);
            }
            // identity part of W' of pair t-1 (that of pair t is already in eu, ew)
            if (idB) {
                float s1[4], s2[4];
                ld4(s1, FR(a.g1t, t - 1) + j);
                if (MODEL == MODEL_IC) ld4(s2, FR(a.g2t, t - 1) + j);
#pragma unroll
                for (int i = 0; i < 4; ++i) { Wt1[i] -= s1[i]; if (MODEL == MODEL_IC) Wt2[i] -= kap * s2[i] + g.wsgn * s1[i]; }
            }
            // A'p
            float ATp[4];
            const int xc = c - g.ox;
            const int Chi = xc / f, bc = xc - f * Chi;
            const bool cborder = !cl || !cr;
            const bool j33 = g.Jr == 3 && g.Jc == 3;
            // The reference's default geometry (zoom 4, 7x7 blur: oy = ox = -3): kernel row phase and
            // low-res row of quad row i are fixed (x = r+i+3: phases 3,0,1,2; rows a,a+1,a+1,a+1 with
            // a = r/4), so the 4 x 3 low-res values sit in registers and the weights are scalars.
            const bool fast77 = PHASED && j33 && g.oy == -3 && g.ox == -3 && !ABL(2048);
            const bool qborder = cborder || r == 0 || r + 4 >= g.Ny;      // the quad touches the frame border
            if (ABL(4)) {
#pragma unroll
                for (int i = 0; i < 4; ++i) ATp[i] = 0.f;
            } else if (fast77 && !qborder) {
                float pv[4][3];
                const float *plb = pl + rq + g.PR * (Chi - Cbase);          // low-res rows a-2 .. a+1
#pragma unroll
                for (int jc = 0; jc < 3; ++jc)
#pragma unroll
                    for (int k = 0; k < 4; ++k) pv[k][jc] = plb[k - g.PR * jc];
                const float *wk = a.ckp + 12 * __builtin_amdgcn_readfirstlane(bc);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int br = (i + 3) & 3, dR = (i + 3) >> 2;
                    float s = 0.f;
#pragma unroll
                    for (int jc = 0; jc < 3; ++jc)
#pragma unroll
                        for (int jr = 0; jr < 3; ++jr)
                            s += wk[br + 4 * jr + 48 * jc] * pv[dR - jr + 2][jc];
                    ATp[i] = s;
                }
            } else if (SEP && (F != 4 || r == 0 || r + 4 >= g.Ny)) {
                // second factor row by row: every quad at the other zoom factors, and at zoom 4 the quads with the
                // first or the last row of the frame, which adds the virtual rows folded onto it
                const int rvlast = max(f * (g.ny - 1) + g.oy + g.CH - 1, g.Ny - 1);
                const float *tcl = tl + g.PR * lc - Rbase;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ri = r + i;
                    const int rv0 = ri == 0 ? min(g.oy, 0) : ri, rv1 = ri == g.Ny - 1 ? rvlast : (i < nv ? ri : ri - 1);
                    float sacc = 0.f;
#pragma unroll 1
                    for (int rv = rv0; rv <= rv1; ++rv) {
                        int Rhv, brv;
                        lr_of(rv - g.oy, Rhv, brv);
                        const float *tc = tcl + Rhv;
                        const float *kw = kys + brv;
#pragma unroll 1
                        for (int jr = 0; jr < g.Jr; ++jr) sacc += kw[f * jr] * tc[-jr];
                    }
                    ATp[i] = sacc;
                }
            } else if (SEP) {
                // second factor of the separable kernel on the column sums: phases and low-res rows of the quad as
                // in the next branch
                const int x0 = r - g.oy, R0 = x0 >> 2, ph = x0 & 3;
                float tq[4] = {0.f, 0.f, 0.f, 0.f};
                const float4 *wq = reinterpret_cast<const float4 *>(kys);
                const float *tcol = tl + (R0 + 1 - Rbase) + g.PR * lc;
                float pa = tcol[0];
#pragma unroll 1
                for (int jr = 0; jr < g.Jr; ++jr) {
                    const float pb = tcol[-jr - 1];
                    const float4 wv = wq[jr];
                    tq[0] += wv.x * (ph > 0 ? pa : pb);
                    tq[1] += wv.y * (ph > 1 ? pa : pb);
                    tq[2] += wv.z * (ph > 2 ? pa : pb);
                    tq[3] += wv.w * pb;
                    pa = pb;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k = (ph + i) & 3;
                    ATp[i] = k == 0 ? tq[0] : k == 1 ? tq[1] : k == 2 ? tq[2] : tq[3];
                }
            } else if (F == 4 && MMC_ATP_QUAD && !qborder) {
                // Interior quads at zoom 4, any kernel size.  Quad row i has kernel row phase (ph + i) & 3 and
                // low-res row R0 + ((ph + i) >> 2), ph the same in the whole tile: the four phases of the kernel
                // rows 4 jr .. 4 jr + 3 are one 16-byte read of the table, and a column of the low-res window is
                // walked once for the four rows.  Sums are kept by phase; each adds its taps in the order of
                // the generic path below.
                const int x0 = r - g.oy, R0 = x0 >> 2, ph = x0 & 3;
                float tq[4] = {0.f, 0.f, 0.f, 0.f};
                const float *ckc = ckp + g.MP * bc;
                const float *plc = pl + (R0 + 1 - Rbase) + g.PR * (Chi - Cbase);
#pragma unroll 1
                for (int jc = 0; jc < g.Jc; ++jc) {
                    const float4 *wq = reinterpret_cast<const float4 *>(ckc + 4 * g.MP * jc);
                    const float *pp = plc - g.PR * jc;
                    float pa = pp[0];                           // low-res row R0 + 1 - jr: rows whose phase wrapped
#pragma unroll 1
                    for (int jr = 0; jr < g.Jr; ++jr) {
                        const float pb = pp[-jr - 1];           // low-res row R0 - jr
                        const float4 wv = wq[jr];
                        tq[0] += wv.x * (ph > 0 ? pa : pb);
                        tq[1] += wv.y * (ph > 1 ? pa : pb);
                        tq[2] += wv.z * (ph > 2 ? pa : pb);
                        tq[3] += wv.w * pb;
                        pa = pb;
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k = (ph + i) & 3;
                    ATp[i] = k == 0 ? tq[0] : k == 1 ? tq[1] : k == 2 ? tq[2] : tq[3];
                }
            } else {
                // Generic path.  The replicate border of the blur (helpers/RepConvMtx.m:23-32) folds
                // every virtual row above the first / below the last row onto that row (same for
                // columns), so a border pixel sums the interior formula over those virtual positions.
                const int rvlast = max(g.f * (g.ny - 1) + g.oy + g.CH - 1, g.Ny - 1);
                const int cvlast = max(g.f * (g.nx - 1) + g.ox + g.CW - 1, g.Nx - 1);
                const int cv0 = c == 0 ? min(g.ox, 0) : c, cv1 = c == g.Nx - 1 ? cvlast : c;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ri = r + i;
                    const int rv0 = ri == 0 ? min(g.oy, 0) : ri, rv1 = ri == g.Ny - 1 ? rvlast : ri;
                    float s = 0.f;
                    for (int cv = cv0; cv <= cv1; ++cv) {
                        const int xcv = cv - g.ox;
                        const int Chv = xcv / f, bcv = xcv - f * Chv;
                        for (int rv = rv0; rv <= rv1; ++rv) {
                            const int x = rv - g.oy;
                            const int Rhi = x / f, br = x - f * Rhi;
                            const float *ck0 = ckp + br + g.MP * bcv;
                            const float *pl0 = pl + (Rhi - Rbase) + g.PR * (Chv - Cbase);
                            if (F == 4 && j33) {
                                // 3 x 3 low-res taps (7x7 or 9x9 blur at zoom 4): MP = 12, fully unrolled
#pragma unroll
                                for (int jc = 0; jc < 3; ++jc)
#pragma unroll
                                    for (int jr = 0; jr < 3; ++jr)
                                        s += ck0[4 * jr + 48 * jc] * pl0[-jr - g.PR * jc];
                            } else {
                                for (int jc = 0; jc < g.Jc; ++jc)
                                    for (int jr = 0; jr < g.Jr; ++jr)
                                        s += ck0[f * jr + g.MP * f * jc] * pl0[-jr - g.PR * jc];
                            }
                        }
                    }
                    ATp[i] = s;
                }
            }
            float KTu[4], KTw[4];
            // rows of A that deviate from the composite kernel: near the frame border only
            if (a.ucorr && (r < g.uc_margin || r + 4 > g.Ny - g.uc_margin || c < g.uc_margin || c >= g.Nx - g.uc_margin)) {
                float uc[4];
                ld4(uc, a.ucorr + (size_t)t * g.N + j);
#pragma unroll
                for (int i = 0; i < 4; ++i) ATp[i] += uc[i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                KTu[i] = ATp[i] + g.inv_h * Wt1[i] + eu[i];
                KTw[i] = (MODEL == MODEL_IC) ? g.inv_h * Wt2[i] + ew[i] : 0.f;
            }
            if (isfix) {
                float q1v[4], q2v[4];
                ld4(q1v, a.q1 + j);
                ld4(q2v, a.q2 + j);
#pragma unroll
                for (int i = 0; i < 4; ++i) { KTu[i] += q1v[i]; KTw[i] += q2v[i]; }
            }
            if (MODE == MODE_RAW) {
                st4m(ub, KTu, nv);
                if (MODEL == MODEL_IC) st4m(wb, KTw, nv);
            } else if (MODE == MODE_DUALOBJ) {
#pragma unroll
                for (int i = 0; i < 4; ++i) if (i < nv) dual_objective_terms(dob, KTu[i], KTw[i], uo[i]);
            } else {
                float un[4], ubn[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    un[i] = fminf(fmaxf(uo[i] - tau_u[i] * KTu[i], 0.f), 1.f);   // G = ind[0,1], :495
                    ubn[i] = 2.f * un[i] - uo[i];
                    if (REDUCE && i < nv) {
                        part[0] += (un[i] - uo[i]) * (un[i] - uo[i]); part[1] += un[i] * un[i];
                        const float q = (un[i] - uo[i]) / tau_u[i];
                        wres[0] += q * q;
                    }
                }
                st4m(u, un, nv);
                st4m(ub, ubn, nv);
                if (MODEL == MODEL_IC) {
                    float wn[4], wbn[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        wn[i] = wo[i] - tau_w[i] * KTw[i];
                        wbn[i] = 2.f * wn[i] - wo[i];
                        if (REDUCE && i < nv) {
                            part[0] += (wn[i] - wo[i]) * (wn[i] - wo[i]); part[1] += wn[i] * wn[i];
                            const float q = (wn[i] - wo[i]) / tau_w[i];
                            wres[0] += q * q;
                        }
                    }
                    st4m(w, wn, nv);
                    st4m(wb, wbn, nv);
                }
            }
        }
    }
    if (MODE == MODE_DUALOBJ) block_accumulate<4>(dob, a.sums, SUM_DUALOBJ);
    else if (REDUCE) { block_accumulate<2>(part, a.sums, SUM_RES); __syncthreads(); block_accumulate<1>(wres, a.sums, SUM_WRES); }
}

// --------------------------------------------------------------------------
// dual sweep: regulariser duals (g1, g2) and data-term dual (p) in one launch
// --------------------------------------------------------------------------
// MODE_UPDATE: g <- proj(g + S K xbar), p <- clamp(p + s (A xbar_u - f), -1, 1)   (:496-498)
// MODE_RAW   : g, p <- K xbar                  (test hook)
// MODE_STEPS : g?t <- sigma of the group, p <- sigma of the data rows   (test hook)
// MODE_ENERGY: primal energy, |K x|^2 and the dual optimality residual at (ub, wb) as given; writes nothing
// Each thread owns a row quad (four consecutive rows of one column) for
// the regulariser duals.  The workgroup also stages xbar_u of its tile plus the
// blur halo in LDS (replicate border applied while staging) and updates the
// low-res duals p it owns from it: those whose first high-res pixel (f R, f C)
// lies in the tile.  Groups of adjacent lanes share a low-res pixel and split
// the composite-kernel columns among them.
// Floats of the dual sweep's LDS between the data-term window and the tap regions: per-wave partial sums and the
// kernel table, or (separable kernel at zoom 4) the row sums of the window and the two factors.
__host__ __device__ inline int dual_mid_floats(const Geom &g, bool sep) {
    const int f = g.f;
    const int psum = ((MMC_THREADS / 64) * (TILE_R / f) * (TILE_C / f) + 3) & ~3;
    if (!sep) return psum + g.MPq * g.CW;
    const int fe = (TILE_R % f == 0 && TILE_C % f == 0) ? f : 1;
    const int sums = ((((TILE_R + f - 1) / f) * (TILE_C + g.CW - fe) + 3) & ~3) + g.MPq + ((g.CW + 3) & ~3);
    return sums > psum ? sums : psum;
}

// Floats per plane of the tap regions.  The instances for separable kernels stage a larger data-term window
// (their kernels are the large ones); with 256 floats less per tap region five of their workgroups still fit a
// CU.  Tiles whose taps spread further gather them from global memory, as tiles beyond MMC_WARP_MAX always do.
__host__ __device__ constexpr int dual_tap_floats(bool sep) { return sep ? MMC_WARP_MAX - 256 : MMC_WARP_MAX; }

template <int MODEL, int MODE, bool REDUCE, int F, bool SEPK = false>
__global__ __launch_bounds__(MMC_THREADS, F == 4 ? MMC_OCC_DUAL : 5) void k_dual4(Geom g, Arrays a, int t0, int nf) {
    extern __shared__ __align__(16) float lds[];
    const int f = F > 0 ? F : g.f;
    const int tid = threadIdx.x;
    int tile, trel;
    if (!decode_block(g, nf, tile, trel, true)) return;
    // The primal sweep walks its (tile, frame) sequence upwards, the dual sweep walks the same sequence
    // downwards: each sweep starts on what the other one touched last, which is still in the Infinity Cache.
    const int t = t0 + trel;
    const int tc = fdiv(tile, g.dv_tiles_r), tr = tile - tc * g.tiles_r;
    const int r0 = tr * TILE_R, c0 = tc * TILE_C;
    const float kap = (MODEL == MODEL_IC) ? g.kappa : 1.f;
    const bool pex = pair_exists(g, t);
    const bool isIF = pair_is_IF(g, t);
    // Zoom 4: the window starts on a row quad (g.wpad extra rows on top, height g.SRp), so that it
    // is staged with 16-byte loads and each low-res pixel reads its kernel rows as aligned float4.
    // The padded window (16-byte staging) serves every zoom factor that divides the tile; the
    // lane-quad data term below is the zoom-4 form of it.
    const bool VEC = g.SRp <= 128 && !ABL(4096);
    const int fe = (TILE_R % f == 0 && TILE_C % f == 0) ? f : 1;      // 1: low-res blocks straddle tile borders
    const int SR = VEC ? g.SRp : TILE_R + g.CH - fe, SC = TILE_C + g.CW - fe;
    const int wpad = VEC ? g.wpad : 0;
    const int LRR = TILE_R / f, LRC = TILE_C / f, nlr = LRR * LRC;
    constexpr int NWV = MMC_THREADS / 64;
    float *ubl = lds;                                   // SR x SC window of xbar_u
    float *psum = lds + ((SR * SC + 3) & ~3);           // [NWV waves][nlr] partial sums of A xbar_u
    float *ckq = psum + ((NWV * nlr + 3) & ~3);         // composite kernel shifted down by wpad rows, MPq x CW
    // Separable composite kernel (Geom::sep): the row sums of the window and the two factors take the place of
    // the partial sums and the kernel table
    const bool SEP = SEPK && VEC;                       // the launcher picks SEPK from Geom::sep
    float *sl = psum;                                   // low-res rows of the tile x SC: sum_m ky[m] window[f R + m, column]
    float *kys = sl + ((((TILE_R + f - 1) / f) * SC + 3) & ~3);      // ky shifted down by wpad rows, MPq floats
    float *kxs = kys + g.MPq;                           // kx, CW floats
    float *wu = SEPK ? psum + dual_mid_floats(g, true) : ckq + g.MPq * g.CW;      // tap region of the warped frame, xbar_u
    float *ww = wu + dual_tap_floats(SEPK);             // ... and xbar_w
    const float *ub = FR(a.ub, t), *wb = FR(a.wb, t);
    const float *ubW = FR(a.ub, isIF ? t + 1 : t), *wbW = FR(a.wb, isIF ? t + 1 : t);   // warped frame
    // Rows/columns of the warped frame that this tile's taps touch (from mmc_init), rows widened
    // to whole quads.  If the region fits, it is staged in LDS with 16-byte loads and the 4 taps
    // per pixel are LDS reads; otherwise (violent flow) they are gathered from global memory.
    int wrlo = 0, wclo = 0, WR = 0, WC = 0, WP = 1;
    bool use_lds = false;
    if (pex && MODE != MODE_STEPS && !ABL(1024)) {
        const int *tb = a.tbox + ((size_t)t * g.ntiles + tile) * 4;
        if (tb[1] >= tb[0]) {
            wrlo = tb[0] & ~3;
            wclo = tb[2];
            WR = (tb[1] | 3) - wrlo + 1;
            WC = tb[3] - wclo + 1;
            // Odd pitch: a wave's 16 row quads x 4 columns read rows 4 apart, which an even pitch folds
            // onto 16 of the 64 LDS banks (4-way conflicts); with an odd pitch all 64 addresses differ mod 64.
            WP = WR + MMC_TAP_ODD_PITCH;
            use_lds = WP * WC <= dual_tap_floats(SEPK);
        }
    }

    const int rq = tid & (TILE_R / 4 - 1), lc = tid / (TILE_R / 4);
    const int r = r0 + 4 * rq, c = c0 + lc;
    const bool active = r < g.Ny && c < g.Nx;
    const int j = active ? r + g.Ny * c : 0;
    const int nv = F == 4 ? 4 : min(g.Ny - r, 4);       // rows of this quad inside the frame (zoom 4: Ny = 4 ny)
    const bool hasx = c < g.Nx - 1;

    // The data-term window of xbar_u also holds this quad's right neighbour column and the row below it
    // when the composite kernel reaches at least one pixel beyond the tile on those sides.
    const bool nb_lds = MMC_DUAL_UX_LDS && VEC && MODE != MODE_STEPS && g.oy <= 0 && g.ox <= 0 &&
                        g.oy + g.CH - fe - 1 >= 0 && g.ox + g.CW - fe - 1 >= 0;
    // own loads first; they stay in flight while the window is staged
    float o1t[4], o1x[4], o1y[4], o2t[4], o2x[4], o2y[4], fvx[4], fvy[4];
    float uc[4], ux[4], wc[4] = {0.f, 0.f, 0.f, 0.f}, wx[4] = {0.f, 0.f, 0.f, 0.f};
    float ubelow = 0.f, wbelow = 0.f;
    if (active) {
        if (pex) { ld4(fvx, FR(a.vx, t) + j); ld4(fvy, FR(a.vy, t) + j); }
        if (!(MMC_DUAL_UC_LDS && nb_lds)) ld4(uc, ub + j);
        if (!nb_lds) {
            if (hasx) ld4(ux, ub + j + g.Ny);
            if (r + 4 < g.Ny) ubelow = ub[j + 4];
        }
        if (MODEL == MODEL_IC) {
            ld4(wc, wb + j);
            if (hasx) ld4(wx, wb + j + g.Ny);
            if (r + 4 < g.Ny) wbelow = wb[j + 4];
        }
        if (MODE == MODE_UPDATE || MODE == MODE_ENERGY) {
            // a frame without a pair (the last of the sequence) has no temporal rows: their duals stay 0
            // and their planes are neither read nor written
            const bool tpl = pex || !MMC_SKIP_NOPAIR;
#pragma unroll
            for (int i = 0; i < 4; ++i) o1t[i] = o2t[i] = 0.f;
            if (tpl) ld4(o1t, FR(a.g1t, t) + j);
            ld4(o1x, FR(a.g1x, t) + j); ld4(o1y, FR(a.g1y, t) + j);
            if (MODEL == MODEL_IC) { if (tpl) ld4(o2t, FR(a.g2t, t) + j); ld4(o2x, FR(a.g2x, t) + j); ld4(o2y, FR(a.g2y, t) + j); }
        }
    }

    // stage the xbar_u window, replicate border (helpers/RepConvMtx.m:23-32)
    if (MODE != MODE_STEPS && !ABL(512)) {
        const int wr0 = r0 + g.oy - wpad, wc0 = c0 + g.ox;
        if (VEC && SEP) {
            if (tid < g.MPq) kys[tid] = a.ksep[2 * MMC_KSEP_PITCH + tid];
            if (tid >= 64 && tid - 64 < g.CW) kxs[tid - 64] = a.ksep[MMC_KSEP_PITCH + tid - 64];
        } else if (VEC) {
            for (int k = tid; k < g.MPq * g.CW; k += MMC_THREADS) ckq[k] = a.ckq[k];
        }
        if (VEC && wr0 >= 0 && wr0 + SR <= g.Ny && wc0 >= 0 && wc0 + SC <= g.Nx) {
            const int kq = tid & 31;            // SR <= 128: at most 32 quads per window column
            if (4 * kq < SR) {
                const float *src = ub + wr0 + 4 * kq + g.Ny * wc0;
                for (int kc = tid >> 5; kc < SC; kc += MMC_THREADS / 32)
                    { float v4[4]; ld4(v4, src + g.Ny * kc); *reinterpret_cast<float4 *>(ubl + 4 * kq + SR * kc) = make_float4(v4[0], v4[1], v4[2], v4[3]); }
            }
        } else {
            const float inv_sr = fast_rcp((float)SR);
            for (int k = tid; k < SR * SC; k += MMC_THREADS) {
                int kc = (int)(((float)k + 0.5f) * inv_sr);
                int kr = k - kc * SR;
                if (kr < 0) { --kc; kr += SR; }
                if (kr >= SR) { ++kc; kr -= SR; }
                const int rr = min(max(wr0 + kr, 0), g.Ny - 1);
                const int cc = min(max(wc0 + kc, 0), g.Nx - 1);
                ubl[k] = ub[rr + g.Ny * cc];
            }
        }
    }
    if (use_lds) {
        const int QR = WR >> 2, nq = QR * WC;
        const float inv_qr = fast_rcp((float)QR);
        for (int k = tid; k < nq; k += MMC_THREADS) {
            int kc = (int)(((float)k + 0.5f) * inv_qr);
            int kr = k - kc * QR;
            if (kr < 0) { --kc; kr += QR; }
            if (kr >= QR) { ++kc; kr -= QR; }
            const int src = wrlo + 4 * kr + g.Ny * (wclo + kc);
            float t4[4];
            ld4(t4, ubW + src);
            const float4 vu = make_float4(t4[0], t4[1], t4[2], t4[3]);
            float *du = wu + 4 * kr + WP * kc;
            if (MMC_TAP_ODD_PITCH) { du[0] = vu.x; du[1] = vu.y; du[2] = vu.z; du[3] = vu.w; }
            else *reinterpret_cast<float4 *>(du) = vu;
            if (MODEL == MODEL_IC) {
                ld4(t4, wbW + src);
                const float4 vw = make_float4(t4[0], t4[1], t4[2], t4[3]);
                float *dw = ww + 4 * kr + WP * kc;
                if (MMC_TAP_ODD_PITCH) { dw[0] = vw.x; dw[1] = vw.y; dw[2] = vw.z; dw[3] = vw.w; }
                else *reinterpret_cast<float4 *>(dw) = vw;
            }
        }
    }
    if (MODE != MODE_STEPS) __syncthreads();

    float part[2] = {0.f, 0.f};
    float wres[1] = {0.f};           // sum ((y+ - y) / sigma)^2: the step-weighted change the step balancing compares
    float diag[2] = {0.f, 0.f};      // MODE_ENERGY: |K x|^2, dual optimality residual^2
    if (active && nb_lds) {
        const float *own = ubl + (4 * rq - g.oy + wpad) + SR * (lc - g.ox);
        if (MMC_DUAL_UC_LDS) ld4(uc, own);
        if (hasx) ld4(ux, own + SR);
        if (r + 4 < g.Ny) ubelow = own[4];
    }
    if (active) {
        float k1t[4], k1x[4], k1y[4], k2t[4], k2x[4], k2y[4], sg1[4], sg2[4];
        bool valid[4];
        float idu[4], idw[4];
        if (pex && !isIF) {   // identity block sits on the next frame
            ld4(idu, FR(a.ub, t + 1) + j);
            if (MODEL == MODEL_IC) ld4(idw, FR(a.wb, t + 1) + j);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool hasy = r + i < g.Ny - 1;
            const float uy = i < 3 ? uc[i < 3 ? i + 1 : 3] : ubelow;
            const float wy = i < 3 ? wc[i < 3 ? i + 1 : 3] : wbelow;
            float tu = 0.f, tw = 0.f;
            valid[i] = false;
            if (pex && i < nv && !ABL(256)) {
                Taps tp = make_taps(g, r + i, c, fvx[i], fvy[i]);
                valid[i] = tp.valid;
                if (tp.valid) {
                    // taps inside the staged region read LDS; a stray vector gathers from global memory
                    const bool in_lds = use_lds && tp.r0 >= wrlo && tp.r1 < wrlo + WR && tp.c0 >= wclo && tp.c1 < wclo + WC;
                    if (in_lds) {
                        tu = sample_lds(tp, wu, wrlo, wclo, WP) - (isIF ? uc[i] : idu[i]);
                        if (MODEL == MODEL_IC) tw = sample_lds(tp, ww, wrlo, wclo, WP) - (isIF ? wc[i] : idw[i]);
                    } else {
                        tu = sample(tp, ubW) - (isIF ? uc[i] : idu[i]);
                        if (MODEL == MODEL_IC) tw = sample(tp, wbW) - (isIF ? wc[i] : idw[i]);
                    }
                }
            }
            float rs1, rs2 = 0.f;
            if (MODEL == MODEL_IC) {
                // K1 (u - w), K1 = [W/h ; kappa G]  (:490-491);  K2 w, K2 = [kappa W/h ; G]  (:492);
                // g.wsgn = +1: temporal row on u + w, the 'backward' terms of init_u_flexBox (:864)
                const float dc = uc[i] - wc[i];
                k1t[i] = (tu + g.wsgn * tw) * g.inv_h;
                k1x[i] = hasx ? kap * ((ux[i] - wx[i]) - dc) : 0.f;
                k1y[i] = hasy ? kap * ((uy - wy) - dc) : 0.f;
                k2t[i] = kap * tw * g.inv_h;
                k2x[i] = hasx ? wx[i] - wc[i] : 0.f;
                k2y[i] = hasy ? wy - wc[i] : 0.f;
                rs1 = fmaxf(valid[i] ? 4.f * g.inv_h : 0.f, fmaxf(hasx ? 4.f * kap : 0.f, hasy ? 4.f * kap : 0.f));
                rs2 = fmaxf(valid[i] ? 2.f * kap * g.inv_h : 0.f, fmaxf(hasx ? 2.f : 0.f, hasy ? 2.f : 0.f));
            } else {
                k1t[i] = tu * g.inv_h;
                k1x[i] = hasx ? ux[i] - uc[i] : 0.f;
                k1y[i] = hasy ? uy - uc[i] : 0.f;
                k2t[i] = k2x[i] = k2y[i] = 0.f;
                if (MODEL == MODEL_K1) {
                    rs1 = fmaxf(valid[i] ? 2.f * g.inv_h : 0.f, fmaxf(hasx ? 2.f : 0.f, hasy ? 2.f : 0.f));
                } else {
                    rs1 = valid[i] ? 2.f * g.inv_h : 0.f;
                    rs2 = fmaxf(hasx ? 2.f : 0.f, hasy ? 2.f : 0.f);
                }
            }
            sg1[i] = g.sigma_s * (rs1 > 0.f ? fast_rcp(rs1) : 1.f);
            sg2[i] = g.sigma_s * (rs2 > 0.f ? fast_rcp(rs2) : 1.f);
        }
        float *g1t = FR(a.g1t, t) + j, *g1x = FR(a.g1x, t) + j, *g1y = FR(a.g1y, t) + j;
        float *g2t = FR(a.g2t, t) + j, *g2x = FR(a.g2x, t) + j, *g2y = FR(a.g2y, t) + j;
        if (MODE == MODE_RAW) {
            st4m(g1t, k1t, nv); st4m(g1x, k1x, nv); st4m(g1y, k1y, nv);
            if (MODEL == MODEL_IC) { st4m(g2t, k2t, nv); st4m(g2x, k2x, nv); st4m(g2y, k2y, nv); }
        } else if (MODE == MODE_STEPS) {
            st4m(g1t, sg1, nv);
            st4m(g2t, sg2, nv);
        } else if (MODE == MODE_ENERGY) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i >= nv) continue;
                diag[0] += k1t[i] * k1t[i] + k1x[i] * k1x[i] + k1y[i] * k1y[i] + k2t[i] * k2t[i] + k2x[i] * k2x[i] + k2y[i] * k2y[i];
                if (MODEL == MODEL_IC) {
                    part[0] += g.radius1 * sqrtf(k1t[i] * k1t[i] + k1x[i] * k1x[i] + k1y[i] * k1y[i]) +
                               g.radius2 * sqrtf(k2t[i] * k2t[i] + k2x[i] * k2x[i] + k2y[i] * k2y[i]);
                    diag[1] += ball_residual2(k1t[i], k1x[i], k1y[i], o1t[i], o1x[i], o1y[i], g.radius1) +
                               ball_residual2(k2t[i], k2x[i], k2y[i], o2t[i], o2x[i], o2y[i], g.radius2);
                } else if (MODEL == MODEL_K1) {
                    part[0] += g.radius1 * sqrtf(k1t[i] * k1t[i] + k1x[i] * k1x[i] + k1y[i] * k1y[i]);
                    diag[1] += ball_residual2(k1t[i], k1x[i], k1y[i], o1t[i], o1x[i], o1y[i], g.radius1);
                } else {
                    part[0] += g.radius1 * fabsf(k1t[i]) + g.radius2 * sqrtf(k1x[i] * k1x[i] + k1y[i] * k1y[i]);
                    diag[1] += box_residual2(k1t[i], o1t[i], g.radius1) + ball_residual2(0.f, k1x[i], k1y[i], 0.f, o1x[i], o1y[i], g.radius2);
                }
            }
            if (MODEL == MODEL_IC && g.fixed && g.frame_begin + t == 0) {      // identity rows of q1, q2: dF* = {x0}
                float x0u[4], x0w[4];
                ld4(x0u, a.u0 + j); ld4(x0w, a.w0 + j);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i >= nv) continue;
                    diag[0] += uc[i] * uc[i] + wc[i] * wc[i];
                    diag[1] += (uc[i] - x0u[i]) * (uc[i] - x0u[i]) + (wc[i] - x0w[i]) * (wc[i] - x0w[i]);
                }
            }
        } else {
            float n1t[4], n1x[4], n1y[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (MODEL == MODEL_ADD) {
                    // |g1| <= alpha (:532) on the temporal row, ||(gx,gy)|| <= alpha (:533)
                    n1t[i] = valid[i] ? fminf(fmaxf(o1t[i] + sg1[i] * k1t[i], -g.radius1), g.radius1) : 0.f;
                    n1x[i] = o1x[i] + sg2[i] * k1x[i];
                    n1y[i] = o1y[i] + sg2[i] * k1y[i];
                    const float sc = ball_scale(n1x[i] * n1x[i] + n1y[i] * n1y[i], g.radius2);
                    n1x[i] *= sc; n1y[i] *= sc;
                } else {
                    // ||(gt,gx,gy)||_2 <= radius (:497, :513)
                    n1t[i] = valid[i] ? o1t[i] + sg1[i] * k1t[i] : 0.f;
                    n1x[i] = o1x[i] + sg1[i] * k1x[i];
                    n1y[i] = o1y[i] + sg1[i] * k1y[i];
                    const float sc = ball_scale(n1t[i] * n1t[i] + n1x[i] * n1x[i] + n1y[i] * n1y[i], g.radius1);
                    n1t[i] *= sc; n1x[i] *= sc; n1y[i] *= sc;
                }
                if (REDUCE && i < nv) {
                    const float dt = n1t[i] - o1t[i], dx = n1x[i] - o1x[i], dy = n1y[i] - o1y[i];
                    part[0] += dt * dt + dx * dx + dy * dy;
                    part[1] += n1t[i] * n1t[i] + n1x[i] * n1x[i] + n1y[i] * n1y[i];
                    const float sxy = MODEL == MODEL_ADD ? sg2[i] : sg1[i];
                    wres[0] += (dt / sg1[i]) * (dt / sg1[i]) + ((dx * dx + dy * dy) / sxy) / sxy;
                }
            }
            if (pex || !MMC_SKIP_NOPAIR) st4m(g1t, n1t, nv);
            st4m(g1x, n1x, nv); st4m(g1y, n1y, nv);
            if (MODEL == MODEL_IC) {
                float n2t[4], n2x[4], n2y[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    n2t[i] = valid[i] ? o2t[i] + sg2[i] * k2t[i] : 0.f;
                    n2x[i] = o2x[i] + sg2[i] * k2x[i];
                    n2y[i] = o2y[i] + sg2[i] * k2y[i];
                    const float sc = ball_scale(n2t[i] * n2t[i] + n2x[i] * n2x[i] + n2y[i] * n2y[i], g.radius2);
                    n2t[i] *= sc; n2x[i] *= sc; n2y[i] *= sc;
                    if (REDUCE && i < nv) {
                        const float dt = n2t[i] - o2t[i], dx = n2x[i] - o2x[i], dy = n2y[i] - o2y[i];
                        part[0] += dt * dt + dx * dx + dy * dy;
                        part[1] += n2t[i] * n2t[i] + n2x[i] * n2x[i] + n2y[i] * n2y[i];
                        wres[0] += ((dt * dt + dx * dx + dy * dy) / sg2[i]) / sg2[i];
                    }
                }
                if (pex || !MMC_SKIP_NOPAIR) st4m(g2t, n2t, nv);
                st4m(g2x, n2x, nv); st4m(g2y, n2y, nv);
                if (g.fixed && g.frame_begin + t == 0) {
                    // q <- q + (xbar - x0): F*(q) = <x0, q>, identity rows, step 1 (:560-564)
                    float q1o[4], q2o[4], x0u[4], x0w[4], q1n[4], q2n[4];
                    ld4(q1o, a.q1 + j); ld4(q2o, a.q2 + j); ld4(x0u, a.u0 + j); ld4(x0w, a.w0 + j);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        q1n[i] = q1o[i] + g.sigma_s * (uc[i] - x0u[i]);
                        q2n[i] = q2o[i] + g.sigma_s * (wc[i] - x0w[i]);
                        if (REDUCE && i < nv) {
                            const float d1 = q1n[i] - q1o[i], d2 = q2n[i] - q2o[i];
                            part[0] += d1 * d1 + d2 * d2;
                            part[1] += q1n[i] * q1n[i] + q2n[i] * q2n[i];
                            wres[0] += ((d1 * d1 + d2 * d2) / g.sigma_s) / g.sigma_s;
                        }
                    }
                    st4m(a.q1 + j, q1n, nv); st4m(a.q2 + j, q2n, nv);
                }
            }
        }
    }

    // ---- data term: p <- clamp(p + s (A xbar_u - f), -1, 1)   (:496) --------
    float epart = 0.f;
    // Zoom 4: four adjacent lanes share a low-res pixel and split the columns of the composite
    // kernel among them (lane part takes columns part, part+4, ...); two lane exchanges add the
    // four partial sums, lane 0 of the group updates p.  No second barrier, no partial sums in LDS;
    // the loads of p, f and the step size are in flight while the sums are formed.
    const bool QUAD = MMC_DATA_QUAD && F == 4 && VEC && 4 * nlr == MMC_THREADS;
    if (QUAD && MODE != MODE_STEPS && !ABL(512)) {
        const int lp = tid >> 2, part4 = tid & 3;
        const int Rl = lp % LRR, Cl = lp / LRR;
        const int R = r0 / f + Rl, C = c0 / f + Cl;
        const bool owner = part4 == 0 && R < g.ny && C < g.nx;
        const size_t o = (size_t)t * g.n + (owner ? R + g.ny * C : 0);
        float po = 0.f, fo = 0.f, sg = 0.f;
        if (owner) {
            if (MODE == MODE_UPDATE) sg = g.sigma_s * a.sigp[R + g.ny * C];
            if (MODE == MODE_UPDATE || MODE == MODE_ENERGY) { po = a.p[o]; fo = a.f[o]; }
        }
        const int mq = g.MPq >> 2;
        float s = 0.f;
        if (SEP) {
            // first factor down the rows of every window column, for the 16 low-res rows of the tile; then each
            // lane takes every fourth column of the second factor
            const float4 *kw = reinterpret_cast<const float4 *>(kys);
#pragma unroll 1
            for (int k = tid; k < LRR * SC; k += MMC_THREADS) {
                const float4 *col = reinterpret_cast<const float4 *>(ubl + f * (k % LRR) + SR * (k / LRR));
                float sr = 0.f;
#pragma unroll 1
                for (int m4 = 0; m4 < mq; ++m4) {
                    const float4 cv = kw[m4], dv = col[m4];
                    sr += cv.x * dv.x + cv.y * dv.y + cv.z * dv.z + cv.w * dv.w;
                }
                sl[k] = sr;
            }
            __syncthreads();
            const float *srow = sl + Rl + LRR * (f * Cl);
#pragma unroll 1
            for (int nn = part4; nn < g.CW; nn += 4) s += kxs[nn] * srow[LRR * nn];
        } else {
            for (int nn = part4; nn < g.CW; nn += 4) {
                const float4 *col = reinterpret_cast<const float4 *>(ubl + f * Rl + SR * (f * Cl + nn));
                const float4 *ck = reinterpret_cast<const float4 *>(ckq + g.MPq * nn);
                for (int m4 = 0; m4 < mq; ++m4) {
                    const float4 cv = ck[m4], dv = col[m4];
                    s += cv.x * dv.x + cv.y * dv.y + cv.z * dv.z + cv.w * dv.w;
                }
            }
        }
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        if (owner) {
            if (a.pcorr) s += a.pcorr[o];
            if (MODE == MODE_RAW) {
                a.p[o] = s;
            } else if (MODE == MODE_ENERGY) {
                epart += fabsf(s - fo);
                diag[0] += s * s;
                diag[1] += box_residual2(s - fo, po, 1.f);
            } else {
                const float pn = fminf(fmaxf(po + sg * (s - fo), -1.f), 1.f);
                a.p[o] = pn;
                if (REDUCE) { part[0] += (pn - po) * (pn - po); part[1] += pn * pn; wres[0] += ((pn - po) / sg) * ((pn - po) / sg); }
            }
        }
    }
    // Every other zoom factor.  A low-res pixel belongs to the tile that holds the first high-res pixel
    // of its block (f R, f C), so factors that do not divide the tile (zoom 3) work as well; the window
    // then reaches f - 1 rows / columns further (g.SRp and SC allow for it).  LPP adjacent lanes share a
    // low-res pixel (zoom 2: one lane each, zoom 8: 16 lanes), split the kernel columns among them and add
    // their partial sums by lane exchanges, as the quads do.  Kernel weights come from the copy in LDS
    // (broadcast reads): as loads from global memory inside these loops they cost a third of the sweep.
    const int LRRm = (TILE_R + f - 1) / f, LRCm = (TILE_C + f - 1) / f;      // most low-res rows / columns a tile owns
    int LPP = 1;
    while (2 * LPP * LRRm * LRCm <= MMC_THREADS && LPP < 64) LPP *= 2;
    const bool LANES = F != 4 && !QUAD && VEC;
    if (LANES && !ABL(512)) {
        const int LR0 = (r0 + f - 1) / f, LC0 = (c0 + f - 1) / f;              // first low-res row / column of this tile
        const int LRRt = max(min((r0 + TILE_R + f - 1) / f, g.ny) - LR0, 0);
        const int LRCt = max(min((c0 + TILE_C + f - 1) / f, g.nx) - LC0, 0);
        const int nlrt = LRRt * LRCt;
        const int sub = tid & (LPP - 1);
        const float inv_lrr = fast_rcp((float)max(LRRt, 1));
        if (SEP && MODE != MODE_STEPS) {
            // separable kernel: first factor down the rows of every window column for the tile's low-res rows
            const int nout = LRRt * SC;
#pragma unroll 1
            for (int k = tid; k < nout; k += MMC_THREADS) {
                int cc = (int)(((float)k + 0.5f) * inv_lrr);
                int Rl = k - cc * LRRt;
                if (Rl < 0) { --cc; Rl += LRRt; }
                if (Rl >= LRRt) { ++cc; Rl -= LRRt; }
                const float *col = ubl + (f * (LR0 + Rl) - r0) + SR * cc;
                float sr = 0.f;
                if ((f & 1) == 0) {
                    const float2 *c2 = reinterpret_cast<const float2 *>(col), *k2 = reinterpret_cast<const float2 *>(kys);
#pragma unroll 1
                    for (int m = 0; m < (g.MPq >> 1); ++m) sr += k2[m].x * c2[m].x + k2[m].y * c2[m].y;
                } else {
#pragma unroll 1
                    for (int mm = 0; mm < g.CH; ++mm) sr += kys[wpad + mm] * col[wpad + mm];
                }
                sl[Rl + LRRt * cc] = sr;
            }
            __syncthreads();
        }
        for (int lp = tid / LPP; lp < nlrt; lp += MMC_THREADS / LPP) {
            int Cl = (int)(((float)lp + 0.5f) * inv_lrr);
            int Rl = lp - Cl * LRRt;
            if (Rl < 0) { --Cl; Rl += LRRt; }
            if (Rl >= LRRt) { ++Cl; Rl -= LRRt; }
            const int R = LR0 + Rl, C = LC0 + Cl;
            const int i = R + g.ny * C;
            const size_t o = (size_t)t * g.n + i;
            const bool owner = sub == 0;
            if (MODE == MODE_STEPS) { if (owner) a.p[o] = g.sigma_s * a.sigp[i]; continue; }
            float po = 0.f, fo = 0.f, sg = 0.f;
            if (owner) {
                if (MODE == MODE_UPDATE) sg = g.sigma_s * a.sigp[i];
                if (MODE == MODE_UPDATE || MODE == MODE_ENERGY) { po = a.p[o]; fo = a.f[o]; }
            }
            const float *win = ubl + (f * R - r0) + SR * (f * C - c0);          // window row of kernel row -wpad
            float s = 0.f;
            if (SEP) {
                const float *srow = sl + Rl + LRRt * (f * C - c0);
#pragma unroll 1
                for (int nn = sub; nn < g.CW; nn += LPP) s += kxs[nn] * srow[LRRt * nn];
            } else if ((f & 1) == 0) {
                // even zoom: these rows start on an even row of the window: 8-byte reads against the
                // kernel table that is shifted down by the padding rows
                const int m2 = g.MPq >> 1;
                for (int nn = sub; nn < g.CW; nn += LPP) {
                    const float2 *col = reinterpret_cast<const float2 *>(win + SR * nn);
                    const float2 *ck = reinterpret_cast<const float2 *>(ckq + g.MPq * nn);
                    for (int k = 0; k < m2; ++k) {
                        const float2 v = col[k], c2 = ck[k];
                        s += c2.x * v.x + c2.y * v.y;
                    }
                }
            } else {
                for (int nn = sub; nn < g.CW; nn += LPP) {
                    const float *col = win + wpad + SR * nn;
                    const float *ck = ckq + wpad + g.MPq * nn;
                    for (int mm = 0; mm < g.CH; ++mm) s += ck[mm] * col[mm];
                }
            }
            for (int off = 1; off < LPP; off <<= 1) s += __shfl_xor(s, off, 64);
            if (owner) {
                if (a.pcorr) s += a.pcorr[o];
                if (MODE == MODE_RAW) {
                    a.p[o] = s;
                } else if (MODE == MODE_ENERGY) {
                    epart += fabsf(s - fo);
                    diag[0] += s * s;
                    diag[1] += box_residual2(s - fo, po, 1.f);
                } else {
                    const float pn = fminf(fmaxf(po + sg * (s - fo), -1.f), 1.f);
                    a.p[o] = pn;
                    if (REDUCE) { part[0] += (pn - po) * (pn - po); part[1] += pn * pn; wres[0] += ((pn - po) / sg) * ((pn - po) / sg); }
                }
            }
        }
    }
    if (!QUAD && !LANES && MODE != MODE_STEPS && !ABL(512)) {
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
        for (int lp = lane; lp < nlr; lp += 64) {
            const int Rl = lp % LRR, Cl = lp / LRR;
            float s = 0.f;
            if (VEC && F == 4) {
                // lanes are 16 bytes apart (conflict-free), the kernel rows are a broadcast
                const int mq = g.MPq >> 2;
                for (int nn = wv; nn < g.CW; nn += NWV) {
                    const float4 *col = reinterpret_cast<const float4 *>(ubl + f * Rl + SR * (f * Cl + nn));
                    const float4 *ck = reinterpret_cast<const float4 *>(ckq + g.MPq * nn);
                    for (int m4 = 0; m4 < mq; ++m4) {
                        const float4 cv = ck[m4], dv = col[m4];
                        s += cv.x * dv.x + cv.y * dv.y + cv.z * dv.z + cv.w * dv.w;
                    }
                }
            } else {
                for (int nn = wv; nn < g.CW; nn += NWV) {
                    const float *col = ubl + wpad + f * Rl + SR * (f * Cl + nn);
                    const float *ck = a.cker + g.CH * nn;
                    for (int mm = 0; mm < g.CH; ++mm) s += ck[mm] * col[mm];
                }
            }
            psum[wv * nlr + lp] = s;
        }
        __syncthreads();
    }
    for (int lp = tid; lp < nlr; lp += MMC_THREADS) {
        if (LANES || (QUAD && MODE != MODE_STEPS)) break;
        const int Rl = lp % LRR, Cl = lp / LRR;
        const int R = r0 / f + Rl, C = c0 / f + Cl;
        if (R >= g.ny || C >= g.nx) continue;
        const int i = R + g.ny * C;
        const size_t o = (size_t)t * g.n + i;
        const float sg = g.sigma_s * a.sigp[i];
        if (MODE == MODE_STEPS) { a.p[o] = sg; continue; }
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < MMC_THREADS / 64; ++k) acc += psum[k * nlr + lp];
        if (a.pcorr) acc += a.pcorr[o];
        if (MODE == MODE_RAW) {
            a.p[o] = acc;
        } else if (MODE == MODE_ENERGY) {
            epart += fabsf(acc - a.f[o]);
            diag[0] += acc * acc;
            diag[1] += box_residual2(acc - a.f[o], a.p[o], 1.f);
        } else {
            const float po = a.p[o];
            const float pn = fminf(fmaxf(po + sg * (acc - a.f[o]), -1.f), 1.f);
            a.p[o] = pn;
            if (REDUCE) { part[0] += (pn - po) * (pn - po); part[1] += pn * pn; wres[0] += ((pn - po) / sg) * ((pn - po) / sg); }
        }
    }
    if (MODE == MODE_ENERGY) { float e[4] = {part[0], epart, diag[0], diag[1]}; block_accumulate<4>(e, a.sums, SUM_ENERGY); }
    else if (REDUCE) { block_accumulate<2>(part, a.sums, SUM_RES + 2); __syncthreads(); block_accumulate<1>(wres, a.sums, SUM_WRES + 1); }
}

// --------------------------------------------------------------------------
// sparse border correction of the data operator (samplers whose rows near the
// frame border deviate from the interior taps): pcorr = E xbar_u, ucorr = E' p
// --------------------------------------------------------------------------
// A row of E has up to a few hundred entries (the patch of a low-res pixel): 16 adjacent lanes share a row,
// take its entries 16 apart (their index and value loads coalesce) and add their partial sums by lane exchanges.
constexpr int CORR_P_LANES = 16;
__global__ __launch_bounds__(MMC_THREADS) void k_corr_p(Geom g, Arrays a, int t0, int nf) {
    const int idx = (blockIdx.x * MMC_THREADS + threadIdx.x) / CORR_P_LANES, sub = threadIdx.x & (CORR_P_LANES - 1);
    const bool live = idx < a.ne * nf;          // whole groups of 16 lanes are live or not; all lanes reach the exchanges
    const int tr = live ? idx / a.ne : 0, e = live ? idx - tr * a.ne : 0, t = t0 + tr;
    const float *ub = FR(a.ub, t);
    float s = 0.f;
    if (live)
        for (int k = a.e_ptr[e] + sub; k < a.e_ptr[e + 1]; k += CORR_P_LANES) s += a.e_val[k] * ub[a.e_col[k]];
#pragma unroll
    for (int off = 1; off < CORR_P_LANES; off <<= 1) s += __shfl_xor(s, off, 64);
    if (live && sub == 0) a.pcorr[(size_t)t * g.n + a.e_row[e]] = s;
}

__global__ __launch_bounds__(MMC_THREADS) void k_corr_u(Geom g, Arrays a, int t0, int nf) {
    const int idx = blockIdx.x * MMC_THREADS + threadIdx.x;
    if (idx >= a.net * nf) return;
    const int tr = idx / a.net, e = idx - tr * a.net, t = t0 + tr;
    const float *p = a.p + (size_t)t * g.n;
    float s = 0.f;
    for (int k = a.et_ptr[e]; k < a.et_ptr[e + 1]; ++k) s += a.et_val[k] * p[a.et_col[k]];
    a.ucorr[(size_t)t * g.N + a.et_row[e]] = s;
}

// --------------------------------------------------------------------------
// one-off kernels
// --------------------------------------------------------------------------

// Column abs-sums of the motion-coupling operator and the per-tile source regions of its
// adjoint.  For a target tile T of the warped frame, the sources are the pixels of pair q with a
// tap inside T.  With a regular flow they fill a rectangle the size of the tile, displaced by the
// flow; a single stray vector (estimators produce them at occlusions) would blow its bounding box
// up, so the region is described as a rectangle plus a list:
//   pass 1  k_init_scatter  tap weights summed per target pixel in 64-bit fixed point
//                           (order-independent), and per tile the sum of the source-minus-tap
//                           displacements and the number of sources
//   pass 2  k_init_srcbox   the rectangle: bounding box of the sources whose displacement lies
//                           within MMC_SRC_MR rows / MMC_SRC_MC columns of the tile's mean
//   pass 3  k_init_overflow<false>  number of sources of T that lie outside its rectangle
//   pass 4  k_init_overflow<true>   their pixel indices, after a prefix sum on the host
// The update sweep walks the rectangle (every pixel in it, whatever its displacement) and then
// the list, so each source of T is visited exactly once.
template <typename F>
__device__ __forceinline__ void for_each_target_tile(const Geom &g, const Taps &tp, F fn) {
    const int tr0 = tp.r0 / TILE_R, tr1 = tp.r1 / TILE_R, tc0 = tp.c0 / TILE_C, tc1 = tp.c1 / TILE_C;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ttr = (k & 1) ? tr1 : tr0, ttc = (k & 2) ? tc1 : tc0;
        if (k == 1 && tr1 == tr0) continue;
        if (k == 2 && tc1 == tc0) continue;
        if (k == 3 && (tr1 == tr0 || tc1 == tc0)) continue;
        fn(ttr + g.tiles_r * ttc);
    }
}

__global__ __launch_bounds__(MMC_THREADS) void k_init_scatter(Geom g, Arrays a, int q, fx64 *tmp) {
    const int j = blockIdx.x * MMC_THREADS + threadIdx.x;
    if (j >= g.N) return;
    const int r = j % g.Ny, c = j / g.Ny;
    Taps tp = make_taps(g, r, c, FR(a.vx, q)[j], FR(a.vy, q)[j]);
    if (!tp.valid) return;
    atomicAdd(&tmp[tp.i00], to_fx((double)tp.w00, 1099511627776.0));
    atomicAdd(&tmp[tp.i10], to_fx((double)tp.w10, 1099511627776.0));
    atomicAdd(&tmp[tp.i01], to_fx((double)tp.w01, 1099511627776.0));
    atomicAdd(&tmp[tp.i11], to_fx((double)tp.w11, 1099511627776.0));
    fx64 *st = a.bstat + (size_t)(q + 1) * g.ntiles * 3;
    for_each_target_tile(g, tp, [&](int tile) {
        fx64 *b = st + (size_t)tile * 3;
        atomicAdd(&b[0], (fx64)(long long)(r - tp.r0));
        atomicAdd(&b[1], (fx64)(long long)(c - tp.c0));
        atomicAdd(&b[2], (fx64)1);
    });
}

// mean displacement (rounded down) of the sources of a tile
__device__ __forceinline__ void tile_mean_disp(const fx64 *b, int &dr, int &dc) {
    const long long n = (long long)b[2];
    const long long sr = (long long)b[0], sc = (long long)b[1];
    dr = (int)floor((double)sr / (double)n);
    dc = (int)floor((double)sc / (double)n);
}

__global__ __launch_bounds__(MMC_THREADS) void k_init_srcbox(Geom g, Arrays a, int q) {
    const int j = blockIdx.x * MMC_THREADS + threadIdx.x;
    if (j >= g.N) return;
    const int r = j % g.Ny, c = j / g.Ny;
    Taps tp = make_taps(g, r, c, FR(a.vx, q)[j], FR(a.vy, q)[j]);
    if (!tp.valid) return;
    const fx64 *st = a.bstat + (size_t)(q + 1) * g.ntiles * 3;
    int *bb = a.bbox + (size_t)(q + 1) * g.ntiles * 4;
    for_each_target_tile(g, tp, [&](int tile) {
        int dr, dc;
        tile_mean_disp(st + (size_t)tile * 3, dr, dc);
        if (abs(r - tp.r0 - dr) > MMC_SRC_MR || abs(c - tp.c0 - dc) > MMC_SRC_MC) return;
        int *b = bb + (size_t)tile * 4;
        atomicMin(&b[0], r); atomicMax(&b[1], r);
        atomicMin(&b[2], c); atomicMax(&b[3], c);
    });
}

// FILL = false: ovf_ptr[slot + 1] += 1 per source outside the rectangle of its target tile (the
// host turns the counts into offsets); FILL = true: the indices, cursor[] counts what is placed.
template <bool FILL>
__global__ __launch_bounds__(MMC_THREADS) void k_init_overflow(Geom g, Arrays a, int q, int *cursor) {
    const int j = blockIdx.x * MMC_THREADS + threadIdx.x;
    if (j >= g.N) return;
    const int r = j % g.Ny, c = j / g.Ny;
    Taps tp = make_taps(g, r, c, FR(a.vx, q)[j], FR(a.vy, q)[j]);
    if (!tp.valid) return;
    const int *bb = a.bbox + (size_t)(q + 1) * g.ntiles * 4;
    for_each_target_tile(g, tp, [&](int tile) {
        const int *b = bb + (size_t)tile * 4;
        if (r >= b[0] && r <= b[1] && c >= b[2] && c <= b[3]) return;
        const size_t slot = (size_t)(q + 1) * g.ntiles + tile;
        if (FILL) a.ovf_idx[a.ovf_ptr[slot] + atomicAdd(&cursor[slot], 1)] = j;
        else atomicAdd(&cursor[slot], 1);
    });
}

// Pass 2 for frame t: cW = scattered weights + 1 for every unmasked row whose
// -Id entry sits on this frame (pair t if 'I-F', pair t-1 if 'F-I').
__global__ __launch_bounds__(MMC_THREADS) void k_init_finalize(Geom g, Arrays a, int t, const fx64 *tmp) {
    const int j = blockIdx.x * MMC_THREADS + threadIdx.x;
    if (j >= g.N) return;
    const int r = j % g.Ny, c = j / g.Ny;
    int nid = 0;
    if (pair_exists(g, t) && pair_is_IF(g, t) && make_taps(g, r, c, FR(a.vx, t)[j], FR(a.vy, t)[j]).valid) ++nid;
    if (pair_exists(g, t - 1) && !pair_is_IF(g, t - 1) &&
        make_taps(g, r, c, FR(a.vx, t - 1)[j], FR(a.vy, t - 1)[j]).valid) ++nid;
    FR(a.cW, t)[j] = from_fx_plus(tmp[j], nid);
}

// Per row quad of frame t: how many unmasked -Id entries of W sit in each of its four columns
// of the operator (0, 1 or 2), two bits per row.  With the scattered tap weights this is cW.
__global__ __launch_bounds__(MMC_THREADS) void k_init_idq(Geom g, Arrays a, int t) {
    const int qi = blockIdx.x * MMC_THREADS + threadIdx.x;
    const int qpc = (g.Ny + 3) >> 2;                    // quads per column; the last one may be short
    if (qi >= qpc * g.Nx) return;
    const int c = qi / qpc, r = 4 * (qi - c * qpc), j = r + g.Ny * c;
    const bool useA = pair_exists(g, t) && pair_is_IF(g, t);
    const bool useB = pair_exists(g, t - 1) && !pair_is_IF(g, t - 1);
    unsigned bits = 0;
    for (int i = 0; i < 4 && r + i < g.Ny; ++i) {
        unsigned n = 0;
        if (useA && make_taps(g, r + i, c, FR(a.vx, t)[j + i], FR(a.vy, t)[j + i]).valid) ++n;
        if (useB && make_taps(g, r + i, c, FR(a.vx, t - 1)[j + i], FR(a.vy, t - 1)[j + i]).valid) ++n;
        bits |= n << (2 * i);
    }
    a.idq[(size_t)t * MMC_IDQ_PER_FRAME(g) + qi] = (uint8_t)bits;
}

// Per (pair, tile): the rows and columns of the warped frame that the dual sweep stages in LDS for
// the taps of the tile's pixels: the bounding box of the taps whose displacement lies within
// MMC_TAP_MR rows / MMC_TAP_MC columns of the tile's mean displacement.  Pixels whose taps leave
// that region (stray vectors, motion boundaries) gather from global memory in the sweep.
__global__ __launch_bounds__(MMC_THREADS) void k_init_tapbox(Geom g, Arrays a) {
    __shared__ int box[4];
    __shared__ int stat[3];
    const int t = blockIdx.y, tile = blockIdx.x;
    if (threadIdx.x == 0) { box[0] = INT_MAX; box[1] = -1; box[2] = INT_MAX; box[3] = -1; stat[0] = stat[1] = stat[2] = 0; }
    __syncthreads();
    const bool pex = pair_exists(g, t);
    const int tc = fdiv(tile, g.dv_tiles_r), tr = tile - tc * g.tiles_r;
    const float *vx = FR(a.vx, t), *vy = FR(a.vy, t);
    if (pex) {
        for (int k = threadIdx.x; k < TILE_R * TILE_C; k += MMC_THREADS) {
            const int r = tr * TILE_R + k % TILE_R, c = tc * TILE_C + k / TILE_R;
            if (r >= g.Ny || c >= g.Nx) continue;
            const int j = r + g.Ny * c;
            Taps tp = make_taps(g, r, c, vx[j], vy[j]);
            if (!tp.valid) continue;
            // |displacement| < Ny, Nx and at most 1024 pixels per tile: the sums fit 32 bits
            atomicAdd(&stat[0], tp.r0 - r); atomicAdd(&stat[1], tp.c0 - c); atomicAdd(&stat[2], 1);
        }
    }
    __syncthreads();
    if (pex && stat[2] > 0) {
        const int dr = (int)floorf((float)stat[0] / (float)stat[2]), dc = (int)floorf((float)stat[1] / (float)stat[2]);
        for (int k = threadIdx.x; k < TILE_R * TILE_C; k += MMC_THREADS) {
            const int r = tr * TILE_R + k % TILE_R, c = tc * TILE_C + k / TILE_R;
            if (r >= g.Ny || c >= g.Nx) continue;
            const int j = r + g.Ny * c;
            Taps tp = make_taps(g, r, c, vx[j], vy[j]);
            if (!tp.valid) continue;
            if (abs(tp.r0 - r - dr) > MMC_TAP_MR || abs(tp.c0 - c - dc) > MMC_TAP_MC) continue;
            atomicMin(&box[0], tp.r0); atomicMax(&box[1], tp.r1);
            atomicMin(&box[2], tp.c0); atomicMax(&box[3], tp.c1);
        }
    }
    __syncthreads();
    if (threadIdx.x < 4) a.tbox[((size_t)t * g.ntiles + tile) * 4 + threadIdx.x] = box[threadIdx.x];
}

// Temporal duals of masked rows are identically zero; enforce it after a flow
// swap so that the primal sweep may subtract them without re-reading the mask.
__global__ __launch_bounds__(MMC_THREADS) void k_sanitize(Geom g, Arrays a) {
    const int q = blockIdx.y;
    const int j = blockIdx.x * MMC_THREADS + threadIdx.x;
    if (j >= g.N) return;
    bool valid = false;
    if (pair_exists(g, q)) {
        const int r = j % g.Ny, c = j / g.Ny;
        valid = make_taps(g, r, c, FR(a.vx, q)[j], FR(a.vy, q)[j]).valid;
    }
    if (!valid) { FR(a.g1t, q)[j] = 0.f; FR(a.g2t, q)[j] = 0.f; }
}

__global__ __launch_bounds__(MMC_THREADS) void k_fold_occ(float *vx, const uint8_t *occ, size_t count) {
    const size_t i = (size_t)blockIdx.x * MMC_THREADS + threadIdx.x;
    if (i < count && occ[i]) vx[i] = __builtin_nanf("");
}

// sums[6] += sum |W ub| over owned pairs, sums[7] += sum |G ub| over local
// frames  (MultiframeMotionCoupling.m:431-437).  PER_BLOCK: the two sums of workgroup b go to
// out[2b], out[2b+1] instead (no atomics; k_h_final adds them up in a fixed order).
template <bool PER_BLOCK>
__global__ __launch_bounds__(MMC_THREADS) void k_h_partials(Geom g, Arrays a, double *out) {
    const int t = blockIdx.y;
    const int j = blockIdx.x * MMC_THREADS + threadIdx.x;
    float part[2] = {0.f, 0.f};
    if (j < g.N) {
        const int r = j % g.Ny, c = j / g.Ny;
        const float *ub = FR(a.ub, t);
        const float uc = ub[j];
        if (c < g.Nx - 1) part[1] += fabsf(ub[j + g.Ny] - uc);
        if (r < g.Ny - 1) part[1] += fabsf(ub[j + 1] - uc);
        if (pair_exists(g, t)) {
            Taps tp = make_taps(g, r, c, FR(a.vx, t)[j], FR(a.vy, t)[j]);
            if (tp.valid) {
                const bool isIF = pair_is_IF(g, t);
                const float *W = FR(a.ub, isIF ? t + 1 : t), *I = FR(a.ub, isIF ? t : t + 1);
                part[0] += fabsf(sample(tp, W) - I[j]);
            }
        }
    }
    if (PER_BLOCK) {
        const float s = block_sum<2>(part);
        if (threadIdx.x < 2) out[2 * ((size_t)blockIdx.x + (size_t)gridDim.x * blockIdx.y) + threadIdx.x] = (double)s;
    } else {
        block_accumulate<2>(part, a.sums, SUM_H);
    }
}

// sums[SUM_DUALLIN] += <f, p> over the local frames, plus <u0, q1> + <w0, q2> on the block that holds
// the pinned first frame: the linear parts of F* (MultiframeMotionCoupling.m:496, :563-564).
__global__ __launch_bounds__(MMC_THREADS) void k_dual_linear(Geom g, Arrays a) {
    const size_t i = (size_t)blockIdx.x * MMC_THREADS + threadIdx.x;
    const size_t np = (size_t)g.L * g.n;
    float part[1] = {0.f};
    if (i < np) part[0] = a.f[i] * a.p[i];
    else if (i < np + (size_t)g.N && g.model == MODEL_IC && g.fixed && g.frame_begin == 0) {
        const size_t j = i - np;
        part[0] = a.u0[j] * a.q1[j] + a.w0[j] * a.q2[j];
    }
    block_accumulate<1>(part, a.sums, SUM_DUALLIN);
}

// One workgroup: result[i] = sum over b < nblocks of part[2b + i], i = 0, 1, in an order that
// depends on nblocks only (thread k takes b = k, k + 256, ...; then a binary tree in LDS).
__global__ __launch_bounds__(MMC_THREADS) void k_h_final(const double *part, int nblocks, double *result) {
    __shared__ double red[2][MMC_THREADS];
    double s0 = 0.0, s1 = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += MMC_THREADS) { s0 += part[2 * (size_t)b]; s1 += part[2 * (size_t)b + 1]; }
    red[0][threadIdx.x] = s0;
    red[1][threadIdx.x] = s1;
    __syncthreads();
    for (int o = MMC_THREADS / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x < 2) result[threadIdx.x] = red[threadIdx.x][0];
}

// u_up = imresize(f, zoom) with the bicubic kernel (MultiframeMotionCoupling.m:415-419): output
// sample x (1-based) sits at x/zoom + (1 - 1/zoom)/2 in input coordinates, six taps from
// floor(that - 2), Keys weights (a = -1/2) normalised to unit sum, indices beyond the edge
// replicate the edge pixel.  For an integer zoom the weights depend on x mod zoom only.
__global__ __launch_bounds__(MMC_THREADS) void k_bicubic_start(Geom g, Arrays a, BicubicTab tab) {
    const int t = blockIdx.y;
    const int j = blockIdx.x * MMC_THREADS + threadIdx.x;
    if (j >= g.N) return;
    const int c = fdiv(j, g.dv_Ny), r = j - c * g.Ny;
    const int R = r / g.f, pr = r - R * g.f, C = c / g.f, pc = c - C * g.f;
    const float *fl = a.f + (size_t)t * g.n;
    float acc = 0.f;
#pragma unroll
    for (int b = 0; b < MMC_BICUBIC_TAPS; ++b) {
        const int cc = min(max(C + tab.first[pc] + b, 0), g.nx - 1);
        float col = 0.f;
#pragma unroll
        for (int k = 0; k < MMC_BICUBIC_TAPS; ++k) {
            const int rr = min(max(R + tab.first[pr] + k, 0), g.ny - 1);
            col += tab.w[pr][k] * fl[rr + g.ny * cc];
        }
        acc += tab.w[pc][b] * col;
    }
    FR(a.u, t)[j] = acc;
    FR(a.ub, t)[j] = acc;
    if (g.model == MODEL_IC) { FR(a.w, t)[j] = acc; FR(a.wb, t)[j] = acc; }
}

// Streaming probe for the placement of the per-pixel planes (see mmc_launch_stream_probe): the
// block walk and the row-quad access pattern of the sweeps without their arithmetic, over planes in
// allocation order u, w, ub, wb, g1t, g1x, g1y, g2t, g2x, g2y, vx, vy.  The primal-like pass reads
// u, w, the six duals and the flow, reads the temporal duals of the previous frame a second time (as
// the scatter does) and writes u, w, ub, wb; the dual-like pass reads ub, wb, the duals and the flow,
// ub and wb of the next frame (as the warp taps do) and writes the duals.  Planes that are read get
// their own values back, ub and wb of the primal-like pass the value 0: the planes are all zero
// when the probe runs and stay so.
struct ProbePlanes { float *p[12]; };
template <bool DUAL>
__global__ __launch_bounds__(MMC_THREADS) void k_stream_probe(Geom g, ProbePlanes a, int nf) {
    int tile, trel;
    if (!decode_block(g, nf, tile, trel, DUAL)) return;
    const int tc = fdiv(tile, g.dv_tiles_r), tr = tile - tc * g.tiles_r;
    const int rq = threadIdx.x & (TILE_R / 4 - 1), lc = threadIdx.x / (TILE_R / 4);
    const int r = tr * TILE_R + 4 * rq, c = tc * TILE_C + lc;
    if (r >= g.Ny || c >= g.Nx) return;
    const size_t j = (size_t)(trel + 1) * (size_t)g.N + (size_t)r + (size_t)g.Ny * c;
    float4 v[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const bool rd = DUAL ? i >= 2 : (i < 2 || i >= 4);
        v[i] = rd ? *reinterpret_cast<const float4 *>(a.p[i] + j) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // second use of a neighbouring frame's planes (ghost slots exist on both sides)
    const size_t j2 = DUAL ? j + (size_t)g.N : j - (size_t)g.N;
    const float4 x0 = *reinterpret_cast<const float4 *>(a.p[DUAL ? 2 : 4] + j2);
    const float4 x1 = *reinterpret_cast<const float4 *>(a.p[DUAL ? 3 : 7] + j2);
    float s = x0.x + x0.y + x0.z + x0.w + x1.x + x1.y + x1.z + x1.w;
#pragma unroll
    for (int i = 0; i < 12; ++i) s += v[i].x + v[i].y + v[i].z + v[i].w;
    // data dependence of the stores on every load (s is finite for finite planes, so this adds exactly 0)
    const float z = (s != s) ? 1.f : 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const bool wr = DUAL ? (i >= 4 && i < 10) : i < 4;
        if (wr) {
            v[i].x += z;
            *reinterpret_cast<float4 *>(a.p[i] + j) = v[i];
        }
    }
}

// One launch of the probe over local frames [0, nf): the dual-like pass walks the block sequence
// backwards like the dual sweep, the primal-like one forwards.  Needs Ny % 4 == 0 (mmc_create probes only such blocks) and planes of at
// least nf + 2 frames.
hipError_t mmc_launch_stream_probe(const Geom &g0, float *const planes[12], int nf, bool dual_like, hipStream_t s) {
    Geom g = g0;
    g.dv_per = make_fastdiv((unsigned)((g.chunk > 0 ? g.chunk : 1) * nf));
    ProbePlanes a;
    for (int i = 0; i < 12; ++i) a.p[i] = planes[i];
    dim3 grid(8u * (unsigned)((g.ntiles + 7) / 8) * (unsigned)nf), block(MMC_THREADS);
    if (dual_like) hipLaunchKernelGGL((k_stream_probe<true>), grid, block, 0, s, g, a, nf);
    else hipLaunchKernelGGL((k_stream_probe<false>), grid, block, 0, s, g, a, nf);
    return hipGetLastError();
}

// --------------------------------------------------------------------------
// launchers
// --------------------------------------------------------------------------

// -DMMC_TUNE_LDS_PAD: env MMC_LDS_PAD = bytes of unused LDS added to every sweep launch, which lowers the
// number of workgroups a CU holds (occupancy experiments); a regular build adds nothing.
static size_t lds_pad() {
#ifdef MMC_TUNE_LDS_PAD
    static const size_t pad = getenv("MMC_LDS_PAD") ? (size_t)atol(getenv("MMC_LDS_PAD")) : 0;
    return pad;
#else
    return 0;
#endif
}

// The block decode divides by chunk * (frames of the launch); the magic number goes in with the launch.
static Geom with_launch_frames(const Geom &g0, int nf) {
    Geom g = g0;
    g.dv_per = make_fastdiv((unsigned)((g.chunk > 0 ? g.chunk : 1) * nf));
    return g;
}

template <int MODEL, int F, bool SEPK>
static void launch_primal4_m(const Geom &g0, const Arrays &a, int t0, int nf, int mode, bool reduce, hipStream_t s) {
    const Geom g = with_launch_frames(g0, nf);
    dim3 grid(8u * (unsigned)((g.ntiles + 7) / 8) * (unsigned)nf), block(MMC_THREADS);
    const size_t lds = (size_t)(6 * TILE_R * TILE_C + ((g.MP * g.NP + 3) & ~3) + ((g.PR * g.PC + 3) & ~3) +
                                (g.sep ? ((g.PR * TILE_C + 3) & ~3) + g.MP + g.NP : 0)) * sizeof(float) + lds_pad();
    if (mode == MODE_RAW) hipLaunchKernelGGL((k_primal4<MODEL, MODE_RAW, false, F, SEPK>), grid, block, lds, s, g, a, t0, nf);
    else if (mode == MODE_DUALOBJ) hipLaunchKernelGGL((k_primal4<MODEL, MODE_DUALOBJ, false, F, SEPK>), grid, block, lds, s, g, a, t0, nf);
    else if (mode == MODE_STEPS) hipLaunchKernelGGL((k_primal4<MODEL, MODE_STEPS, false, F, SEPK>), grid, block, lds, s, g, a, t0, nf);
    else if (mode == MODE_STEPS_UPD) hipLaunchKernelGGL((k_primal4<MODEL, MODE_STEPS_UPD, false, F, SEPK>), grid, block, lds, s, g, a, t0, nf);
    else if (reduce) hipLaunchKernelGGL((k_primal4<MODEL, MODE_UPDATE, true, F, SEPK>), grid, block, lds, s, g, a, t0, nf);
    else hipLaunchKernelGGL((k_primal4<MODEL, MODE_UPDATE, false, F, SEPK>), grid, block, lds, s, g, a, t0, nf);
}

template <int F>
static void launch_primal4(const Geom &g, const Arrays &a, int t0, int nf, int mode, bool reduce, hipStream_t s) {
    if (g.sep) {
        if (g.model == MODEL_IC) launch_primal4_m<MODEL_IC, F, true>(g, a, t0, nf, mode, reduce, s);
        else if (g.model == MODEL_K1) launch_primal4_m<MODEL_K1, F, true>(g, a, t0, nf, mode, reduce, s);
        else launch_primal4_m<MODEL_ADD, F, true>(g, a, t0, nf, mode, reduce, s);
        return;
    }
    if (g.model == MODEL_IC) launch_primal4_m<MODEL_IC, F, false>(g, a, t0, nf, mode, reduce, s);
    else if (g.model == MODEL_K1) launch_primal4_m<MODEL_K1, F, false>(g, a, t0, nf, mode, reduce, s);
    else launch_primal4_m<MODEL_ADD, F, false>(g, a, t0, nf, mode, reduce, s);
}

hipError_t mmc_launch_primal(const Geom &g, const Arrays &a, int t0, int t1, int mode, bool reduce, hipStream_t s) {
    if (t1 <= t0) return hipSuccess;
    if (a.net > 0 && mode != MODE_STEPS) {
        const int total = a.net * (t1 - t0);
        hipLaunchKernelGGL(k_corr_u, dim3((total + MMC_THREADS - 1) / MMC_THREADS), dim3(MMC_THREADS), 0, s, g, a, t0, t1 - t0);
    }
    if (g.f == 4) launch_primal4<4>(g, a, t0, t1 - t0, mode, reduce, s);
    else launch_primal4<0>(g, a, t0, t1 - t0, mode, reduce, s);
    return hipGetLastError();
}

template <int MODEL, int F, bool SEPK>
static void launch_dual4_m(const Geom &g0, const Arrays &a, int t0, int nf, int mode, bool reduce, hipStream_t s) {
    const Geom g = with_launch_frames(g0, nf);
    dim3 grid(8u * (unsigned)((g.ntiles + 7) / 8) * (unsigned)nf), block(MMC_THREADS);
    const int f = g.f;
    const int fe = (TILE_R % f == 0 && TILE_C % f == 0) ? f : 1;
    const int SRmax = g.SRp > TILE_R + g.CH - fe ? g.SRp : TILE_R + g.CH - fe;
    const size_t lds = (size_t)(((SRmax * (TILE_C + g.CW - fe) + 3) & ~3) + dual_mid_floats(g, SEPK) + 2 * dual_tap_floats(SEPK)) * sizeof(float) + lds_pad();
    if (mode == MODE_RAW) hipLaunchKernelGGL((k_dual4<MODEL, MODE_RAW, false, F, SEPK>), grid, block, lds, s, g, a, t0, nf);
    else if (mode == MODE_STEPS) hipLaunchKernelGGL((k_dual4<MODEL, MODE_STEPS, false, F, SEPK>), grid, block, lds, s, g, a, t0, nf);
    else if (mode == MODE_ENERGY) hipLaunchKernelGGL((k_dual4<MODEL, MODE_ENERGY, false, F, SEPK>), grid, block, lds, s, g, a, t0, nf);
    else if (reduce) hipLaunchKernelGGL((k_dual4<MODEL, MODE_UPDATE, true, F, SEPK>), grid, block, lds, s, g, a, t0, nf);
    else hipLaunchKernelGGL((k_dual4<MODEL, MODE_UPDATE, false, F, SEPK>), grid, block, lds, s, g, a, t0, nf);
}

template <int F>
static void launch_dual4(const Geom &g, const Arrays &a, int t0, int nf, int mode, bool reduce, hipStream_t s) {
    if (g.sep) {
        if (g.model == MODEL_IC) launch_dual4_m<MODEL_IC, F, true>(g, a, t0, nf, mode, reduce, s);
        else if (g.model == MODEL_K1) launch_dual4_m<MODEL_K1, F, true>(g, a, t0, nf, mode, reduce, s);
        else launch_dual4_m<MODEL_ADD, F, true>(g, a, t0, nf, mode, reduce, s);
        return;
    }
    if (g.model == MODEL_IC) launch_dual4_m<MODEL_IC, F, false>(g, a, t0, nf, mode, reduce, s);
    else if (g.model == MODEL_K1) launch_dual4_m<MODEL_K1, F, false>(g, a, t0, nf, mode, reduce, s);
    else launch_dual4_m<MODEL_ADD, F, false>(g, a, t0, nf, mode, reduce, s);
}

hipError_t mmc_launch_dual(const Geom &g, const Arrays &a, int t0, int t1, int mode, bool reduce, hipStream_t s) {
    if (t1 <= t0) return hipSuccess;
    const int nf = t1 - t0;
    if (a.ne > 0 && mode != MODE_STEPS) {
        const int total = a.ne * nf;
        const int per_block = MMC_THREADS / CORR_P_LANES;
        hipLaunchKernelGGL(k_corr_p, dim3((total + per_block - 1) / per_block), dim3(MMC_THREADS), 0, s, g, a, t0, nf);
    }
    // regulariser duals and data-term dual in one launch
    if (g.f == 4) launch_dual4<4>(g, a, t0, nf, mode, reduce, s);
    else launch_dual4<0>(g, a, t0, nf, mode, reduce, s);
    return hipGetLastError();
}

// Pairs whose warp samples a local frame (their adjoint scatters into it): pair t-1 of frame t if
// that pair is 'I-F', pair t if it is 'F-I'.
template <typename F>
static void for_each_scattering_pair(const Geom &g, F fn) {
    auto exists = [&](int q) { int gq = g.frame_begin + q; return q >= -1 && q <= g.L - 1 && gq >= 0 && gq <= g.nc_global - 2; };
    auto is_if = [&](int q) { return g.dir == 1 ? true : g.dir == 0 ? false : ((g.frame_begin + q) & 1) == 0; };
    for (int t = 0; t < g.L; ++t) {
        if (exists(t - 1) && is_if(t - 1)) fn(t, t - 1);
        if (exists(t) && !is_if(t)) fn(t, t);
    }
}

hipError_t mmc_launch_init_warp(const Geom &g, const Arrays &a, void *frame_words, hipStream_t s) {
    dim3 grid((g.N + MMC_THREADS - 1) / MMC_THREADS), block(MMC_THREADS);
    fx64 *tmp = static_cast<fx64 *>(frame_words);
    for (int t = 0; t < g.L; ++t) {
        hipError_t e = hipMemsetAsync(tmp, 0, (size_t)g.N * sizeof(fx64), s);
        if (e != hipSuccess) return e;
        for_each_scattering_pair(g, [&](int tf, int q) {
            if (tf == t) hipLaunchKernelGGL(k_init_scatter, grid, block, 0, s, g, a, q, tmp);
        });
        hipLaunchKernelGGL(k_init_finalize, grid, block, 0, s, g, a, t, tmp);
        hipLaunchKernelGGL(k_init_idq, dim3((unsigned)((MMC_IDQ_PER_FRAME(g) + MMC_THREADS - 1) / MMC_THREADS)), block, 0, s, g, a, t);
    }
    for_each_scattering_pair(g, [&](int, int q) { hipLaunchKernelGGL(k_init_srcbox, grid, block, 0, s, g, a, q); });
    hipLaunchKernelGGL(k_init_tapbox, dim3(g.ntiles, g.L), block, 0, s, g, a);
    return hipGetLastError();
}

hipError_t mmc_launch_init_overflow(const Geom &g, const Arrays &a, bool fill, int *cursor, hipStream_t s) {
    dim3 grid((g.N + MMC_THREADS - 1) / MMC_THREADS), block(MMC_THREADS);
    for_each_scattering_pair(g, [&](int, int q) {
        if (fill) hipLaunchKernelGGL(k_init_overflow<true>, grid, block, 0, s, g, a, q, cursor);
        else hipLaunchKernelGGL(k_init_overflow<false>, grid, block, 0, s, g, a, q, cursor);
    });
    return hipGetLastError();
}

hipError_t mmc_launch_sanitize(const Geom &g, const Arrays &a, hipStream_t s) {
    dim3 grid((g.N + MMC_THREADS - 1) / MMC_THREADS, g.L), block(MMC_THREADS);
    hipLaunchKernelGGL(k_sanitize, grid, block, 0, s, g, a);
    return hipGetLastError();
}

hipError_t mmc_launch_fold_occ(const Geom &g, float *vx, const uint8_t *occ, size_t count, hipStream_t s) {
    (void)g;
    if (count == 0) return hipSuccess;
    dim3 grid((unsigned)((count + MMC_THREADS - 1) / MMC_THREADS)), block(MMC_THREADS);
    hipLaunchKernelGGL(k_fold_occ, grid, block, 0, s, vx, occ, count);
    return hipGetLastError();
}

hipError_t mmc_launch_bicubic_start(const Geom &g, const Arrays &a, const BicubicTab &tab, hipStream_t s) {
    dim3 grid((g.N + MMC_THREADS - 1) / MMC_THREADS, g.L), block(MMC_THREADS);
    hipLaunchKernelGGL(k_bicubic_start, grid, block, 0, s, g, a, tab);
    return hipGetLastError();
}

hipError_t mmc_launch_dual_linear(const Geom &g, const Arrays &a, hipStream_t s) {
    const size_t total = (size_t)g.L * g.n + ((g.model == MODEL_IC && g.fixed && g.frame_begin == 0) ? (size_t)g.N : 0);
    hipLaunchKernelGGL(k_dual_linear, dim3((unsigned)((total + MMC_THREADS - 1) / MMC_THREADS)), dim3(MMC_THREADS), 0, s, g, a);
    return hipGetLastError();
}

hipError_t mmc_launch_h_partials(const Geom &g, const Arrays &a, hipStream_t s) {
    dim3 grid((g.N + MMC_THREADS - 1) / MMC_THREADS, g.L), block(MMC_THREADS);
    hipLaunchKernelGGL(k_h_partials<false>, grid, block, 0, s, g, a, (double *)nullptr);
    return hipGetLastError();
}

// --------------------------------------------------------------------------
// h sums without a handle and without any allocation
// --------------------------------------------------------------------------
// Per-workgroup partial sums of mmc_h_partials_static live in this file-scope device array (every
// GPU the library runs on holds its own copy until the process ends): two doubles per workgroup
// of k_h_partials.  Host code never uses the array's name as an address; it asks the runtime for
// the address on each device once (static_scratch_ptr) and passes that to the kernels.
static __device__ __align__(256) double scratch[2 * (size_t)MMC_STATIC_MAX_BLOCKS];

enum { MMC_STATIC_MAX_DEVICES = 64 };
static std::mutex g_scratch_mutex;
static double *g_scratch_ptr[MMC_STATIC_MAX_DEVICES];      // per device; null until first use there

// Address of `scratch` on `device`, which must be the current device.
static hipError_t static_scratch_ptr(int device, double **out) {
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void *p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        g_scratch_ptr[device] = static_cast<double *>(p);
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}

extern "C" int mmc_h_partials_static(const float *d_u, const float *d_v, int Ny, int Nx, int nc,
                                     int flow_direction, int device, void *stream, double *d_out) {
    // everything that can be rejected is rejected before the first HIP call
    if (!d_u || !d_out) return mmc_static_fail(MMC_ERR_INVALID_ARG, "mmc_h_partials_static: null frames or null output");
    if (Ny <= 0 || Nx <= 0 || nc <= 0) return mmc_static_fail(MMC_ERR_INVALID_ARG, "mmc_h_partials_static: empty sequence");
    if (nc > 1 && !d_v) return mmc_static_fail(MMC_ERR_INVALID_ARG, "mmc_h_partials_static: null flow");
    if (flow_direction < 0 || flow_direction > 2) return mmc_static_fail(MMC_ERR_INVALID_ARG, "Invalid flow direction.");
    if ((long long)Ny * Nx > (long long)INT_MAX / 4)
        return mmc_static_fail(MMC_ERR_INVALID_ARG, "mmc_h_partials_static: frame too large for 32-bit pixel indexing");
    if (device < 0 || device >= MMC_STATIC_MAX_DEVICES)
        return mmc_static_fail(MMC_ERR_INVALID_ARG, "mmc_h_partials_static: device ordinal out of range");
    const long long per_frame = ((long long)Ny * Nx + MMC_THREADS - 1) / MMC_THREADS;
    if (per_frame * nc > (long long)MMC_STATIC_MAX_BLOCKS || nc > 65535)
        return mmc_static_fail(MMC_ERR_INVALID_ARG,
                               "mmc_h_partials_static: sequence exceeds the static workspace "
                               "(ceil(Ny*Nx/256)*nc must not exceed MMC_STATIC_MAX_BLOCKS = 524288); use mmc_h_partials on a handle");
    // The caller's current device is put back before the call returns (DeviceGuard below).
    struct DeviceGuard {
        int prev = -1;
        ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    } guard;
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) return mmc_static_fail(MMC_ERR_HIP, "mmc_h_partials_static: hipGetDevice failed");
    if (cur != device) {
        if (hipSetDevice(device) != hipSuccess) return mmc_static_fail(MMC_ERR_HIP, "mmc_h_partials_static: hipSetDevice failed");
        guard.prev = cur;
    }
    (void)hipGetLastError();      // an error left behind by an unrelated earlier call must not be reported as ours
    double *part = nullptr;
    if (static_scratch_ptr(device, &part) != hipSuccess)
        return mmc_static_fail(MMC_ERR_HIP, "mmc_h_partials_static: hipGetSymbolAddress failed");

    Geom g;
    memset(&g, 0, sizeof(g));
    g.Ny = Ny; g.Nx = Nx; g.N = Ny * Nx;
    g.L = nc; g.frame_begin = 0; g.nc_global = nc;
    g.dir = flow_direction;
    Arrays a;
    memset(&a, 0, sizeof(a));
    // the kernels address frame t of a plane as base + (t+1)*N (slot 0 is a ghost frame that a
    // whole sequence never touches), so the bases sit one frame before the caller's buffers
    const size_t N = (size_t)g.N;
    a.ub = const_cast<float *>(d_u) - N;
    if (nc > 1) {
        a.vx = const_cast<float *>(d_v) - N;
        a.vy = const_cast<float *>(d_v) + (size_t)(nc - 1) * N - N;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    dim3 grid((unsigned)per_frame, (unsigned)nc), block(MMC_THREADS);
    // every slot the final sum reads is written by the first kernel: nothing has to start at zero
    hipLaunchKernelGGL(k_h_partials<true>, grid, block, 0, s, g, a, part);
    hipLaunchKernelGGL(k_h_final, dim3(1), block, 0, s, part, (int)(per_frame * nc), d_out);
    if (hipGetLastError() != hipSuccess) return mmc_static_fail(MMC_ERR_HIP, "mmc_h_partials_static: kernel launch failed");
    return MMC_OK;
}
