// gfx950 kernels of the classical single-frame super-resolution solve
// (classical methods/singleframeMotionSR_mitzel_prost.m:75-163, ..._unger_prost.m):
//
//     min_u  sum_i H_delta(D B W_i u - f_i) + alpha TV(u)
//
// one high-resolution frame u, one data term per low-resolution frame i, W_i the bilinear warp from
// the central frame to frame i.  Saddle-point form with duals p_i (n each) and q (2N):
//     K = [A W_1; ...; A W_nf; G],   A = D B
// One PDHG iteration:
//   primal  u+ = u - T (sum_i W_i' A' p_i + G' q),   ub = 2u+ - u
//   dual    p_i <- clamp((p_i + S (A W_i ub - f_i)) / (1 + S delta), -1, 1)
//           q   <- projection of q + S G ub onto the alpha-balls
// Five simple kernels per iteration.  This is the comparison method of the reference, built for
// parity and completeness; it is not tuned like the sweeps of mmc_kernels.hip:
//   k_sf_warp     W_i ub for all frames into a scratch plane per frame (4-tap gather)
//   k_sf_dual_p   A (W_i ub) per low-resolution pixel from that plane, p update
//   k_sf_dual_q   gradient of ub, q update
//   k_sf_scatter  W_i' (A' p_i): per source pixel the gather form of A' p_i, then four 64-bit
//                 fixed-point global atomics (order-independent, so runs are bitwise reproducible)
//   k_sf_primal   u update from the accumulator (cleared on the way) and G' q
// The same kernels applied to constant input give the diagonal step sizes (the operator has no
// negative entry, which the host checks, so absolute row and column sums are K 1 and K' 1).
#include <hip/hip_runtime.h>
#include <limits.h>
#include "../../include/mmc_hip.h"
#include "mmc_internal.h"

#include "mmc_device.h"

// W_i img for every frame i (MODE_RAW / MODE_UPDATE), or the row mask of W_i (MODE_STEPS)
template <int MODE>
__global__ __launch_bounds__(MMC_THREADS) void k_sf_warp(Geom g, SfArrays a, const float *__restrict__ img) {
    const size_t idx = (size_t)blockIdx.x * MMC_THREADS + threadIdx.x;
    if (idx >= (size_t)g.N * a.nf) return;
    const int i = (int)(idx / g.N), j = (int)(idx - (size_t)i * g.N);
    const int c = fdiv(j, g.dv_Ny), r = j - c * g.Ny;
    const Taps tp = make_taps(g, r, c, a.vx[idx], a.vy[idx]);
    float out = 0.f;
    if (tp.valid) out = MODE == MODE_STEPS ? 1.f : sample(tp, img);
    a.wimg[idx] = out;
}

// (A wimg_i)(R,C) with the replicate border of helpers/RepConvMtx.m:23-32, then the update of p_i
template <int MODE, bool REDUCE>
__global__ __launch_bounds__(MMC_THREADS) void k_sf_dual_p(Geom g, SfArrays a) {
    const size_t idx = (size_t)blockIdx.x * MMC_THREADS + threadIdx.x;
    float part[2] = {0.f, 0.f};
    if (idx < (size_t)g.n * a.nf) {
        const int i = (int)(idx / g.n), l = (int)(idx - (size_t)i * g.n);
        const int R = l % g.ny, C = l / g.ny;
        const float *w = a.wimg + (size_t)i * g.N;
        float acc = 0.f;
        for (int nn = 0; nn < g.CW; ++nn) {
            const int cc = min(max(g.f * C + g.ox + nn, 0), g.Nx - 1);
            const float *col = w + (size_t)g.Ny * cc;
            for (int mm = 0; mm < g.CH; ++mm) {
                const int rr = min(max(g.f * R + g.oy + mm, 0), g.Ny - 1);
                acc += a.cker[mm + g.CH * nn] * col[rr];
            }
        }
        if (MODE == MODE_RAW) {
            a.p[idx] = acc;
        } else if (MODE == MODE_STEPS) {
            a.sigp[idx] = acc > 0.f ? 1.f / acc : 1.f;
        } else {
            const float sg = a.sigp[idx], po = a.p[idx];
            // ind{|p|<=1} + <f,p> + delta/2 |p|^2  (sum_1d('ind_box01',0.5,-0.5,1,f,delta), :148-161)
            const float pn = fminf(fmaxf((po + sg * (acc - a.f[idx])) / (1.f + sg * a.delta), -1.f), 1.f);
            a.p[idx] = pn;
            if (REDUCE) { part[0] = (pn - po) * (pn - po); part[1] = pn * pn; }
        }
    }
    if (REDUCE) block_accumulate<2>(part, a.sums, 2);
}

// The two kernels above in one, per tile of SF_TL x SF_TL low-resolution pixels of one frame: the
// window of W_i ub that the tile's rows of A touch (tile * zoom + kernel halo, replicate border applied
// while staging) is computed straight into LDS, the composite kernel is applied from there.  Saves the
// scratch plane's write and read; the flow is read once per window pixel (halo: 1.2x at zoom 4).
enum { SF_TL = 16, SF_WIN_MAX = 12032 };        // window floats that fit next to the kernel table in 48 KB of LDS
template <int MODE, bool REDUCE>
__global__ __launch_bounds__(MMC_THREADS) void k_sf_dual_tile(Geom g, SfArrays a, const float *__restrict__ img, int tiles_R) {
    extern __shared__ __align__(16) float lds[];
    const int SR = SF_TL * g.f + g.CH - g.f, SC = SF_TL * g.f + g.CW - g.f;
    float *win = lds, *ck = lds + SR * SC;
    const int i = blockIdx.y;
    const int tC = blockIdx.x / tiles_R, tR = blockIdx.x - tC * tiles_R;
    const int R0 = SF_TL * tR, C0 = SF_TL * tC;
    const int wr0 = g.f * R0 + g.oy, wc0 = g.f * C0 + g.ox;
    const float *vx = a.vx + (size_t)i * g.N, *vy = a.vy + (size_t)i * g.N;
    for (int k = threadIdx.x; k < g.CH * g.CW; k += MMC_THREADS) ck[k] = a.cker[k];
    for (int k = threadIdx.x; k < SR * SC; k += MMC_THREADS) {
        const int kc = k / SR, kr = k - kc * SR;
        const int r = min(max(wr0 + kr, 0), g.Ny - 1), c = min(max(wc0 + kc, 0), g.Nx - 1);
        const int j = r + g.Ny * c;
        const Taps tp = make_taps(g, r, c, vx[j], vy[j]);
        float out = 0.f;
        if (tp.valid) out = MODE == MODE_STEPS ? 1.f : sample(tp, img);
        win[k] = out;
    }
    __syncthreads();
    float part[2] = {0.f, 0.f};
    for (int l = threadIdx.x; l < SF_TL * SF_TL; l += MMC_THREADS) {
        const int Cl = l / SF_TL, Rl = l - Cl * SF_TL;
        const int R = R0 + Rl, C = C0 + Cl;
        if (R >= g.ny || C >= g.nx) continue;
        float acc = 0.f;
        for (int nn = 0; nn < g.CW; ++nn) {
            const float *col = win + g.f * Rl + SR * (g.f * Cl + nn);
            for (int mm = 0; mm < g.CH; ++mm) acc += ck[mm + g.CH * nn] * col[mm];
        }
        const size_t idx = (size_t)i * g.n + R + (size_t)g.ny * C;
        if (MODE == MODE_RAW) {
            a.p[idx] = acc;
        } else if (MODE == MODE_STEPS) {
            a.sigp[idx] = acc > 0.f ? 1.f / acc : 1.f;
        } else {
            const float sg = a.sigp[idx], po = a.p[idx];
            const float pn = fminf(fmaxf((po + sg * (acc - a.f[idx])) / (1.f + sg * a.delta), -1.f), 1.f);
            a.p[idx] = pn;
            if (REDUCE) { part[0] += (pn - po) * (pn - po); part[1] += pn * pn; }
        }
    }
    if (REDUCE) block_accumulate<2>(part, a.sums, 2);
}

// q <- projection of q + s G ub onto ||q_j|| <= alpha  (sum_norm2(2,false,'ind_leq0',1/alpha,...), :163)
template <int MODE, bool REDUCE>
__global__ __launch_bounds__(MMC_THREADS) void k_sf_dual_q(Geom g, SfArrays a, const float *__restrict__ img) {
    const int j = blockIdx.x * MMC_THREADS + threadIdx.x;
    float part[2] = {0.f, 0.f};
    if (j < g.N) {
        const int c = fdiv(j, g.dv_Ny), r = j - c * g.Ny;
        const bool hasx = c < g.Nx - 1, hasy = r < g.Ny - 1;
        const float v = img[j];
        const float dx = hasx ? img[j + g.Ny] - v : 0.f;
        const float dy = hasy ? img[j + 1] - v : 0.f;
        if (MODE == MODE_RAW) {
            a.qx[j] = dx; a.qy[j] = dy;
        } else {
            // one step per isotropic group: 1 / (largest absolute row sum of the group)
            const float rs = fmaxf(hasx ? 2.f : 0.f, hasy ? 2.f : 0.f);
            const float sg = rs > 0.f ? 1.f / rs : 1.f;
            if (MODE == MODE_STEPS) {
                a.qx[j] = sg; a.qy[j] = sg;
            } else {
                const float ox = a.qx[j], oy = a.qy[j];
                float nx = ox + sg * dx, ny = oy + sg * dy;
                const float n2 = nx * nx + ny * ny;
                if (n2 > a.alpha * a.alpha) { const float sc = a.alpha / sqrtf(n2); nx *= sc; ny *= sc; }
                a.qx[j] = nx; a.qy[j] = ny;
                if (REDUCE) { part[0] = (nx - ox) * (nx - ox) + (ny - oy) * (ny - oy); part[1] = nx * nx + ny * ny; }
            }
        }
    }
    if (REDUCE) block_accumulate<2>(part, a.sums, 2);
}

// acc += W_i' z_i, z_i = A' p_i (MODE_UPDATE / MODE_RAW) or the column sums of A (MODE_STEPS).
// The lanes of a wave are consecutive rows of one column; where the flow is smooth, the lower tap
// row of lane k is the upper tap row of lane k+1.  Lane k then adds its neighbour's upper
// contributions to its own lower ones before the atomics (integer sums: the result does not depend on
// who adds), which halves the number of atomics.
template <int MODE>
__global__ __launch_bounds__(MMC_THREADS) void k_sf_scatter(Geom g, SfArrays a) {
    const size_t idx = (size_t)blockIdx.x * MMC_THREADS + threadIdx.x;
    const bool inside = idx < (size_t)g.N * a.nf;
    int i00 = -1, i10 = -1, i01 = -1, i11 = -1;
    fx64 f00 = 0, f10 = 0, f01 = 0, f11 = 0;
    if (inside) {
        const int i = (int)(idx / g.N), j = (int)(idx - (size_t)i * g.N);
        const int c = fdiv(j, g.dv_Ny), r = j - c * g.Ny;
        const Taps tp = make_taps(g, r, c, a.vx[idx], a.vy[idx]);
        if (tp.valid) {
            const float z = MODE == MODE_STEPS ? a.cA[j] : gather_ATp(g, a.cker, a.p + (size_t)i * g.n, r, c);
            if (z != 0.f) {
                const double s40 = fx_scaled(z);
                i00 = tp.i00; i10 = tp.i10; i01 = tp.i01; i11 = tp.i11;
                f00 = to_fx((double)tp.w00, s40); f10 = to_fx((double)tp.w10, s40);
                f01 = to_fx((double)tp.w01, s40); f11 = to_fx((double)tp.w11, s40);
            }
        }
    }
    const int lane = threadIdx.x & 63;
    const int n00 = __shfl_down(i00, 1, 64), n01 = __shfl_down(i01, 1, 64);
    const fx64 nf00 = __shfl_down(f00, 1, 64), nf01 = __shfl_down(f01, 1, 64);
    const int p10 = __shfl_up(i10, 1, 64), p11 = __shfl_up(i11, 1, 64);
    if (lane < 63 && i10 >= 0 && n00 == i10) f10 += nf00;
    if (lane < 63 && i11 >= 0 && n01 == i11) f11 += nf01;
    const bool given0 = lane > 0 && i00 >= 0 && p10 == i00;      // the previous lane has taken these
    const bool given1 = lane > 0 && i01 >= 0 && p11 == i01;
    if (i00 < 0) return;
    if (!given0 && f00 != 0) atomicAdd(a.acc + i00, f00);
    if (f10 != 0) atomicAdd(a.acc + i10, f10);
    if (!given1 && f01 != 0) atomicAdd(a.acc + i01, f01);
    if (f11 != 0) atomicAdd(a.acc + i11, f11);
}

// u+ = u - T (acc + G' q), ub = 2u+ - u; the accumulator is left at zero for the next iteration
template <int MODE, bool REDUCE>
__global__ __launch_bounds__(MMC_THREADS) void k_sf_primal(Geom g, SfArrays a) {
    const int j = blockIdx.x * MMC_THREADS + threadIdx.x;
    float part[2] = {0.f, 0.f};
    if (j < g.N) {
        const int c = fdiv(j, g.dv_Ny), r = j - c * g.Ny;
        const float s = from_fx(a.acc[j]);
        a.acc[j] = 0;
        if (MODE == MODE_STEPS) {
            // absolute column sum: data rows (all entries >= 0) + the gradient's (up to four entries of size 1)
            const float col = s + (c > 0 ? 1.f : 0.f) + (c < g.Nx - 1 ? 1.f : 0.f) + (r > 0 ? 1.f : 0.f) + (r < g.Ny - 1 ? 1.f : 0.f);
            a.tau[j] = col > 0.f ? 1.f / col : 1.f;
        } else {
            const float kt = s + grad_adj(g, a.qx, a.qy, r, c, j);
            if (MODE == MODE_RAW) {
                a.ub[j] = kt;
            } else {
                const float uo = a.u[j];
                const float un = uo - a.tau[j] * kt;          // no constraint on u (prost.variable(N), :97)
                a.u[j] = un;
                a.ub[j] = 2.f * un - uo;
                if (REDUCE) { part[0] = (un - uo) * (un - uo); part[1] = un * un; }
            }
        }
    }
    if (REDUCE) block_accumulate<2>(part, a.sums, 0);
}

// --------------------------------------------------------------------------
// launchers
// --------------------------------------------------------------------------

static dim3 grid_for(size_t count) { return dim3((unsigned)((count + MMC_THREADS - 1) / MMC_THREADS)); }

template <int MODE, bool REDUCE>
static void launch_dual_mode(const Geom &g, const SfArrays &a, hipStream_t s) {
    const dim3 block(MMC_THREADS);
    const int SR = SF_TL * g.f + g.CH - g.f, SC = SF_TL * g.f + g.CW - g.f;
    if (SR * SC <= SF_WIN_MAX) {
        const int tiles_R = (g.ny + SF_TL - 1) / SF_TL, tiles_C = (g.nx + SF_TL - 1) / SF_TL;
        const size_t lds = (size_t)(SR * SC + g.CH * g.CW) * sizeof(float);
        hipLaunchKernelGGL((k_sf_dual_tile<MODE, REDUCE>), dim3(tiles_R * tiles_C, a.nf), block, lds, s, g, a, a.ub, tiles_R);
    } else {
        hipLaunchKernelGGL(k_sf_warp<MODE>, grid_for((size_t)g.N * a.nf), block, 0, s, g, a, a.ub);
        hipLaunchKernelGGL((k_sf_dual_p<MODE, REDUCE>), grid_for((size_t)g.n * a.nf), block, 0, s, g, a);
    }
    hipLaunchKernelGGL((k_sf_dual_q<MODE, REDUCE>), grid_for((size_t)g.N), block, 0, s, g, a, a.ub);
}

hipError_t mmc_sf_launch_dual(const Geom &g, const SfArrays &a, int mode, bool reduce, hipStream_t s) {
    if (mode == MODE_STEPS) launch_dual_mode<MODE_STEPS, false>(g, a, s);
    else if (mode == MODE_RAW) launch_dual_mode<MODE_RAW, false>(g, a, s);
    else if (reduce) launch_dual_mode<MODE_UPDATE, true>(g, a, s);
    else launch_dual_mode<MODE_UPDATE, false>(g, a, s);
    return hipGetLastError();
}

hipError_t mmc_sf_launch_primal(const Geom &g, const SfArrays &a, int mode, bool reduce, hipStream_t s) {
    const dim3 block(MMC_THREADS);
    const size_t npix = (size_t)g.N * a.nf;
    if (mode == MODE_STEPS) {
        hipLaunchKernelGGL(k_sf_scatter<MODE_STEPS>, grid_for(npix), block, 0, s, g, a);
        hipLaunchKernelGGL((k_sf_primal<MODE_STEPS, false>), grid_for((size_t)g.N), block, 0, s, g, a);
    } else if (mode == MODE_RAW) {
        hipLaunchKernelGGL(k_sf_scatter<MODE_RAW>, grid_for(npix), block, 0, s, g, a);
        hipLaunchKernelGGL((k_sf_primal<MODE_RAW, false>), grid_for((size_t)g.N), block, 0, s, g, a);
    } else {
        hipLaunchKernelGGL(k_sf_scatter<MODE_UPDATE>, grid_for(npix), block, 0, s, g, a);
        if (reduce) hipLaunchKernelGGL((k_sf_primal<MODE_UPDATE, true>), grid_for((size_t)g.N), block, 0, s, g, a);
        else hipLaunchKernelGGL((k_sf_primal<MODE_UPDATE, false>), grid_for((size_t)g.N), block, 0, s, g, a);
    }
    return hipGetLastError();
}
