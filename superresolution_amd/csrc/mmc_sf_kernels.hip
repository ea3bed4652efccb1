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
// Three kernels per iteration (this is the comparison method of the reference; it follows the design
// of the sweeps in mmc_kernels.hip without their tuning):
//   k_sf_dual_tile    per tile of 16 x 16 low-resolution pixels of a frame: the window of W_i ub that
//                     the tile's rows of A touch goes straight into LDS, A from there, p update
//                     (k_sf_warp + k_sf_dual_p through a scratch plane when the window does not fit)
//   k_sf_dual_q       gradient of ub, q update
//   k_sf_atp          A' p_i at every pixel of every frame, gather form, into the scratch plane
//   k_sf_scatter_tile per 64 x 16 target tile and group of frames: walk the rectangle of sources with
//                     a tap in the tile (k_sf_bbox at set-up), tap contributions added in LDS as 64-bit
//                     fixed point, one global atomic per pixel and group into a.acc.  Integer sums are
//                     order-independent, so runs are bitwise reproducible.
//   k_sf_scatter      frames whose flow is too violent for rectangles (a.irregular) go through
//                     per-tap 64-bit global atomics into a.acc instead
//   k_sf_primal       G' q and the u update from a.acc (cleared on the way)
// The same kernels applied to constant input give the diagonal step sizes (the operator has no
// negative entry, which the host checks, so absolute row and column sums are K 1 and K' 1).
#include <hip/hip_runtime.h>
#include <limits.h>
#include "../../include/mmc_hip.h"
#include "mmc_internal.h"

#include "mmc_device.h"

// k / d and k % d for 0 <= k < 2^23 and a workgroup-uniform d through one float multiply (inv = 1/d)
__device__ __forceinline__ void divmod_small(int k, int d, float inv, int &q, int &rem) {
    q = (int)(((float)k + 0.5f) * inv);
    rem = k - q * d;
    if (rem < 0) { --q; rem += d; }
    if (rem >= d) { ++q; rem -= d; }
}

// W_i img for every frame i (MODE_RAW / MODE_UPDATE), or the row mask of W_i (MODE_STEPS)
template <int MODE>
__global__ __launch_bounds__(MMC_THREADS) void k_sf_warp(Geom g, SfArrays a, const float *__restrict__ img) {
    const int i = blockIdx.y, j = blockIdx.x * MMC_THREADS + threadIdx.x;
    if (j >= g.N) return;
    const size_t idx = (size_t)i * g.N + j;
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
    const float inv_sr = 1.f / (float)SR;
    // four window pixels per thread and pass: their flow loads are issued together, then their gathers
    for (int kb = threadIdx.x; kb < SR * SC; kb += 4 * MMC_THREADS) {
        int rr[4], cc[4];
        float fx[4], fy[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = min(kb + u * MMC_THREADS, SR * SC - 1);
            int kc, kr;
            divmod_small(k, SR, inv_sr, kc, kr);
            rr[u] = min(max(wr0 + kr, 0), g.Ny - 1);
            cc[u] = min(max(wc0 + kc, 0), g.Nx - 1);
            const int j = rr[u] + g.Ny * cc[u];
            fx[u] = vx[j]; fy[u] = vy[j];
        }
        float out[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const Taps tp = make_taps(g, rr[u], cc[u], fx[u], fy[u]);       // invalid taps point at pixel 0
            const float v = MODE == MODE_STEPS ? 1.f : sample(tp, img);
            out[u] = tp.valid ? v : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = kb + u * MMC_THREADS;
            if (k < SR * SC) win[k] = out[u];
        }
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

// q <- projection of (q + s G ub) / (1 + s delta/alpha) onto ||q_j|| <= alpha: the prox of
// ind{||q_j|| <= alpha} + (delta/alpha)/2 ||q_j||^2, i.e. sum_norm2(2,false,'ind_leq0',1/alpha,1,1,0,huber_eps/alpha)
// of singleframeMotionSR_unger_prost.m:163 (Huber-smoothed TV); delta = 0 is the plain projection of
// singleframeMotionSR_mitzel_prost.m:163.
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
            const float ox = a.qx[j], oy = a.qy[j];
            const float shrink = 1.f / (1.f + sg * a.delta / a.alpha);
            float nx = (ox + sg * dx) * shrink, ny = (oy + sg * dy) * shrink;
            const float n2 = nx * nx + ny * ny;
            if (n2 > a.alpha * a.alpha) { const float sc = a.alpha / sqrtf(n2); nx *= sc; ny *= sc; }
            a.qx[j] = nx; a.qy[j] = ny;
            if (REDUCE) { part[0] = (nx - ox) * (nx - ox) + (ny - oy) * (ny - oy); part[1] = nx * nx + ny * ny; }
        }
    }
    if (REDUCE) block_accumulate<2>(part, a.sums, 2);
}

// z_i(r,c) = (A' p_i)(r,c), or the column sum of A there (MODE_STEPS).  Zoom 4 away from the frame
// border: with x = r - oy the low-resolution rows that have a kernel row on r are R = (x >> 2) - jr with
// kernel row (x & 3) + 4 jr, jr < Jr; a fixed Jr x Jc loop with masked terms, so that the loads of
// several sources can be in flight together.  Border pixels take the general gather (gather_ATp), which
// handles the virtual rows and columns that the replicate border folds onto the frame's edge.
template <int MODE>
__device__ __forceinline__ float sf_z(const Geom &g, const SfArrays &a, int i, int r, int c, int j) {
    if (MODE == MODE_STEPS) return a.cA[j];
    const float *p = a.p + (size_t)i * g.n;
    if (g.f == 4 && r > 0 && r < g.Ny - 1 && c > 0 && c < g.Nx - 1) {
        const int x = r - g.oy, y = c - g.ox;
        const int R0 = x >> 2, C0 = y >> 2, m0 = x & 3, n0 = y & 3;
        float acc = 0.f;
        for (int jc = 0; jc < g.Jc; ++jc) {
            const int C = C0 - jc, nn = n0 + 4 * jc;
            const bool okc = C >= 0 && C < g.nx && nn < g.CW;
            const float *ck = a.cker + g.CH * min(nn, g.CW - 1);
            const float *pc = p + g.ny * min(max(C, 0), g.nx - 1);
            for (int jr = 0; jr < g.Jr; ++jr) {
                const int R = R0 - jr, mm = m0 + 4 * jr;
                const bool ok = okc && R >= 0 && R < g.ny && mm < g.CH;
                const float t = ck[min(mm, g.CH - 1)] * pc[min(max(R, 0), g.ny - 1)];
                acc += ok ? t : 0.f;
            }
        }
        return acc;
    }
    return gather_ATp(g, a.cker, p, r, c);
}

// acc += W_i' z_i, z_i = A' p_i (MODE_UPDATE / MODE_RAW) or the column sums of A (MODE_STEPS).
// The lanes of a wave are consecutive rows of one column; where the flow is smooth, the lower tap
// row of lane k is the upper tap row of lane k+1.  Lane k then adds its neighbour's upper
// contributions to its own lower ones before the atomics (integer sums: the result does not depend on
// who adds), which halves the number of atomics.
template <int MODE>
__global__ __launch_bounds__(MMC_THREADS) void k_sf_scatter(Geom g, SfArrays a) {
    const int i = blockIdx.y, j = blockIdx.x * MMC_THREADS + threadIdx.x;
    const bool inside = j < g.N;
    const size_t idx = (size_t)i * g.N + (inside ? j : 0);
    int i00 = -1, i10 = -1, i01 = -1, i11 = -1;
    fx64 f00 = 0, f10 = 0, f01 = 0, f11 = 0;
    if (inside) {
        const int c = fdiv(j, g.dv_Ny), r = j - c * g.Ny;
        Taps tp;
        tp.valid = false;
        if (a.irregular[i]) tp = make_taps(g, r, c, a.vx[idx], a.vy[idx]);     // the other frames are scattered per tile
        if (tp.valid) {
            const float z = sf_z<MODE>(g, a, i, r, c, j);
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

// Source rectangles: for every frame and every 64 x 16 tile the bounding box of the source pixels
// with at least one tap inside the tile.
__global__ __launch_bounds__(MMC_THREADS) void k_sf_bbox_clear(int *bbox, int count) {
    const int k = blockIdx.x * MMC_THREADS + threadIdx.x;
    if (k < count) bbox[k] = (k & 1) ? -1 : INT_MAX;            // rmin, rmax, cmin, cmax
}

__global__ __launch_bounds__(MMC_THREADS) void k_sf_bbox(Geom g, SfArrays a) {
    const int i = blockIdx.y, j = blockIdx.x * MMC_THREADS + threadIdx.x;
    if (j >= g.N) return;
    const size_t idx = (size_t)i * g.N + j;
    const int c = fdiv(j, g.dv_Ny), r = j - c * g.Ny;
    const Taps tp = make_taps(g, r, c, a.vx[idx], a.vy[idx]);
    if (!tp.valid) return;
    const int tr0 = tp.r0 / TILE_R, tr1 = tp.r1 / TILE_R, tc0 = tp.c0 / TILE_C, tc1 = tp.c1 / TILE_C;
    int *base = a.bbox + (size_t)i * g.ntiles * 4;
    for (int kc = 0; kc < 2; ++kc) {
        if (kc == 1 && tc1 == tc0) break;
        for (int kr = 0; kr < 2; ++kr) {
            if (kr == 1 && tr1 == tr0) break;
            int *bb = base + ((kr ? tr1 : tr0) + g.tiles_r * (kc ? tc1 : tc0)) * 4;
            atomicMin(bb + 0, r); atomicMax(bb + 1, r); atomicMin(bb + 2, c); atomicMax(bb + 3, c);
        }
    }
}

// z_i at every pixel of every regular frame into the scratch plane: fully parallel, so that the
// tile scatter below reads one float per source.
template <int MODE>
__global__ __launch_bounds__(MMC_THREADS) void k_sf_atp(Geom g, SfArrays a) {
    const int i = blockIdx.y, j = blockIdx.x * MMC_THREADS + threadIdx.x;
    if (j >= g.N) return;
    const size_t idx = (size_t)i * g.N + j;
    if (a.irregular[i]) return;
    const int c = fdiv(j, g.dv_Ny), r = j - c * g.Ny;
    a.wimg[idx] = sf_z<MODE>(g, a, i, r, c, j);
}

// The same at zoom 4, the way the primal sweep of mmc_kernels.hip forms A'p: the workgroup stages the
// low-resolution window of p_i under its 64 x 16 tile in LDS (zero outside the frame, so the sums need no
// masks), each thread owns four consecutive rows of one column, and the columns of a wave are taken 4
// apart, so that the kernel phase of every tap is wave-uniform: weights are scalar loads, the values of p
// are LDS reads at lane base + uniform offset.  Pixels on the frame border take the general gather.
// J3: kernels up to 12 x 12, taken from the table zero-padded to 12 x 12 (a.ck12): the 3 x 3 taps per row are
// unrolled without bounds, their weights are scalar loads at constant offsets.
template <bool J3>
__global__ __launch_bounds__(MMC_THREADS) void k_sf_atp4(Geom g, SfArrays a) {
    extern __shared__ __align__(16) float pl[];
    const int Jr = J3 ? 3 : g.Jr, Jc = J3 ? 3 : g.Jc;
    const int i = blockIdx.y, tile = blockIdx.x;
    if (a.irregular[i]) return;
    const int tc = fdiv(tile, g.dv_tiles_r), tr = tile - tc * g.tiles_r;
    const int r0 = tr * TILE_R, c0 = tc * TILE_C;
    const float *p = a.p + (size_t)i * g.n;
    float *z = a.wimg + (size_t)i * g.N;
    // low-resolution rows / columns with a kernel row / column on the tile (oy, ox <= 0)
    const int Rlo = ((r0 - g.oy) >> 2) - (Jr - 1), Rhi = (r0 + TILE_R - 1 - g.oy) >> 2;
    const int Clo = ((c0 - g.ox) >> 2) - (Jc - 1), Chi = (c0 + TILE_C - 1 - g.ox) >> 2;
    const int PR = Rhi - Rlo + 1, PC = Chi - Clo + 1;
    const float inv_pr = 1.f / (float)PR;
    for (int k = threadIdx.x; k < PR * PC; k += MMC_THREADS) {
        int kc, kr;
        divmod_small(k, PR, inv_pr, kc, kr);
        const int R = Rlo + kr, C = Clo + kc;
        pl[k] = (R >= 0 && R < g.ny && C >= 0 && C < g.nx) ? p[R + g.ny * C] : 0.f;
    }
    __syncthreads();
    const int rq = threadIdx.x & 15, cg = (threadIdx.x >> 4) & 3, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = r0 + 4 * rq, c = c0 + 4 * cg + wave;
    if (r >= g.Ny || c >= g.Nx) return;
    const int yb = c0 + wave - g.ox;                           // uniform part of y = c - ox
    const int n0 = yb & 3, Cl = cg + (yb >> 2) - Clo;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int xb = r0 + q - g.oy;                          // uniform part of x = r + q - oy
        const int m0 = xb & 3;
        const float *pq = pl + rq + (xb >> 2) - Rlo + PR * Cl;
        if (J3) {
            const float *ck = a.ck12 + m0 + 12 * n0;
#pragma unroll
            for (int jc = 0; jc < 3; ++jc)
#pragma unroll
                for (int jr = 0; jr < 3; ++jr) acc[q] += ck[4 * jr + 48 * jc] * pq[-jr - PR * jc];
            continue;
        }
        for (int jc = 0; jc < g.Jc; ++jc) {
            const int nn = n0 + 4 * jc;
            if (nn >= g.CW) break;
            for (int jr = 0; jr < g.Jr; ++jr) {
                const int mm = m0 + 4 * jr;
                if (mm >= g.CH) break;
                acc[q] += a.cker[mm + g.CH * nn] * pq[-jr - PR * jc];
            }
        }
    }
    const int j = r + g.Ny * c;
    const bool cborder = c == 0 || c == g.Nx - 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (r + q >= g.Ny) break;
        if (cborder || r + q == 0 || r + q == g.Ny - 1) acc[q] = gather_ATp(g, a.cker, p, r + q, c);
        z[j + q] = acc[q];
    }
}

// W' of the primal half-iteration per target tile and frame group: workgroup (tile, grp) walks, for
// the regular frames i = grp, grp + ngroups, ..., the source rectangle of its tile, takes A' p_i
// (k_sf_atp) at every source with a tap inside the tile and adds the tap contributions in LDS (64-bit
// fixed point, integer atomics); the tile's sums then go to a.acc with one global atomic per pixel.
// Splitting the frames over groups keeps small frames from leaving most CUs idle.
__global__ __launch_bounds__(MMC_THREADS) void k_sf_scatter_tile(Geom g, SfArrays a, int ngroups) {
    __shared__ fx64 acc[TILE_R * TILE_C];
    const int tile = blockIdx.x, grp = blockIdx.y;
    const int tc = fdiv(tile, g.dv_tiles_r), tr = tile - tc * g.tiles_r;
    const int r0 = tr * TILE_R, c0 = tc * TILE_C;
    for (int k = threadIdx.x; k < TILE_R * TILE_C; k += MMC_THREADS) acc[k] = 0;
    __syncthreads();
    bool any = false;
    for (int i = grp; i < a.nf; i += ngroups) {
        if (a.irregular[i]) continue;
        const int *bb = a.bbox + ((size_t)i * g.ntiles + tile) * 4;
        const int rmin = bb[0], rmax = bb[1], cmin = bb[2], cmax = bb[3];
        if (rmax < rmin) continue;
        any = true;
        const int bh = rmax - rmin + 1, cnt = bh * (cmax - cmin + 1);
        const float inv_bh = 1.f / (float)bh;
        const float *vx = a.vx + (size_t)i * g.N, *vy = a.vy + (size_t)i * g.N, *zi = a.wimg + (size_t)i * g.N;
        // four sources per thread and pass: their flow loads are issued together
        for (int kb = threadIdx.x; kb < cnt; kb += 4 * MMC_THREADS) {
            int sr[4], sc[4];
            float fx[4], fy[4], zz[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = kb + u * MMC_THREADS;
                const bool ok = k < cnt;
                int kc = 0, kr = 0;
                if (ok) divmod_small(k, bh, inv_bh, kc, kr);
                sr[u] = ok ? rmin + kr : -1;
                sc[u] = cmin + kc;
                const int j = ok ? sr[u] + g.Ny * sc[u] : 0;
                fx[u] = vx[j]; fy[u] = vy[j]; zz[u] = zi[j];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (sr[u] < 0) continue;
                const Taps tp = make_taps(g, sr[u], sc[u], fx[u], fy[u]);
                if (!tp.valid) continue;
                const bool ra = tp.r0 >= r0 && tp.r0 < r0 + TILE_R, rb = tp.r1 >= r0 && tp.r1 < r0 + TILE_R;
                const bool ca = tp.c0 >= c0 && tp.c0 < c0 + TILE_C, cb = tp.c1 >= c0 && tp.c1 < c0 + TILE_C;
                if (!((ra || rb) && (ca || cb))) continue;
                const float z = zz[u];
                if (z == 0.f) continue;
                const double s40 = fx_scaled(z);
                if (ra && ca && tp.w00 != 0.f) fx_add(acc + (tp.r0 - r0) + TILE_R * (tp.c0 - c0), tp.w00, s40);
                if (rb && ca && tp.w10 != 0.f) fx_add(acc + (tp.r1 - r0) + TILE_R * (tp.c0 - c0), tp.w10, s40);
                if (ra && cb && tp.w01 != 0.f) fx_add(acc + (tp.r0 - r0) + TILE_R * (tp.c1 - c0), tp.w01, s40);
                if (rb && cb && tp.w11 != 0.f) fx_add(acc + (tp.r1 - r0) + TILE_R * (tp.c1 - c0), tp.w11, s40);
            }
        }
    }
    if (!any) return;            // uniform over the workgroup
    __syncthreads();
    for (int k = threadIdx.x; k < TILE_R * TILE_C; k += MMC_THREADS) {
        const int lc = k / TILE_R, lr = k - lc * TILE_R;
        const int r = r0 + lr, c = c0 + lc;
        if (r >= g.Ny || c >= g.Nx || acc[k] == 0) continue;
        atomicAdd(a.acc + r + (size_t)g.Ny * c, acc[k]);
    }
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
static dim3 grid_frames(const Geom &g, const SfArrays &a) { return dim3((unsigned)((g.N + MMC_THREADS - 1) / MMC_THREADS), (unsigned)a.nf); }

template <int MODE, bool REDUCE>
static void launch_dual_mode(const Geom &g, const SfArrays &a, hipStream_t s) {
    const dim3 block(MMC_THREADS);
    const int SR = SF_TL * g.f + g.CH - g.f, SC = SF_TL * g.f + g.CW - g.f;
    if (SR * SC <= SF_WIN_MAX) {
        const int tiles_R = (g.ny + SF_TL - 1) / SF_TL, tiles_C = (g.nx + SF_TL - 1) / SF_TL;
        const size_t lds = (size_t)(SR * SC + g.CH * g.CW) * sizeof(float);
        hipLaunchKernelGGL((k_sf_dual_tile<MODE, REDUCE>), dim3(tiles_R * tiles_C, a.nf), block, lds, s, g, a, a.ub, tiles_R);
    } else {
        hipLaunchKernelGGL(k_sf_warp<MODE>, grid_frames(g, a), block, 0, s, g, a, a.ub);
        hipLaunchKernelGGL((k_sf_dual_p<MODE, REDUCE>), grid_for((size_t)g.n * a.nf), block, 0, s, g, a);
    }
    // the step of the gradient rows depends on the pixel position only; the update recomputes it
    if (MODE != MODE_STEPS) hipLaunchKernelGGL((k_sf_dual_q<MODE, REDUCE>), grid_for((size_t)g.N), block, 0, s, g, a, a.ub);
}

hipError_t mmc_sf_launch_dual(const Geom &g, const SfArrays &a, int mode, bool reduce, hipStream_t s) {
    if (mode == MODE_STEPS) launch_dual_mode<MODE_STEPS, false>(g, a, s);
    else if (mode == MODE_RAW) launch_dual_mode<MODE_RAW, false>(g, a, s);
    else if (reduce) launch_dual_mode<MODE_UPDATE, true>(g, a, s);
    else launch_dual_mode<MODE_UPDATE, false>(g, a, s);
    return hipGetLastError();
}

// frame groups of the tile scatter: enough workgroups to fill the GPU a few times over
static int scatter_groups(const Geom &g, const SfArrays &a) {
    const int want = (4096 + g.ntiles - 1) / g.ntiles;
    return want < 1 ? 1 : (want > a.nf ? a.nf : want);
}

template <int MODE, bool REDUCE>
static void launch_primal_mode(const Geom &g, const SfArrays &a, hipStream_t s) {
    const dim3 block(MMC_THREADS);
    const int ngroups = scatter_groups(g, a);
    if (a.any_irregular) hipLaunchKernelGGL(k_sf_scatter<MODE>, grid_frames(g, a), block, 0, s, g, a);
    if (MODE != MODE_STEPS && g.f == 4) {
        const int Jr = a.ck12_ok ? 3 : g.Jr, Jc = a.ck12_ok ? 3 : g.Jc;
        const size_t lds = (size_t)(TILE_R / 4 + Jr + 1) * (TILE_C / 4 + Jc + 1) * sizeof(float);
        if (a.ck12_ok) hipLaunchKernelGGL(k_sf_atp4<true>, dim3(g.ntiles, a.nf), block, lds, s, g, a);
        else hipLaunchKernelGGL(k_sf_atp4<false>, dim3(g.ntiles, a.nf), block, lds, s, g, a);
    }
    else hipLaunchKernelGGL(k_sf_atp<MODE>, grid_frames(g, a), block, 0, s, g, a);
    hipLaunchKernelGGL(k_sf_scatter_tile, dim3(g.ntiles, ngroups), block, 0, s, g, a, ngroups);
    hipLaunchKernelGGL((k_sf_primal<MODE, REDUCE>), grid_for((size_t)g.N), block, 0, s, g, a);
}

hipError_t mmc_sf_launch_primal(const Geom &g, const SfArrays &a, int mode, bool reduce, hipStream_t s) {
    if (mode == MODE_STEPS) launch_primal_mode<MODE_STEPS, false>(g, a, s);
    else if (mode == MODE_RAW) launch_primal_mode<MODE_RAW, false>(g, a, s);
    else if (reduce) launch_primal_mode<MODE_UPDATE, true>(g, a, s);
    else launch_primal_mode<MODE_UPDATE, false>(g, a, s);
    return hipGetLastError();
}

hipError_t mmc_sf_launch_bbox(const Geom &g, const SfArrays &a, hipStream_t s) {
    const int count = a.nf * g.ntiles * 4;
    hipLaunchKernelGGL(k_sf_bbox_clear, grid_for((size_t)count), dim3(MMC_THREADS), 0, s, a.bbox, count);
    hipLaunchKernelGGL(k_sf_bbox, grid_frames(g, a), dim3(MMC_THREADS), 0, s, g, a);
    return hipGetLastError();
}
