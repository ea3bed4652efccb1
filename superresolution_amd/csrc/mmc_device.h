// Device helpers shared by the kernel files (mmc_kernels.hip, mmc_sf_kernels.hip): bilinear warp
// taps, 64-bit fixed-point accumulation, magic-number division, reductions, the block walk of the
// row-quad sweeps and the gather forms of the adjoints of D*B and of the gradient.  HIP device code;
// include after <hip/hip_runtime.h> and mmc_internal.h.
#pragma once

#define FR(ptr, t) ((ptr) + (size_t)((t) + 1) * (size_t)g.N)

// --------------------------------------------------------------------------
// small device helpers
// --------------------------------------------------------------------------

__device__ __forceinline__ bool pair_exists(const Geom &g, int q) {
    int gq = g.frame_begin + q;
    return q >= -1 && q <= g.L - 1 && gq >= 0 && gq <= g.nc_global - 2;
}

// true: row block q is  W_q x_{q+1} - x_q  ('I-F', constructWarpFF.m:40-48)
// false:               W_q x_q - x_{q+1}  ('F-I', constructWarpFF.m:49-57)
__device__ __forceinline__ bool pair_is_IF(const Geom &g, int q) {
    if (g.dir == 1) return true;
    if (g.dir == 0) return false;
    return ((g.frame_begin + q) & 1) == 0;   // constructWarpFB.m:36, 1-based odd
}

struct Taps {
    int i00, i10, i01, i11;      // in-frame linear indices of the four taps
    int r0, r1, c0, c1;
    float w00, w10, w01, w11;
    bool valid;
};

// Bilinear taps of the sample position (r + vy, c + vx).  A position outside
// the closed pixel domain (or NaN) masks the row.  Float32 arithmetic in this
// exact order; the CPU oracle repeats it to reach the same floor() decisions.
__device__ __forceinline__ Taps make_taps(const Geom &g, int r, int c, float vx, float vy) {
    // Each operation is rounded on its own (no fused multiply-add, e.g. of 1 - a into the products):
    // every kernel that forms the taps of a pixel gets the same bits, as the oracle does.
#pragma clang fp contract(off)
    Taps t;
    float y = (float)r + vy;
    float x = (float)c + vx;
    t.valid = (y >= 0.f) && (y <= (float)(g.Ny - 1)) && (x >= 0.f) && (x <= (float)(g.Nx - 1));
    if (!t.valid) { y = 0.f; x = 0.f; }
    float y0 = floorf(y), x0 = floorf(x);
    float a = y - y0, b = x - x0;
    t.r0 = (int)y0;
    t.c0 = (int)x0;
    t.r1 = min(t.r0 + 1, g.Ny - 1);
    t.c1 = min(t.c0 + 1, g.Nx - 1);
    float na = 1.f - a, nb = 1.f - b;
    t.w00 = na * nb;
    t.w10 = a * nb;
    t.w01 = na * b;
    t.w11 = a * b;
    t.i00 = t.r0 + g.Ny * t.c0;
    t.i10 = t.r1 + g.Ny * t.c0;
    t.i01 = t.r0 + g.Ny * t.c1;
    t.i11 = t.r1 + g.Ny * t.c1;
    return t;
}

__device__ __forceinline__ float sample(const Taps &t, const float *img) {
    return t.w00 * img[t.i00] + t.w10 * img[t.i10] + t.w01 * img[t.i01] + t.w11 * img[t.i11];
}


// The adjoint of the bilinear warp scatters.  Sums are accumulated in LDS as
// 64-bit fixed point (scale 2^40): integer LDS atomics run at full LDS rate on
// gfx950 whereas float LDS atomics take about 3 clocks per lane, integer sums do
// not depend on arrival order (runs are bitwise reproducible), and 2^-40 is far
// below float32 resolution for any |value| < 2^23.
// Ablation switches for kernel tuning (env MMC_DEBUG, see tools/build_variant.sh): compiled in
// only with -DMMC_ABLATION; a regular build has none of these paths.
#ifdef MMC_ABLATION
#define ABL(bit) ((g.dbg & (bit)) != 0)
#else
#define ABL(bit) false
#endif
#ifndef MMC_CW_SCATTER
#define MMC_CW_SCATTER 1
#endif
typedef unsigned long long fx64;
// round(w * s40) as a two's-complement integer, |w * s40| < 2^51: adding 1.5*2^52 in double
// leaves the integer in the mantissa, so the bit pattern minus that of 1.5*2^52 is the result.
__device__ __forceinline__ fx64 to_fx(double w, double s40) {
    const double d = __fma_rn(w, s40, 6755399441055744.0);
    return (fx64)(__double_as_longlong(d) - 0x4338000000000000ll);
}
__device__ __forceinline__ double fx_scaled(float s) { return (double)s * 1099511627776.0; }
__device__ __forceinline__ float from_fx(fx64 v) {
    const double d = (double)(int)(unsigned)(v >> 32) * 4294967296.0 + (double)(unsigned)(v & 0xffffffffull);
    return (float)(d * (1.0 / 1099511627776.0));
}
// The fixed-point sum plus a small integer, rounded to float once (the sum of the scattered tap
// weights of a pixel plus its count of identity entries: every path that forms it gets the same bits).
__device__ __forceinline__ float from_fx_plus(fx64 v, int n) {
    const double d = (double)(int)(unsigned)(v >> 32) * 4294967296.0 + (double)(unsigned)(v & 0xffffffffull);
    return (float)(d * (1.0 / 1099511627776.0) + (double)n);
}
__device__ __forceinline__ void fx_add(fx64 *acc, float w, double s40) { atomicAdd(acc, to_fx((double)w, s40)); }

// The same interpolation from an LDS copy of the rows [rlo, rlo+WR) x columns [clo, ...) of the frame.
__device__ __forceinline__ float sample_lds(const Taps &t, const float *img, int rlo, int clo, int WR) {
    const int i0 = (t.r0 - rlo) + WR * (t.c0 - clo);
    const int dr = t.r1 - t.r0, dc = (t.c1 - t.c0) * WR;
    return t.w00 * img[i0] + t.w10 * img[i0 + dr] + t.w01 * img[i0 + dc] + t.w11 * img[i0 + dr + dc];
}

__device__ __forceinline__ int fdiv(int n, FastDiv d) { return d.m ? (int)(__umulhi((unsigned)n, d.m) >> d.s) : n; }
// Hardware reciprocal and reciprocal square root (1 ulp) in place of the correctly rounded
// division / sqrt sequences: step sizes and projection scales do not need the last bit.
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
// Scale that projects a vector of squared length n2 onto the ball of radius R.
__device__ __forceinline__ float ball_scale(float n2, float R) { return n2 > R * R ? R * fast_rsq(n2) : 1.f; }

// Squared distance of d = (K x)_j from the normal cone of the dual constraint set at the dual value,
// i.e. this row group's share of the dual optimality residual dist(K x, dF*(y))^2.
// Ball of radius R at g: the cone is {0} inside and {lambda g, lambda >= 0} on the sphere.
__device__ __forceinline__ float ball_residual2(float dt, float dx, float dy, float gt, float gx, float gy, float R) {
    const float n2 = gt * gt + gx * gx + gy * gy;
    float lam = 0.f;
    if (n2 > 0.f && n2 >= R * R * (1.f - 1e-5f)) lam = fmaxf(dt * gt + dx * gx + dy * gy, 0.f) / n2;
    const float rt = dt - lam * gt, rx = dx - lam * gx, ry = dy - lam * gy;
    return rt * rt + rx * rx + ry * ry;
}
// Interval [-b, b] at y: the cone is {0} inside, [0, inf) at y = b, (-inf, 0] at y = -b.
__device__ __forceinline__ float box_residual2(float d, float y, float b) {
    const float r = y >= b ? fminf(d, 0.f) : (y <= -b ? fmaxf(d, 0.f) : d);
    return r * r;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// Sum NV per-thread values over the workgroup: thread i < NV returns the sum of quantity i
// (wave shuffle, then the waves' sums in wave order), every other thread returns 0.
template <int NV>
__device__ __forceinline__ float block_sum(float (&v)[NV]) {
    __shared__ float red[NV][MMC_THREADS / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float s = wave_sum(v[i]);
        if (lane == 0) red[i][wid] = s;
    }
    __syncthreads();
    float s = 0.f;
    if (threadIdx.x < NV) {
#pragma unroll
        for (int k = 0; k < MMC_THREADS / 64; ++k) s += red[threadIdx.x][k];
    }
    return s;
}

// ... and add them to the slotted double accumulators sums[first + i][slot].
template <int NV>
__device__ __forceinline__ void block_accumulate(float (&v)[NV], double *sums, int first) {
    const float s = block_sum<NV>(v);
    if (threadIdx.x < NV) {
        int slot = (blockIdx.x + blockIdx.y * 7) & (MMC_SUM_SLOTS - 1);   // spread the atomics
        atomicAdd(&sums[(size_t)(first + threadIdx.x) * MMC_SUM_SLOTS + slot], (double)s);
    }
}


// Work item (tile, frame) of a workgroup in the 1-D grids of the row-quad
// kernels.  Workgroups b and b+8 are observed to share an XCD (and its L2).
// Column tiles are not aligned to 128-byte lines (a column is Ny*4 bytes), so
// vertically adjacent tiles share cache lines; residue class b%8 therefore owns
// a contiguous range of the (row-fastest) tile index and walks it in order,
// frame after frame, so that those shared lines are fetched once per L2 and
// DRAM pages are visited in address order.  Measured on a 9-read/4-write float4
// streaming kernel over 1080-row frames: 5.4 TB/s against 4.3 TB/s for
// round-robin tiles.  Placement only affects speed; surplus workgroups exit.
__device__ __forceinline__ bool decode_block(const Geom &g, int nf, int &tile, int &trel, bool reverse = false) {
    const int b = blockIdx.x;
    if (ABL(32)) { tile = b % g.ntiles; trel = b / g.ntiles; return trel < nf; }
    const int x = b & 7;
    int s = b >> 3;
    const int q = (g.ntiles + 7) >> 3;
    const int lo = x * q, cnt = min(q, g.ntiles - lo);
    if (cnt <= 0 || s >= cnt * nf) return false;
    if (reverse) s = cnt * nf - 1 - s;
    if (g.chunk > 0) {
        const int grp = fdiv(s, g.dv_per), rem = s - grp * (g.chunk * nf);
        const int sg = min(g.chunk, cnt - grp * g.chunk);
        trel = sg == g.chunk ? fdiv(rem, g.dv_chunk) : rem / sg;
        tile = lo + grp * g.chunk + (rem - trel * sg);
        return true;
    }
    trel = fdiv(s, cnt == q ? g.dv_q : g.dv_qlast);
    tile = lo + (s - trel * cnt);
    return true;
}

// (A' p)(r,c): gather form of the adjoint of D*B with the replicate border of
// helpers/RepConvMtx.m:23-32 folded in: a border pixel also collects the
// weights of the virtual rows / columns that clamp onto it.
__device__ __forceinline__ float gather_ATp(const Geom &g, const float *__restrict__ cker,
                                            const float *__restrict__ p, int r, int c) {
    int rlo = r, rhi = r, clo = c, chi = c;
    if (r == 0) rlo = min(g.oy, 0);
    if (r == g.Ny - 1) rhi = max(g.Ny - 1, g.f * (g.ny - 1) + g.oy + g.CH - 1);
    if (c == 0) clo = min(g.ox, 0);
    if (c == g.Nx - 1) chi = max(g.Nx - 1, g.f * (g.nx - 1) + g.ox + g.CW - 1);
    float acc = 0.f;
    for (int cv = clo; cv <= chi; ++cv) {
        int numc = cv - g.ox - g.CW + 1;
        int Clo = numc > 0 ? (numc + g.f - 1) / g.f : 0;
        int Chi = min((cv - g.ox) / g.f, g.nx - 1);
        for (int C = Clo; C <= Chi; ++C) {
            int nn = cv - g.f * C - g.ox;
            for (int rv = rlo; rv <= rhi; ++rv) {
                int numr = rv - g.oy - g.CH + 1;
                int Rlo = numr > 0 ? (numr + g.f - 1) / g.f : 0;
                int Rhi = min((rv - g.oy) / g.f, g.ny - 1);
                for (int R = Rlo; R <= Rhi; ++R) {
                    int mm = rv - g.f * R - g.oy;
                    acc += cker[mm + g.CH * nn] * p[R + g.ny * C];
                }
            }
        }
    }
    return acc;
}

// (G' (gx,gy))(r,c) for forward differences with a zero last column / row.
__device__ __forceinline__ float grad_adj(const Geom &g, const float *__restrict__ gx,
                                          const float *__restrict__ gy, int r, int c, int j) {
    float d = 0.f;
    if (c > 0) d += gx[j - g.Ny];
    if (c < g.Nx - 1) d -= gx[j];
    if (r > 0) d += gy[j - 1];
    if (r < g.Ny - 1) d -= gy[j];
    return d;
}
