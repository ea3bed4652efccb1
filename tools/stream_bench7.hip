// Calibration 7: does a frame-interleaved plane layout let a tile column walk through all frames
// (so that the second use of a neighbouring frame's data is an L2 hit) without losing streaming rate?
// Model of the dual sweep: per 64x16 tile of frame t, 10 float4 read streams, 6 of them written back
// in place, plus (WIN) a re-read of planes 0 and 1 of frame t+1 over the tile shifted by one quad / column.
//   layout 0  plane[frame][column][row]                 (what the library uses)
//   layout 1  plane[column/16][frame][column%16][row]   (tile columns of all frames adjacent)
//   walk 0    XCD class = contiguous tile range of a frame, frame after frame (descending)
//   walk 1    XCD class = contiguous range of tile columns, each column through all frames (descending)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int NR = 10, NIP = 6, NPL = 10;
struct P { float *p[NPL]; };

template <int LAYOUT>
__device__ __forceinline__ size_t addr(int Ny, int Nx, int nf, int t, int r, int c) {
    if (LAYOUT == 0) return ((size_t)t * Nx + c) * Ny + r;
    return (((size_t)(c >> 4) * nf + t) * 16 + (c & 15)) * Ny + r;
}

template <int LAYOUT, int WALK, bool WIN>
__global__ __launch_bounds__(256) void k(P a, int Ny, int Nx, int nf) {
    const int tiles_r = (Ny + 63) / 64, tiles_c = (Nx + 15) / 16, ntiles = tiles_r * tiles_c;
    const int b = blockIdx.x, x = b & 7;
    int s = b >> 3, tr, tc, t;
    if (WALK == 0) {
        const int q = (ntiles + 7) >> 3, lo = x * q, cnt = min(q, ntiles - lo);
        if (cnt <= 0 || s >= cnt * nf) return;
        s = cnt * nf - 1 - s;
        const int trel = s / cnt, tile = lo + (s - trel * cnt);
        t = trel; tc = tile / tiles_r; tr = tile - tc * tiles_r;
    } else {
        const int q = (tiles_c + 7) >> 3, lo = x * q, cnt = min(q, tiles_c - lo);
        if (cnt <= 0 || s >= cnt * nf * tiles_r) return;
        const int per = nf * tiles_r, cl = s / per, rem = s - cl * per;
        const int trel = rem / tiles_r;
        tc = lo + cl; t = nf - 1 - trel; tr = rem - trel * tiles_r;
    }
    const int rq = threadIdx.x & 15, lc = threadIdx.x >> 4;
    const int r = tr * 64 + 4 * rq, c = tc * 16 + lc;
    if (r >= Ny || c >= Nx) return;
    const size_t j = addr<LAYOUT>(Ny, Nx, nf, t, r, c);
    float4 v[NR], sm = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NR; ++i) v[i] = *reinterpret_cast<const float4 *>(a.p[i] + j);
    if (WIN && t + 1 < nf) {
        const int r2 = min(r + 4, Ny - 4), c2 = min(c + 1, Nx - 1);
        const size_t j2 = addr<LAYOUT>(Ny, Nx, nf, t + 1, r2, c2);
        const float4 w0 = *reinterpret_cast<const float4 *>(a.p[0] + j2), w1 = *reinterpret_cast<const float4 *>(a.p[1] + j2);
        sm.x += w0.x + w1.x; sm.y += w0.y + w1.y; sm.z += w0.z + w1.z; sm.w += w0.w + w1.w;
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) { sm.x += v[i].x; sm.y += v[i].y; sm.z += v[i].z; sm.w += v[i].w; }
#pragma unroll
    for (int i = 0; i < NIP; ++i) { *reinterpret_cast<float4 *>(a.p[NR - NIP + i] + j) = sm; sm.x += 1.f; }
}

// One workgroup takes G consecutive frames of a tile (descending); only the first of them re-reads the
// neighbouring frame (the others would find it in LDS, left there by the previous step of the loop).
template <int G, bool PREFETCH>
__global__ __launch_bounds__(256) void kg(P a, int Ny, int Nx, int nf) {
    const int tiles_r = (Ny + 63) / 64, tiles_c = (Nx + 15) / 16, ntiles = tiles_r * tiles_c;
    const int ngrp = nf / G;
    const int b = blockIdx.x, x = b & 7;
    int s = b >> 3;
    const int q = (ntiles + 7) >> 3, lo = x * q, cnt = min(q, ntiles - lo);
    if (cnt <= 0 || s >= cnt * ngrp) return;
    s = cnt * ngrp - 1 - s;
    const int grp = s / cnt, tile = lo + (s - grp * cnt);
    const int tc = tile / tiles_r, tr = tile - tc * tiles_r;
    const int rq = threadIdx.x & 15, lc = threadIdx.x >> 4;
    const int r = tr * 64 + 4 * rq, c = tc * 16 + lc;
    if (r >= Ny || c >= Nx) return;
    float4 v[NR], nx[NR];
    if (PREFETCH) {
        const size_t j0 = addr<0>(Ny, Nx, nf, grp * G + G - 1, r, c);
#pragma unroll
        for (int i = 0; i < NR; ++i) nx[i] = *reinterpret_cast<const float4 *>(a.p[i] + j0);
    }
    for (int gi = 0; gi < G; ++gi) {
        const int t = grp * G + (G - 1 - gi);
        const size_t j = addr<0>(Ny, Nx, nf, t, r, c);
        float4 sm = make_float4(0, 0, 0, 0);
        if (PREFETCH) {
#pragma unroll
            for (int i = 0; i < NR; ++i) v[i] = nx[i];
            if (gi + 1 < G) {
                const size_t jn = addr<0>(Ny, Nx, nf, t - 1, r, c);
#pragma unroll
                for (int i = 0; i < NR; ++i) nx[i] = *reinterpret_cast<const float4 *>(a.p[i] + jn);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NR; ++i) v[i] = *reinterpret_cast<const float4 *>(a.p[i] + j);
        }
        if (gi == 0 && t + 1 < nf) {
            const int r2 = min(r + 4, Ny - 4), c2 = min(c + 1, Nx - 1);
            const size_t j2 = addr<0>(Ny, Nx, nf, t + 1, r2, c2);
            const float4 w0 = *reinterpret_cast<const float4 *>(a.p[0] + j2), w1 = *reinterpret_cast<const float4 *>(a.p[1] + j2);
            sm.x += w0.x + w1.x; sm.y += w0.y + w1.y; sm.z += w0.z + w1.z; sm.w += w0.w + w1.w;
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) { sm.x += v[i].x; sm.y += v[i].y; sm.z += v[i].z; sm.w += v[i].w; }
#pragma unroll
        for (int i = 0; i < NIP; ++i) { *reinterpret_cast<float4 *>(a.p[NR - NIP + i] + j) = sm; sm.x += 1.f; }
    }
}

template <int G, bool PREFETCH>
int rung(P a, int Ny, int Nx, int nf) {
    const int tiles_r = (Ny + 63) / 64, tiles_c = (Nx + 15) / 16, ntiles = tiles_r * tiles_c;
    const unsigned grid = 8u * ((ntiles + 7) / 8) * (nf / G);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((kg<G, PREFETCH>), dim3(grid), dim3(256), 0, 0, a, Ny, Nx, nf);
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((kg<G, PREFETCH>), dim3(grid), dim3(256), 0, 0, a, Ny, Nx, nf);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double alg = (double)Ny * Nx * nf * 4.0 * (NR + NIP);
    printf("groups of %2d frames per workgroup, prefetch=%d  grid=%6u  %7.3f ms  algorithmic %5.2f TB/s\n", G, (int)PREFETCH, grid, ms, alg / ms / 1e9);
    return 0;
}

template <int LAYOUT, int WALK, bool WIN>
int run(P a, int Ny, int Nx, int nf, const char *name) {
    const int tiles_r = (Ny + 63) / 64, tiles_c = (Nx + 15) / 16, ntiles = tiles_r * tiles_c;
    const unsigned grid = WALK == 0 ? 8u * ((ntiles + 7) / 8) * nf : 8u * ((tiles_c + 7) / 8) * nf * tiles_r;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<LAYOUT, WALK, WIN>), dim3(grid), dim3(256), 0, 0, a, Ny, Nx, nf);
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<LAYOUT, WALK, WIN>), dim3(grid), dim3(256), 0, 0, a, Ny, Nx, nf);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double alg = (double)Ny * Nx * nf * 4.0 * (NR + NIP);
    printf("%-44s layout=%d walk=%d win=%d  %7.3f ms  algorithmic %5.2f TB/s\n", name, LAYOUT, WALK, (int)WIN, ms, alg / ms / 1e9);
    return 0;
}

int main(int argc, char **argv) {
    const int Ny = argc > 1 ? atoi(argv[1]) : 1080, Nx = argc > 2 ? atoi(argv[2]) : 1920, nf = argc > 3 ? atoi(argv[3]) : 30;
    const size_t per = (size_t)Ny * (((size_t)Nx + 15) / 16 * 16) * nf;
    P a;
    for (int i = 0; i < NPL; ++i) { CK(hipMalloc((void **)&a.p[i], per * 4)); CK(hipMemset(a.p[i], 0, per * 4)); }
    if (run<0, 0, false>(a, Ny, Nx, nf, "frame-major layout, frame-major walk")) return 1;
    if (run<0, 0, true>(a, Ny, Nx, nf, "frame-major layout, frame-major walk")) return 1;
    if (run<0, 1, false>(a, Ny, Nx, nf, "frame-major layout, column-through-frames")) return 1;
    if (run<0, 1, true>(a, Ny, Nx, nf, "frame-major layout, column-through-frames")) return 1;
    if (run<1, 1, false>(a, Ny, Nx, nf, "interleaved layout, column-through-frames")) return 1;
    if (run<1, 1, true>(a, Ny, Nx, nf, "interleaved layout, column-through-frames")) return 1;
    if (run<1, 0, false>(a, Ny, Nx, nf, "interleaved layout, frame-major walk")) return 1;
    if (run<1, 0, true>(a, Ny, Nx, nf, "interleaved layout, frame-major walk")) return 1;
    if (rung<1, false>(a, Ny, Nx, nf) || rung<2, false>(a, Ny, Nx, nf) || rung<3, false>(a, Ny, Nx, nf) || rung<5, false>(a, Ny, Nx, nf) ||
        rung<6, false>(a, Ny, Nx, nf) || rung<10, false>(a, Ny, Nx, nf) || rung<15, false>(a, Ny, Nx, nf) || rung<30, false>(a, Ny, Nx, nf)) return 1;
    if (rung<2, true>(a, Ny, Nx, nf) || rung<3, true>(a, Ny, Nx, nf) || rung<5, true>(a, Ny, Nx, nf) || rung<10, true>(a, Ny, Nx, nf)) return 1;
    return 0;
}
