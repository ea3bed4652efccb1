// Calibration 2: 9 read + 4 write float4 streams over column-major frames
// (Ny4 quads per column), tile = QR quads x TC columns per 256-thread block.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Ptrs { float4 *p[16]; };
constexpr int NR = 9, NW = 4;

__global__ __launch_bounds__(256) void k(Ptrs a, int Ny4, int Nx, int QR, int TC, int xcd_mode) {
    const int tiles_r = (Ny4 + QR - 1) / QR, tiles_c = (Nx + TC - 1) / TC;
    const int tpf = tiles_r * tiles_c;
    int b = blockIdx.x, t, tile;
    if (xcd_mode) {
        const int nf = gridDim.x / (8 * ((tpf + 7) / 8));
        const int x = b & 7, s = b >> 3, q = (tpf + 7) >> 3, lo = x * q, cnt = min(q, tpf - lo);
        if (cnt <= 0 || s >= cnt * nf) return;
        t = s / cnt; tile = lo + (s - t * cnt);
    } else { t = b / tpf; tile = b % tpf; }
    const int tr = tile % tiles_r, tc = tile / tiles_r;
    const size_t per_frame = (size_t)Ny4 * Nx;
    for (int k = threadIdx.x; k < QR * TC; k += 256) {
        const int rq = tr * QR + k % QR, c = tc * TC + k / QR;
        if (rq >= Ny4 || c >= Nx) continue;
        const size_t q = (size_t)t * per_frame + (size_t)c * Ny4 + rq;
        float4 s = make_float4(0, 0, 0, 0), v[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) v[i] = a.p[i][q];
#pragma unroll
        for (int i = 0; i < NR; ++i) { s.x += v[i].x; s.y += v[i].y; s.z += v[i].z; s.w += v[i].w; }
#pragma unroll
        for (int i = 0; i < NW; ++i) { a.p[NR + i][q] = s; s.x += 1.f; }
    }
}

int run(Ptrs a, int Ny4, int Nx, int nf, int QR, int TC, int xcd) {
    const int tpf = ((Ny4 + QR - 1) / QR) * ((Nx + TC - 1) / TC);
    unsigned grid = xcd ? 8u * ((tpf + 7) / 8) * nf : (unsigned)tpf * nf;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) k<<<grid, 256>>>(a, Ny4, Nx, QR, TC, xcd);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) k<<<grid, 256>>>(a, Ny4, Nx, QR, TC, xcd);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    double bytes = (double)Ny4 * Nx * nf * 16 * (NR + NW);
    printf("QR=%3d (rows %4d) TC=%2d xcd=%d  %7.3f ms  %5.2f TB/s\n", QR, QR * 4, TC, xcd, ms, bytes / ms / 1e9);
    return 0;
}

int main_unused() {
    const int Ny4 = 270, Nx = 1920, nf = 30;
    const size_t nquads = (size_t)Ny4 * Nx * nf;
    Ptrs a;
    for (int i = 0; i < 13; ++i) { CK(hipMalloc((void **)&a.p[i], nquads * 16)); CK(hipMemset(a.p[i], 0, nquads * 16)); }
    int cfg[][2] = {{16, 16}, {32, 8}, {64, 4}, {30, 8}, {45, 4}, {90, 2}, {90, 4}, {135, 2}, {270, 1}, {270, 2}, {270, 4}};
    for (auto &c : cfg) for (int x = 0; x < 2; ++x) if (run(a, Ny4, Nx, nf, c[0], c[1], x)) return 1;
    return 0;
}

// Calibration 3: does a 128-byte aligned column pitch (272 quads = 1088 rows) stream faster than
// the natural 1080-row pitch?  Same kernel, 16x16-quad tiles, contiguous XCD ranges.
__global__ __launch_bounds__(256) void klin(Ptrs a, size_t nq) {
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < nq; q += (size_t)gridDim.x * 256) {
        float4 s = make_float4(0, 0, 0, 0), v[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) v[i] = a.p[i][q];
#pragma unroll
        for (int i = 0; i < NR; ++i) { s.x += v[i].x; s.y += v[i].y; s.z += v[i].z; s.w += v[i].w; }
#pragma unroll
        for (int i = 0; i < NW; ++i) { a.p[NR + i][q] = s; s.x += 1.f; }
    }
}
int main() {
    const int Nx = 1920, nf = 30;
    const size_t nquads = (size_t)272 * Nx * nf;
    Ptrs a;
    for (int i = 0; i < 13; ++i) { CK(hipMalloc((void **)&a.p[i], nquads * 16)); CK(hipMemset(a.p[i], 0, nquads * 16)); }
    for (int rep = 0; rep < 2; ++rep) {
        for (int Ny4 : {270, 272, 256}) if (run(a, Ny4, Nx, nf, 16, 16, 1)) return 1;
        for (int Ny4 : {270, 272}) if (run(a, Ny4, Nx, nf, 16, 16, 0)) return 1;
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (unsigned grid : {2048u, 4096u, 16384u, 65536u}) {
        klin<<<grid, 256>>>(a, nquads);
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) klin<<<grid, 256>>>(a, nquads);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
        printf("linear grid=%6u  %7.3f ms  %5.2f TB/s\n", grid, ms, (double)nquads * 16 * 13 / ms / 1e9);
    }
    return 0;
}
