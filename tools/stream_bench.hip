// Calibration: achievable HBM rate of a float4 streaming kernel that reads NR
// arrays and writes NW arrays (each ARR_BYTES large), linear vs column-tile order.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Ptrs { float4 *p[16]; };

template <int NR, int NW, int TILED>
__global__ __launch_bounds__(256) void k(Ptrs a, size_t nquads, int Ny4, int Nx) {
    size_t q;
    if (TILED) {
        // 64-row x 16-col tiles of a column-major (Ny4*4 x Nx) frame, one quad per thread
        const int tiles_r = (Ny4 + 15) / 16;
        const size_t per_frame = (size_t)Ny4 * Nx;
        const int tiles_per_frame = tiles_r * ((Nx + 15) / 16);
        const size_t b = blockIdx.x;
        const int t = (int)(b / tiles_per_frame), tile = (int)(b % tiles_per_frame);
        const int tr = tile % tiles_r, tc = tile / tiles_r;
        const int rq = tr * 16 + (threadIdx.x & 15), c = tc * 16 + (threadIdx.x >> 4);
        if (rq >= Ny4 || c >= Nx) return;
        q = (size_t)t * per_frame + (size_t)c * Ny4 + rq;
    } else {
        q = (size_t)blockIdx.x * 256 + threadIdx.x;
    }
    if (q >= nquads) return;
    float4 s = make_float4(0, 0, 0, 0);
    float4 v[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) v[i] = a.p[i][q];
#pragma unroll
    for (int i = 0; i < NR; ++i) { s.x += v[i].x; s.y += v[i].y; s.z += v[i].z; s.w += v[i].w; }
#pragma unroll
    for (int i = 0; i < NW; ++i) { a.p[NR + i][q] = s; s.x += 1.f; }
}

template <int NR, int NW, int TILED>
int run(Ptrs a, size_t nquads, int Ny4, int Nx, int nframes) {
    unsigned grid = TILED ? (unsigned)(((Ny4 + 15) / 16) * ((Nx + 15) / 16) * nframes) : (unsigned)((nquads + 255) / 256);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) k<NR, NW, TILED><<<grid, 256>>>(a, nquads, Ny4, Nx);
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) k<NR, NW, TILED><<<grid, 256>>>(a, nquads, Ny4, Nx);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    double bytes = (double)nquads * 16 * (NR + NW);
    printf("NR=%2d NW=%d %-6s %7.3f ms  %6.2f TB/s\n", NR, NW, TILED ? "tiled" : "linear", ms, bytes / ms / 1e9);
    return 0;
}

int main() {
    const int Ny4 = 270, Nx = 1920, nframes = 30;
    const size_t nquads = (size_t)Ny4 * Nx * nframes;
    Ptrs a;
    for (int i = 0; i < 16; ++i) { CK(hipMalloc((void **)&a.p[i], nquads * 16)); CK(hipMemset(a.p[i], 0, nquads * 16)); }
    if (run<1, 1, 0>(a, nquads, Ny4, Nx, nframes)) return 1;
    if (run<1, 1, 1>(a, nquads, Ny4, Nx, nframes)) return 1;
    if (run<9, 4, 0>(a, nquads, Ny4, Nx, nframes)) return 1;
    if (run<9, 4, 1>(a, nquads, Ny4, Nx, nframes)) return 1;
    if (run<10, 6, 0>(a, nquads, Ny4, Nx, nframes)) return 1;
    if (run<10, 6, 1>(a, nquads, Ny4, Nx, nframes)) return 1;
    if (run<12, 0, 0>(a, nquads, Ny4, Nx, nframes)) return 1;
    if (run<4, 0, 0>(a, nquads, Ny4, Nx, nframes)) return 1;
    return 0;
}
