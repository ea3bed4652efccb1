// Calibration 8: do the base addresses of the planes matter?  The dual-sweep model of calibration 7
// (10 float4 read streams, 6 written back in place, frame-major XCD walk, no window re-read) with the
// planes cut from one allocation at a stride of (plane size rounded up to 2 MiB) + PAD bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int NR = 10, NIP = 6;
struct P { float *p[NR]; };

__global__ __launch_bounds__(256) void k(P a, int Ny, int Nx, int nf) {
    const int tiles_r = (Ny + 63) / 64, tiles_c = (Nx + 15) / 16, ntiles = tiles_r * tiles_c;
    const int b = blockIdx.x, x = b & 7;
    int s = b >> 3;
    const int q = (ntiles + 7) >> 3, lo = x * q, cnt = min(q, ntiles - lo);
    if (cnt <= 0 || s >= cnt * nf) return;
    s = cnt * nf - 1 - s;
    const int t = s / cnt, tile = lo + (s - t * cnt);
    const int tc = tile / tiles_r, tr = tile - tc * tiles_r;
    const int rq = threadIdx.x & 15, lc = threadIdx.x >> 4;
    const int r = tr * 64 + 4 * rq, c = tc * 16 + lc;
    if (r >= Ny || c >= Nx) return;
    const size_t j = ((size_t)t * Nx + c) * Ny + r;
    float4 v[NR], sm = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NR; ++i) v[i] = *reinterpret_cast<const float4 *>(a.p[i] + j);
#pragma unroll
    for (int i = 0; i < NR; ++i) { sm.x += v[i].x; sm.y += v[i].y; sm.z += v[i].z; sm.w += v[i].w; }
#pragma unroll
    for (int i = 0; i < NIP; ++i) { *reinterpret_cast<float4 *>(a.p[NR - NIP + i] + j) = sm; sm.x += 1.f; }
}

int main() {
    const int Ny = 1080, Nx = 1920, nf = 30;
    const size_t plane = (size_t)Ny * Nx * nf * 4, plane2m = (plane + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
    char *base;
    const size_t maxpad = 4u << 20;
    CK(hipMalloc((void **)&base, NR * (plane2m + maxpad)));
    CK(hipMemset(base, 0, NR * (plane2m + maxpad)));
    std::vector<size_t> pads = {0};
    for (size_t kk : {32, 64, 96, 128, 160, 192, 224, 256, 320, 384, 448, 512, 576, 640, 704, 768, 832, 896, 960, 1024, 1088, 1152, 1280, 1536, 1600, 1792, 1856, 1984}) pads.push_back(kk << 10);
    pads.push_back(0);
    const int tiles = ((Ny + 63) / 64) * ((Nx + 15) / 16);
    const unsigned grid = 8u * ((tiles + 7) / 8) * nf;
    for (size_t pad : pads) {
        P a;
        for (int i = 0; i < NR; ++i) a.p[i] = reinterpret_cast<float *>(base + i * (plane2m + pad));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, a, Ny, Nx, nf);
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, a, Ny, Nx, nf);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
        printf("plane stride = 2 MiB multiple + %8zu B   %7.3f ms   %5.2f TB/s\n", pad, ms, (double)plane * (NR + NIP) / ms / 1e9);
    }
    return 0;
}
