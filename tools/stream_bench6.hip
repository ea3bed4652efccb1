// Calibration 6: what bounds a many-plane streaming sweep?  NR float4 read streams and NW write
// streams, IP of the writes in place (they overwrite the first IP read planes, as the dual sweep
// overwrites the six dual planes it reads).  Layouts:
//   planar       one array per plane (what the library uses)
//   interleaved  chunks of 256 float4 (4 KiB) per plane, the planes of a chunk adjacent, so that a
//                workgroup's accesses fall into one contiguous region
// Prints moved bytes / time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NR, int NW, int IP, bool INTER>
__global__ __launch_bounds__(256) void k(float4 *base, size_t plane_quads, int nplanes, int nchunks) {
    for (int ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
        float4 v[NR], s = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const size_t q = INTER ? ((size_t)ch * nplanes + i) * 256 + threadIdx.x : (size_t)i * plane_quads + (size_t)ch * 256 + threadIdx.x;
            v[i] = base[q];
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) { s.x += v[i].x; s.y += v[i].y; s.z += v[i].z; s.w += v[i].w; }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int pl = i < IP ? i : NR + (i - IP);
            const size_t q = INTER ? ((size_t)ch * nplanes + pl) * 256 + threadIdx.x : (size_t)pl * plane_quads + (size_t)ch * 256 + threadIdx.x;
            base[q] = s;
            s.x += 1.f;
        }
    }
}

template <int NR, int NW, int IP, bool INTER>
int run(float4 *base, size_t plane_quads, const char *name, unsigned grid_mult) {
    const int nchunks = (int)(plane_quads / 256), nplanes = NR + NW - IP;
    const unsigned grid = grid_mult ? 256u * grid_mult : (unsigned)nchunks;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<NR, NW, IP, INTER>), dim3(grid), dim3(256), 0, 0, base, plane_quads, nplanes, nchunks);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<NR, NW, IP, INTER>), dim3(grid), dim3(256), 0, 0, base, plane_quads, nplanes, nchunks);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    printf("%-28s NR=%2d NW=%d IP=%d %s grid=%7u  %7.3f ms  %5.2f TB/s\n", name, NR, NW, IP, INTER ? "interleaved" : "planar     ", grid, ms,
           (double)plane_quads * 16 * (NR + NW) / ms / 1e9);
    return 0;
}

int main() {
    const size_t plane_quads = (size_t)270 * 1920 * 30;          // 30 frames of 1080x1920 floats
    float4 *base;
    CK(hipMalloc((void **)&base, plane_quads * 16 * 16));
    CK(hipMemset(base, 0, plane_quads * 16 * 16));
    for (unsigned gm : {0u, 8u, 16u}) {
        if (run<1, 1, 0, false>(base, plane_quads, "copy", gm)) return 1;
        if (run<1, 1, 1, false>(base, plane_quads, "in-place 1", gm)) return 1;
        if (run<13, 0, 0, false>(base, plane_quads, "read only", gm)) return 1;
        if (run<9, 4, 0, false>(base, plane_quads, "9r 4w", gm)) return 1;
        if (run<9, 4, 0, true>(base, plane_quads, "9r 4w", gm)) return 1;
        if (run<10, 6, 6, false>(base, plane_quads, "dual-like 10r 6w in place", gm)) return 1;
        if (run<10, 6, 6, true>(base, plane_quads, "dual-like 10r 6w in place", gm)) return 1;
        if (run<10, 4, 2, false>(base, plane_quads, "primal-like 10r 4w 2 in pl.", gm)) return 1;
        if (run<10, 4, 2, true>(base, plane_quads, "primal-like 10r 4w 2 in pl.", gm)) return 1;
    }
    return 0;
}
