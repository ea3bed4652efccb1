// Calibration 9: what does a hit in the 256 MB Infinity Cache cost?  A working set of S bytes (10
// planes) is walked again and again by back-to-back launches: read-only (10 float4 streams) and in the
// dual sweep's mix (10 read, 6 written back in place).  S below ~32 MB stays in the L2s, S up to
// ~200 MB in the Infinity Cache, larger S streams from HBM.  Second experiment: a producer launch
// writes 4 planes of a band of B bytes per plane and a consumer launch reads them right after
// (the order a frame-band wavefront of primal and dual sweeps would give), bands cycling through 3 GB.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int NR = 10;
struct P { float *p[NR]; };

template <int NW>
__global__ __launch_bounds__(256) void k(P a, size_t nquads) {
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= nquads) return;
    float4 v[NR], sm = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NR; ++i) v[i] = reinterpret_cast<const float4 *>(a.p[i])[q];
#pragma unroll
    for (int i = 0; i < NR; ++i) { sm.x += v[i].x; sm.y += v[i].y; sm.z += v[i].z; sm.w += v[i].w; }
    if (NW == 0) { if (sm.x == 12345.f) reinterpret_cast<float4 *>(a.p[0])[q] = sm; }
#pragma unroll
    for (int i = 0; i < NW; ++i) { reinterpret_cast<float4 *>(a.p[NR - NW + i])[q] = sm; sm.x += 1e-30f; }
}

// producer: reads planes 0..9 at the band, writes planes 0..3; consumer: reads planes 0..3 (just written) and
// 4..9 (just read by the producer) plus writes 4..9
__global__ __launch_bounds__(256) void k_prod(P a, size_t q0, size_t nquads) {
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= nquads) return;
    float4 sm = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NR; ++i) { const float4 v = reinterpret_cast<const float4 *>(a.p[i])[q0 + q]; sm.x += v.x; sm.y += v.y; sm.z += v.z; sm.w += v.w; }
#pragma unroll
    for (int i = 0; i < 4; ++i) reinterpret_cast<float4 *>(a.p[i])[q0 + q] = sm;
}
__global__ __launch_bounds__(256) void k_cons(P a, size_t q0, size_t nquads) {
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= nquads) return;
    float4 sm = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NR; ++i) { const float4 v = reinterpret_cast<const float4 *>(a.p[i])[q0 + q]; sm.x += v.x; sm.y += v.y; sm.z += v.z; sm.w += v.w; }
#pragma unroll
    for (int i = 4; i < NR; ++i) reinterpret_cast<float4 *>(a.p[i])[q0 + q] = sm;
}

int main() {
    const size_t plane = (size_t)320 << 20;                 // 320 MiB per plane, 3.2 GB in all
    const size_t stride = plane + (320u << 10);
    char *base;
    CK(hipMalloc((void **)&base, NR * stride));
    CK(hipMemset(base, 0, NR * stride));
    P a;
    for (int i = 0; i < NR; ++i) a.p[i] = reinterpret_cast<float *>(base + i * stride);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (size_t mb : {8, 16, 32, 64, 96, 128, 160, 192, 224, 256, 320, 512, 1024, 3200}) {
        const size_t per_plane = (mb << 20) / NR, nquads = per_plane / 16;
        const unsigned grid = (unsigned)((nquads + 255) / 256);
        const int reps = (int)std::max<size_t>(4, (size_t)6400 / mb);
        for (int mode = 0; mode < 2; ++mode) {
            for (int i = 0; i < 3; ++i) { if (mode) hipLaunchKernelGGL(k<6>, dim3(grid), dim3(256), 0, 0, a, nquads); else hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, a, nquads); }
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) { if (mode) hipLaunchKernelGGL(k<6>, dim3(grid), dim3(256), 0, 0, a, nquads); else hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, a, nquads); }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
            printf("working set %5zu MB  %s  %8.4f ms  %6.2f TB/s moved\n", mb, mode ? "10r/6w" : "10r   ", ms,
                   (double)per_plane * (NR + (mode ? 6 : 0)) / ms / 1e9);
        }
    }
    // wavefront model: producer band then consumer band, bands cycling through the planes
    for (size_t bmb : {1, 2, 4, 8, 16, 32, 80}) {
        const size_t band = bmb << 20, nquads = band / 16, nb = plane / band;
        const unsigned grid = (unsigned)((nquads + 255) / 256);
        for (int lag = 0; lag < 2; ++lag) {              // consumer works on the band the producer wrote `lag+1` launches ago
            for (int pass = 0; pass < 2; ++pass) {
                if (pass) CK(hipEventRecord(e0));
                for (size_t b = 0; b < nb + lag; ++b) {
                    if (b < nb) hipLaunchKernelGGL(k_prod, dim3(grid), dim3(256), 0, 0, a, b * nquads, nquads);
                    if (b >= (size_t)lag) hipLaunchKernelGGL(k_cons, dim3(grid), dim3(256), 0, 0, a, (b - lag) * nquads, nquads);
                }
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("wavefront, band %3zu MB/plane, lag %d: %8.3f ms for 3.2 GB x (20r + 10w)  %6.2f TB/s moved (%zu launches)\n",
                   bmb, lag, ms, (double)plane * 30 / ms / 1e9, 2 * nb);
        }
    }
    // the same work as two whole sweeps
    {
        const size_t nquads = plane / 16; const unsigned grid = (unsigned)((nquads + 255) / 256);
        for (int pass = 0; pass < 2; ++pass) {
            if (pass) CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_prod, dim3(grid), dim3(256), 0, 0, a, (size_t)0, nquads);
            hipLaunchKernelGGL(k_cons, dim3(grid), dim3(256), 0, 0, a, (size_t)0, nquads);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("two whole sweeps: %8.3f ms  %6.2f TB/s moved\n", ms, (double)plane * 30 / ms / 1e9);
    }
    return 0;
}
