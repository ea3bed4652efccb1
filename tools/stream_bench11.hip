// Calibration 11: does the number of separate planes matter?  The dual sweep's mix (10 streams read, 6 of
// them written back) over 3.2 GB, once with every stream in a plane of its own (as the library lays its
// variables out) and once with the streams interleaved in one array at a granularity of G bytes (the 16-byte
// quads of one stream stay together in runs of G bytes, the runs of the 10 streams follow each other), so that
// what a workgroup touches is one contiguous region instead of ten regions 320 MB apart.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int NR = 10, NW = 6;

// quads per run = G / 16; quad q of stream i sits at ((q / QR) * NR + i) * QR + q % QR
template <bool INTER>
__global__ __launch_bounds__(256) void k(float4 *base, size_t nquads, size_t plane_quads, int QR) {
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= nquads) return;
    float4 v[NR], sm = make_float4(0, 0, 0, 0);
    size_t at[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) at[i] = INTER ? ((q / QR) * NR + i) * QR + q % QR : (size_t)i * plane_quads + q;
#pragma unroll
    for (int i = 0; i < NR; ++i) v[i] = base[at[i]];
#pragma unroll
    for (int i = 0; i < NR; ++i) { sm.x += v[i].x; sm.y += v[i].y; sm.z += v[i].z; sm.w += v[i].w; }
#pragma unroll
    for (int i = 0; i < NW; ++i) { base[at[NR - NW + i]] = sm; sm.x += 1e-30f; }
}

int main() {
    const size_t plane = (size_t)320 << 20, nquads = plane / 16;
    const size_t stride_quads = (plane + (320u << 10)) / 16;           // planes skewed as in the library
    float4 *base;
    CK(hipMalloc((void **)&base, NR * stride_quads * 16));
    CK(hipMemset(base, 0, NR * stride_quads * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned grid = (unsigned)((nquads + 255) / 256);
    const int reps = 10;
    for (int round = 0; round < 2; ++round) {
        for (int QR : {0, 16, 64, 256, 1024, 4096}) {
            for (int i = 0; i < reps + 2; ++i) {
                if (i == 2) CK(hipEventRecord(e0));
                if (QR == 0) hipLaunchKernelGGL(k<false>, dim3(grid), dim3(256), 0, 0, base, nquads, stride_quads, 1);
                else hipLaunchKernelGGL(k<true>, dim3(grid), dim3(256), 0, 0, base, nquads, stride_quads, QR);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
            if (QR == 0) printf("separate planes             %8.4f ms  %6.2f TB/s moved\n", ms, (double)plane * (NR + NW) / ms / 1e9);
            else printf("interleaved, runs of %6d B %8.4f ms  %6.2f TB/s moved\n", QR * 16, ms, (double)plane * (NR + NW) / ms / 1e9);
        }
    }
    return 0;
}
