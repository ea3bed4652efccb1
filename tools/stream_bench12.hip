// Calibration 12: what makes a launch of 1152 workgroups that moves 61 MB (the primal sweep of the quick-start
// size with all its arithmetic compiled out) take 16 us when a plain streaming kernel over as many bytes takes
// 10?  The same bytes per thread as that skeleton (13 16-byte loads, 4 16-byte stores), with, one at a time:
// 30 KB of dynamic LDS per workgroup, two barriers between the loads and the stores, a staging round trip
// through LDS, 800 bytes of kernel arguments, 90 live registers, and the loads in two dependent rounds.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int NL = 13, NS = 4;
struct Big { float *p[NL]; int pad[170]; };      // about 800 bytes
struct Small { float *p[NL]; };

template <typename A, bool BARRIERS, bool STAGE, bool TWO_ROUNDS, bool REGS>
__global__ __launch_bounds__(256) void k(A a, size_t nquads) {
    extern __shared__ __align__(16) float lds[];
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= nquads) return;
    float4 v[NL];
    constexpr int FIRST = TWO_ROUNDS ? 8 : NL;
#pragma unroll
    for (int i = 0; i < FIRST; ++i) v[i] = reinterpret_cast<const float4 *>(a.p[i])[q];
    if (STAGE) {
        reinterpret_cast<float4 *>(lds)[threadIdx.x] = v[0];
        __syncthreads();
        v[0] = reinterpret_cast<float4 *>(lds)[threadIdx.x ^ 1];
    } else if (BARRIERS) {
        __syncthreads();
    }
    if (BARRIERS) __syncthreads();
    if (TWO_ROUNDS) {
        // the second round's addresses depend on the first round's data having arrived
        const size_t q2 = q + (v[0].x == 12345.f ? 1 : 0);
#pragma unroll
        for (int i = FIRST; i < NL; ++i) v[i] = reinterpret_cast<const float4 *>(a.p[i])[q2];
    }
    float4 sm = make_float4(0, 0, 0, 0);
    if (REGS) {
        // keep every loaded value live until the stores (the sweeps hold 86-95 registers)
        float acc[NL * 4];
#pragma unroll
        for (int i = 0; i < NL; ++i) { acc[4 * i] = v[i].x; acc[4 * i + 1] = v[i].y; acc[4 * i + 2] = v[i].z; acc[4 * i + 3] = v[i].w; }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NL * 4; ++i) acc[i] = acc[i] * 1.0001f + acc[(i + 7) % (NL * 4)];
#pragma unroll
        for (int i = 0; i < NL * 4; ++i) sm.x += acc[i];
    } else {
#pragma unroll
        for (int i = 0; i < NL; ++i) { sm.x += v[i].x; sm.y += v[i].y; sm.z += v[i].z; sm.w += v[i].w; }
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) { reinterpret_cast<float4 *>(a.p[i])[q] = sm; sm.x += 1e-30f; }
}

template <typename F>
static float time_it(F launch, hipEvent_t e0, hipEvent_t e1) {
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e0);
    const int reps = 200;
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1e3f;
}

int main() {
    const size_t nwg = 1152, nquads = nwg * 256, plane = nquads * 16;      // 4.7 MB per plane, 61 MB loaded, 19 MB stored
    char *base;
    CK(hipMalloc((void **)&base, NL * (plane + 4096)));
    CK(hipMemset(base, 0, NL * (plane + 4096)));
    Small s; Big b;
    for (int i = 0; i < NL; ++i) s.p[i] = b.p[i] = reinterpret_cast<float *>(base + i * (plane + 4096));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const dim3 grid((unsigned)nwg), block(256);
    for (int round = 0; round < 2; ++round) {
        printf("plain                          %6.2f us\n", time_it([&] { hipLaunchKernelGGL((k<Small, false, false, false, false>), grid, block, 0, 0, s, nquads); }, e0, e1));
        printf("30 KB LDS per workgroup        %6.2f us\n", time_it([&] { hipLaunchKernelGGL((k<Small, false, false, false, false>), grid, block, 30000, 0, s, nquads); }, e0, e1));
        printf("two barriers                   %6.2f us\n", time_it([&] { hipLaunchKernelGGL((k<Small, true, false, false, false>), grid, block, 0, 0, s, nquads); }, e0, e1));
        printf("staging through LDS + barriers %6.2f us\n", time_it([&] { hipLaunchKernelGGL((k<Small, true, true, false, false>), grid, block, 30000, 0, s, nquads); }, e0, e1));
        printf("800 bytes of arguments         %6.2f us\n", time_it([&] { hipLaunchKernelGGL((k<Big, false, false, false, false>), grid, block, 0, 0, b, nquads); }, e0, e1));
        printf("all loaded values live         %6.2f us\n", time_it([&] { hipLaunchKernelGGL((k<Small, false, false, false, true>), grid, block, 0, 0, s, nquads); }, e0, e1));
        printf("loads in two dependent rounds  %6.2f us\n", time_it([&] { hipLaunchKernelGGL((k<Small, false, false, true, false>), grid, block, 0, 0, s, nquads); }, e0, e1));
        printf("all of it                      %6.2f us\n", time_it([&] { hipLaunchKernelGGL((k<Big, true, true, true, true>), grid, block, 30000, 0, b, nquads); }, e0, e1));
    }
    return 0;
}
