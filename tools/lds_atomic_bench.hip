// Microbenchmark: throughput of LDS atomic adds on gfx950 (float, u32, u64) with
// conflict-free consecutive addresses, one wave-instruction per lane per step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <typename T> __device__ T mk(int i);
template <> __device__ float mk<float>(int i) { return 1.0f + i; }
template <> __device__ unsigned mk<unsigned>(int i) { return 1u + i; }
template <> __device__ unsigned long long mk<unsigned long long>(int i) { return 1ull + i; }

template <typename T, int MODE>
__global__ __launch_bounds__(256) void k(T *out, int iters) {
    __shared__ T acc[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) acc[i] = T(0);
    __syncthreads();
    T v = mk<T>(threadIdx.x);
    int idx = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) atomicAdd(&acc[idx], v);                 // atomic
        else { acc[idx] = acc[idx] + v; }                       // plain read-modify-write
        idx = (idx + 257) & 2047;
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = acc[5];
}

template <typename T, int MODE>
void run(const char *name) {
    T *out;
    hipMalloc(&out, 4096 * sizeof(T));
    const int blocks = 2048, iters = 2000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<T, MODE><<<blocks, 256>>>(out, iters);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<T, MODE><<<blocks, 256>>>(out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    double lane_ops = (double)blocks * 256 * iters;
    printf("%-28s %8.3f ms  %8.2f G lane-ops/s  (%.2f clk/lane/CU at 2.4GHz)\n", name, ms, lane_ops / ms / 1e6,
           2.4e9 * 256 / (lane_ops / (ms * 1e-3)));
    hipFree(out);
}

int main() {
    run<float, 0>("ds atomic add f32");
    run<unsigned, 0>("ds atomic add u32");
    run<unsigned long long, 0>("ds atomic add u64");
    run<float, 1>("ds plain rmw f32");
    return 0;
}
