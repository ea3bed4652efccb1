// Calibration 4: the 9-read / 4-write float4 stream with nontemporal stores and / or loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Ptrs { float4 *p[16]; };
constexpr int NR = 9, NW = 4;
typedef float v4f __attribute__((ext_vector_type(4)));

template <int NTL, int NTS>
__global__ __launch_bounds__(256) void k(Ptrs a, int Ny4, int Nx, int nf) {
    const int QR = 16, TC = 16;
    const int tiles_r = (Ny4 + QR - 1) / QR, tiles_c = (Nx + TC - 1) / TC, tpf = tiles_r * tiles_c;
    const int b = blockIdx.x, x = b & 7, s = b >> 3, q = (tpf + 7) >> 3, lo = x * q, cnt = min(q, tpf - lo);
    if (cnt <= 0 || s >= cnt * nf) return;
    const int t = s / cnt, tile = lo + (s - t * cnt);
    const int tr = tile % tiles_r, tc = tile / tiles_r;
    const int k = threadIdx.x;
    const int rq = tr * QR + k % QR, c = tc * TC + k / QR;
    if (rq >= Ny4 || c >= Nx) return;
    const size_t qi = (size_t)t * Ny4 * Nx + (size_t)c * Ny4 + rq;
    v4f sum = {0, 0, 0, 0}, v[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const v4f *src = reinterpret_cast<const v4f *>(a.p[i]) + qi;
        v[i] = NTL ? __builtin_nontemporal_load(src) : *src;
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) sum += v[i];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        v4f *dst = reinterpret_cast<v4f *>(a.p[NR + i]) + qi;
        if (NTS) __builtin_nontemporal_store(sum, dst); else *dst = sum;
        sum.x += 1.f;
    }
}

template <int NTL, int NTS>
int run(Ptrs a, int Ny4, int Nx, int nf) {
    const int tpf = ((Ny4 + 15) / 16) * ((Nx + 15) / 16);
    unsigned grid = 8u * ((tpf + 7) / 8) * nf;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) k<NTL, NTS><<<grid, 256>>>(a, Ny4, Nx, nf);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) k<NTL, NTS><<<grid, 256>>>(a, Ny4, Nx, nf);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    printf("nt loads %d  nt stores %d  %7.3f ms  %5.2f TB/s\n", NTL, NTS, ms, (double)Ny4 * Nx * nf * 16 * (NR + NW) / ms / 1e9);
    return 0;
}

int main() {
    const int Ny4 = 270, Nx = 1920, nf = 30;
    const size_t nquads = (size_t)Ny4 * Nx * nf;
    Ptrs a;
    for (int i = 0; i < 13; ++i) { CK(hipMalloc((void **)&a.p[i], nquads * 16)); CK(hipMemset(a.p[i], 0, nquads * 16)); }
    for (int rep = 0; rep < 2; ++rep) {
        if (run<0, 0>(a, Ny4, Nx, nf) || run<0, 1>(a, Ny4, Nx, nf) || run<1, 0>(a, Ny4, Nx, nf) || run<1, 1>(a, Ny4, Nx, nf)) return 1;
    }
    return 0;
}
