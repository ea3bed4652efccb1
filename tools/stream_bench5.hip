// Calibration 5: the 9-read / 4-write float4 stream walked "chunk of S tiles through all frames,
// then the next chunk" (what would let a tile's data be reused across frames in L2), with the
// frame stride padded by `pad` quads, against the frame-major walk.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Ptrs { float4 *p[16]; };
constexpr int NR = 9, NW = 4;

__global__ __launch_bounds__(256) void k(Ptrs a, int Ny4, int Nx, int nf, int S, size_t fstride) {
    const int QR = 16, TC = 16;
    const int tiles_r = (Ny4 + QR - 1) / QR, tiles_c = (Nx + TC - 1) / TC, tpf = tiles_r * tiles_c;
    const int b = blockIdx.x, x = b & 7, s = b >> 3, q = (tpf + 7) >> 3, lo = x * q, cnt = min(q, tpf - lo);
    if (cnt <= 0 || s >= cnt * nf) return;
    int t, tile;
    if (S <= 0) { t = s / cnt; tile = lo + (s - t * cnt); }
    else {
        const int per = S * nf, grp = s / per, rem = s - grp * per, sg = min(S, cnt - grp * S);
        t = rem / sg; tile = lo + grp * S + (rem - t * sg);
    }
    const int tr = tile % tiles_r, tc = tile / tiles_r;
    const int kk = threadIdx.x;
    const int rq = tr * QR + kk % QR, c = tc * TC + kk / QR;
    if (rq >= Ny4 || c >= Nx) return;
    const size_t qi = (size_t)t * fstride + (size_t)c * Ny4 + rq;
    float4 sum = make_float4(0, 0, 0, 0), v[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) v[i] = a.p[i][qi];
#pragma unroll
    for (int i = 0; i < NR; ++i) { sum.x += v[i].x; sum.y += v[i].y; sum.z += v[i].z; sum.w += v[i].w; }
#pragma unroll
    for (int i = 0; i < NW; ++i) { a.p[NR + i][qi] = sum; sum.x += 1.f; }
}

int run(Ptrs a, int Ny4, int Nx, int nf, int S, size_t pad) {
    const int tpf = ((Ny4 + 15) / 16) * ((Nx + 15) / 16);
    unsigned grid = 8u * ((tpf + 7) / 8) * nf;
    const size_t fstride = (size_t)Ny4 * Nx + pad;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) k<<<grid, 256>>>(a, Ny4, Nx, nf, S, fstride);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) k<<<grid, 256>>>(a, Ny4, Nx, nf, S, fstride);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    printf("S=%4d pad=%6zu quads  %7.3f ms  %5.2f TB/s\n", S, pad, ms, (double)Ny4 * Nx * nf * 16 * (NR + NW) / ms / 1e9);
    return 0;
}

int main() {
    const int Ny4 = 270, Nx = 1920, nf = 30;
    const size_t maxpad = 70000;
    const size_t nquads = ((size_t)Ny4 * Nx + maxpad) * nf;
    Ptrs a;
    for (int i = 0; i < 13; ++i) { CK(hipMalloc((void **)&a.p[i], nquads * 16)); CK(hipMemset(a.p[i], 0, nquads * 16)); }
    const size_t pads[] = {0, 16, 64, 272, 1040, 4112, 16400, 65552, 1111, 33333};
    if (run(a, Ny4, Nx, nf, 0, 0)) return 1;
    for (int S : {1, 8, 32})
        for (size_t pad : pads) if (run(a, Ny4, Nx, nf, S, pad)) return 1;
    if (run(a, Ny4, Nx, nf, 0, 0)) return 1;
    return 0;
}
