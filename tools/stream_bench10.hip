// Calibration 10: a dual sweep and a primal sweep inside ONE launch, the primal work item of a tile
// following the dual work item of the same tile of the same frame at a lag of G tiles, so that the six
// dual planes the primal sweep reads are still near (Infinity Cache) instead of coming from HBM a
// second time.  Legal for the 'backward' flow direction: within a frame every coupling is local (3 x 3
// tiles), and the warp couples frame t only with frame t-1, which is a whole frame behind.
//
// The model moves the bytes of the two sweeps (row-quad float4 accesses, 64 x 16 tiles, the XCD-class
// block walk) without their arithmetic:
//   dual-like item   reads xbar_u, xbar_w of frames t and t+1, the flow, six dual planes; writes the six
//   primal-like item reads u, w, six dual planes, the flow, the temporal duals of frame t-1; writes u, w, xbar_u, xbar_w
// Variants: two launches with plain accesses (what the library does), and one launch in which
// work items are taken in ticket order per XCD class, a dual-like item publishes its six planes with
// write-through (sc1) stores and raises a per-tile flag, and a primal-like item polls the flags of
// its 3 x 3 tile neighbourhood and reads the dual planes with sc1 loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int NP = 12;      // u w ub wb g1t g1x g1y g2t g2x g2y vx vy
struct Planes { float *p[NP]; };
struct Geo { int Ny, Nx, N, tiles_r, tiles_c, ntiles, L; };

__device__ __forceinline__ float4 ldp(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float4 ld_sc1(const float *base, size_t off) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base) + off, 0, 16, 0x00020000);
    v4i v = __builtin_amdgcn_raw_buffer_load_b128(r, 0, 0, 16);
    return make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
}
__device__ __forceinline__ void st_sc1(float *base, size_t off, float4 f) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base + off, 0, 16, 0x00020000);
    v4i v = {__float_as_int(f.x), __float_as_int(f.y), __float_as_int(f.z), __float_as_int(f.w)};
    __builtin_amdgcn_raw_buffer_store_b128(v, r, 0, 0, 16);
}
__device__ __forceinline__ void acc(float4 &s, float4 v) { s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }

template <bool SC1>
__device__ __forceinline__ void dual_item(const Geo &g, const Planes &a, int t, int tile) {
    const int tc = tile / g.tiles_r, tr = tile - tc * g.tiles_r;
    const int r = tr * 64 + 4 * (threadIdx.x & 15), c = tc * 16 + (threadIdx.x >> 4);
    if (r >= g.Ny || c >= g.Nx) return;
    const size_t j = (size_t)(t + 1) * g.N + r + (size_t)g.Ny * c;
    float4 s = make_float4(0, 0, 0, 0);
    acc(s, ldp(a.p[2] + j)); acc(s, ldp(a.p[3] + j));
    acc(s, ldp(a.p[2] + j + g.N)); acc(s, ldp(a.p[3] + j + g.N));
    acc(s, ldp(a.p[10] + j)); acc(s, ldp(a.p[11] + j));
    float4 o[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) o[i] = ldp(a.p[4 + i] + j);
    const float z = (s.x != s.x) ? 1.f : 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        o[i].x += z;
        if (SC1) st_sc1(a.p[4 + i], j, o[i]); else *reinterpret_cast<float4 *>(a.p[4 + i] + j) = o[i];
    }
}

template <bool SC1>
__device__ __forceinline__ void primal_item(const Geo &g, const Planes &a, int t, int tile) {
    const int tc = tile / g.tiles_r, tr = tile - tc * g.tiles_r;
    const int r = tr * 64 + 4 * (threadIdx.x & 15), c = tc * 16 + (threadIdx.x >> 4);
    if (r >= g.Ny || c >= g.Nx) return;
    const size_t j = (size_t)(t + 1) * g.N + r + (size_t)g.Ny * c;
    float4 s = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 6; ++i) acc(s, SC1 ? ld_sc1(a.p[4 + i], j) : ldp(a.p[4 + i] + j));
    // temporal duals and flow of the previous pair (the scatter's sources)
    acc(s, SC1 ? ld_sc1(a.p[4], j - g.N) : ldp(a.p[4] + j - g.N));
    acc(s, SC1 ? ld_sc1(a.p[7], j - g.N) : ldp(a.p[7] + j - g.N));
    acc(s, ldp(a.p[10] + j - g.N)); acc(s, ldp(a.p[11] + j - g.N));
    float4 u = ldp(a.p[0] + j), w = ldp(a.p[1] + j);
    const float z = (s.x != s.x) ? 1.f : 0.f;
    u.x += z; w.x += z;
    *reinterpret_cast<float4 *>(a.p[0] + j) = u;
    *reinterpret_cast<float4 *>(a.p[1] + j) = w;
    *reinterpret_cast<float4 *>(a.p[2] + j) = u;
    *reinterpret_cast<float4 *>(a.p[3] + j) = w;
}

__device__ __forceinline__ bool class_range(const Geo &g, int x, int &lo, int &cnt) {
    const int q = (g.ntiles + 7) >> 3;
    lo = x * q; cnt = min(q, g.ntiles - lo);
    return cnt > 0;
}

// two-launch form: the block walk of the library's sweeps
template <bool DUAL>
__global__ __launch_bounds__(256) void k_sweep(Geo g, Planes a) {
    const int x = blockIdx.x & 7; int s = blockIdx.x >> 3, lo, cnt;
    if (!class_range(g, x, lo, cnt) || s >= cnt * g.L) return;
    if (DUAL) s = cnt * g.L - 1 - s;
    const int t = s / cnt, tile = lo + s % cnt;
    if (DUAL) dual_item<false>(g, a, t, tile); else primal_item<false>(g, a, t, tile);
}

// one launch: per class, frame after frame, dual item of tile i+G then primal item of tile i
template <bool SC1S, bool SC1L, bool FLAGS, bool FENCE = false>
__global__ __launch_bounds__(256) void k_fused(Geo g, Planes a, int *tickets, int *flags, int epoch, int G, int *err, int ticket_base) {
    __shared__ int sh_s;
    const int x = blockIdx.x & 7; int lo, cnt;
    if (!class_range(g, x, lo, cnt)) return;
    if (threadIdx.x == 0) sh_s = atomicAdd(&tickets[x * 64], 1) - ticket_base;
    __syncthreads();
    const int s = sh_s;
    if (s >= 2 * cnt * g.L) return;
    const int t = s / (2 * cnt), rem = s - t * 2 * cnt;
    const int Gc = min(G, cnt);
    // positions: [0,Gc) dual 0..Gc-1; then pairs (dual Gc+i, primal i) for i < cnt-Gc; then primal cnt-Gc..cnt-1
    bool is_dual; int idx;
    if (rem < Gc) { is_dual = true; idx = rem; }
    else if (rem < Gc + 2 * (cnt - Gc)) { const int k = rem - Gc; is_dual = (k & 1) == 0; idx = is_dual ? Gc + (k >> 1) : (k >> 1); }
    else { is_dual = false; idx = cnt - Gc + (rem - Gc - 2 * (cnt - Gc)); }
    if (is_dual) {
        const int tile = lo + idx;
        dual_item<SC1S>(g, a, t, tile);
        if (FLAGS) __builtin_amdgcn_s_waitcnt(0);          // this wave's stores have left
        if (FLAGS) __syncthreads();
        if (FENCE && threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (FLAGS && threadIdx.x == 0) __hip_atomic_store(&flags[t * g.ntiles + tile], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        // the first tiles_r + 2 tiles of the range border on the end of the previous class: take them last
        const int S = min(g.tiles_r + 2, cnt);
        const int tile = lo + (idx + S) % cnt;
        const int tc = tile / g.tiles_r, tr = tile - tc * g.tiles_r;
        if (FLAGS && threadIdx.x < 9) {
            const int dr = (int)threadIdx.x % 3 - 1, dc = (int)threadIdx.x / 3 - 1;
            const int nr = tr + dr, nc = tc + dc;
            if (nr >= 0 && nr < g.tiles_r && nc >= 0 && nc < g.tiles_c) {
                const int *f = &flags[t * g.ntiles + nr + g.tiles_r * nc];
                int spins = 0;
                while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > (1 << 22)) { atomicAdd(err, 1); break; }
                }
            }
        }
        if (FENCE && threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (FLAGS) __syncthreads();
        primal_item<SC1L>(g, a, t, tile);
    }
}

int main(int argc, char **argv) {
    Geo g; g.Ny = 1080; g.Nx = 1920; g.N = g.Ny * g.Nx; g.L = argc > 1 ? atoi(argv[1]) : 30;
    g.tiles_r = (g.Ny + 63) / 64; g.tiles_c = (g.Nx + 15) / 16; g.ntiles = g.tiles_r * g.tiles_c;
    const size_t plane = (size_t)(g.L + 2) * g.N * sizeof(float);
    Planes a;
    char *base; const size_t stride = plane + (16u << 20);
    CK(hipMalloc((void **)&base, NP * stride));
    CK(hipMemset(base, 0, NP * stride));
    for (int i = 0; i < NP; ++i) a.p[i] = reinterpret_cast<float *>(base + i * stride + (size_t)i * (320u << 10));
    int *tickets, *flags, *err;
    CK(hipMalloc((void **)&tickets, 8 * 64 * sizeof(int))); CK(hipMemset(tickets, 0, 8 * 64 * sizeof(int)));
    CK(hipMalloc((void **)&flags, (size_t)g.L * g.ntiles * sizeof(int))); CK(hipMemset(flags, 0, (size_t)g.L * g.ntiles * sizeof(int)));
    CK(hipMalloc((void **)&err, sizeof(int))); CK(hipMemset(err, 0, sizeof(int)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned grid1 = 8u * ((g.ntiles + 7) / 8) * g.L, grid2 = 2 * grid1;
    const int q = (g.ntiles + 7) / 8;
    const double bytes = (double)g.N * g.L * 4.0 * (14 + 4 + 12 + 6);     // moved per iteration, second uses included
    const int reps = 40;
    for (int round = 0; round < 2; ++round) {
        for (int i = 0; i < 5 + reps; ++i) {
            if (i == 5) CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_sweep<false>, dim3(grid1), dim3(256), 0, 0, g, a);
            hipLaunchKernelGGL(k_sweep<true>, dim3(grid1), dim3(256), 0, 0, g, a);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("two launches, plain accesses        : %7.4f ms per iteration  %5.2f TB/s moved\n", ms, bytes / ms / 1e9);
        int epoch = 0, launches = 0;
        for (int variant : {0, 1, 5}) {
          for (int G : {40, 96, 128, 160, 192, 224, q}) {
            for (int i = 0; i < 5 + reps; ++i) {
                if (i == 5) CK(hipEventRecord(e0));
                ++epoch;
                const int ep = epoch + 100000 * round, tb = launches * (int)(grid2 / 8);
                switch (variant) {
                case 0: hipLaunchKernelGGL((k_fused<false, false, false>), dim3(grid2), dim3(256), 0, 0, g, a, tickets, flags, ep, G, err, tb); break;
                case 1: hipLaunchKernelGGL((k_fused<false, false, true>), dim3(grid2), dim3(256), 0, 0, g, a, tickets, flags, ep, G, err, tb); break;
                case 2: hipLaunchKernelGGL((k_fused<true, false, true>), dim3(grid2), dim3(256), 0, 0, g, a, tickets, flags, ep, G, err, tb); break;
                case 3: hipLaunchKernelGGL((k_fused<false, true, true>), dim3(grid2), dim3(256), 0, 0, g, a, tickets, flags, ep, G, err, tb); break;
                case 5: hipLaunchKernelGGL((k_fused<false, false, true, true>), dim3(grid2), dim3(256), 0, 0, g, a, tickets, flags, ep, G, err, tb); break;
                default: hipLaunchKernelGGL((k_fused<true, true, true>), dim3(grid2), dim3(256), 0, 0, g, a, tickets, flags, ep, G, err, tb); break;
                }
                ++launches;
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
            int herr; CK(hipMemcpy(&herr, err, sizeof(int), hipMemcpyDeviceToHost));
            static const char *names[6] = {"tickets only, plain", "+ flags and polls", "+ sc1 stores", "+ sc1 loads (plain stores)", "+ sc1 stores and loads", "+ flags, release/acquire fences"};
            printf("one launch, lag %4d tiles, %-28s: %7.4f ms per iteration  %5.2f TB/s moved  (timeouts %d)\n", G, names[variant], ms, bytes / ms / 1e9, herr);
            if (herr) return 2;
          }
        }
        CK(hipMemset(tickets, 0, 8 * 64 * sizeof(int)));
    }
    return 0;
}
