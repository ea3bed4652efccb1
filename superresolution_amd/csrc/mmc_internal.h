// Internal interface between the C-ABI host code (mmc_capi.cpp) and the
// gfx950 kernels (mmc_kernels.hip).  Not installed; include/mmc_hip.h is the
// public header.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

enum { MODEL_IC = 0, MODEL_K1 = 1, MODEL_ADD = 2 };
// MODE_STEPS_UPD (primal sweep only): the step sizes exactly as MODE_UPDATE derives them (column sums of W
// from the scatter and the identity bits), where MODE_STEPS reads the cW plane written by mmc_init.
// MODE_ENERGY (dual sweep, evaluated at the iterate): primal energy, |K x|^2 and the dual optimality residual.
// MODE_DUALOBJ (primal sweep): the K'y terms of the dual objective, |K'y|^2 and the primal optimality residual;
// writes nothing.
enum { MODE_UPDATE = 0, MODE_RAW = 1, MODE_STEPS = 2, MODE_ENERGY = 3, MODE_STEPS_UPD = 4, MODE_DUALOBJ = 5 };

// One workgroup of 256 threads owns a TILE_R x TILE_C pixel tile of one frame.
// Rows are the contiguous axis (column-major frames), so one 64-lane wave
// covers 64 consecutive rows of a column.
#ifndef MMC_TILE_R
#define MMC_TILE_R 64
#endif
#ifndef MMC_TILE_C
#define MMC_TILE_C 16
#endif
#ifndef MMC_NTHREADS
#define MMC_NTHREADS 256
#endif
enum { TILE_R = MMC_TILE_R, TILE_C = MMC_TILE_C, MMC_THREADS = MMC_NTHREADS };
static_assert(TILE_R * TILE_C == 4 * MMC_THREADS && (TILE_R & (TILE_R - 1)) == 0, "tile must hold one row quad per thread");
enum { MMC_MAX_CK = 40 };      // max side of the composite (sampling * blur) kernel
enum { MMC_WARP_MAX = 2304 }; // floats of the warped frame a tile may stage in LDS per plane
// Regular part of a tile's flow: displacements within these margins of the tile's mean.  The tap
// region of the dual sweep must fit MMC_WARP_MAX: (TILE_R + 1 + 2*MR + 6 alignment + 1 pitch) x (TILE_C + 1 + 2*MC).
enum { MMC_TAP_MR = 4, MMC_TAP_MC = 5, MMC_SRC_MR = 4, MMC_SRC_MC = 4 };
static_assert((MMC_TILE_R + 1 + 2 * MMC_TAP_MR + 6 + 1) * (MMC_TILE_C + 1 + 2 * MMC_TAP_MC) <= MMC_WARP_MAX, "tap region must fit its LDS buffer");
enum { MMC_SUM_SLOTS = 64 };   // partial-sum slots per reduced quantity
// Reduced quantities kept on the device, MMC_SUM_SLOTS partial sums each:
//   0..3   sum (x+ - x)^2, sum (x+)^2, sum (y+ - y)^2, sum (y+)^2          (update sweeps with reduce)
//   4..7   regulariser energy, data energy, |K x|^2, dual optimality residual^2   (dual sweep, MODE_ENERGY)
//   8, 9   sum |W u0|, sum |grad u0|                                       (h)
//   10..13 sum max(0, -(K'y)_u), sum |(K'y)_w|, |K'y|^2, primal optimality residual^2   (primal sweep, MODE_DUALOBJ)
//   14     <f, p> (+ <u0, q1> + <w0, q2>)                                  (k_dual_linear)
//   15, 16 sum ((x+ - x)/tau)^2, sum ((y+ - y)/sigma)^2                     (update sweeps with reduce; step balancing)
enum { MMC_NUM_SUMS_INT = 18 };
enum { SUM_RES = 0, SUM_ENERGY = 4, SUM_H = 8, SUM_DUALOBJ = 10, SUM_DUALLIN = 14, SUM_WRES = 15 };
// MMC_STATIC_MAX_BLOCKS (public header): workgroups of k_h_partials that mmc_h_partials_static has room for

// Division of 0 <= n < 2^31 by a fixed d through one high multiply: n / d = umulhi(n, m) >> s
// with m = ceil(2^(31+l) / d), s = l - 1, l = ceil(log2 d); d = 1 is flagged by m = 0.
struct FastDiv { unsigned m, s; };
static inline FastDiv make_fastdiv(unsigned d) {
    FastDiv r = {0u, 0u};
    if (d <= 1u) return r;
    unsigned l = 0;
    while ((1u << l) < d) ++l;
    r.m = (unsigned)((((uint64_t)1 << (31 + l)) + d - 1) / d);
    r.s = l - 1;
    return r;
}

#ifndef MMC_SEPARABLE
#define MMC_SEPARABLE 1     // 0: never split a separable composite kernel into its two factors
#endif
#ifndef MMC_SEP_MIN_TAPS4
#define MMC_SEP_MIN_TAPS4 10  // the same at zoom 4, whose one-pass form has 3 x 3 low-res taps unrolled with scalar weights
#endif
#ifndef MMC_SEP_MIN_TAPS
#define MMC_SEP_MIN_TAPS 9    // zoom factors other than 4: low-res taps (Jr x Jc) from which a separable kernel takes two passes
#endif
#define MMC_KSEP_PITCH (MMC_MAX_CK + 8)       // floats between the three vectors of Arrays::ksep
// identity bits (Arrays::idq): one byte per row quad, ceil(Ny / 4) quads per column, column after column
#define MMC_IDQ_PER_FRAME(g) ((size_t)(((g).Ny + 3) >> 2) * (size_t)(g).Nx)

struct Geom {
    int Ny, Nx, N;             // high-resolution frame
    int ny, nx, n;             // low-resolution frame
    int f;                     // zoom factor
    int L;                     // local frames
    int frame_begin, nc_global;
    int dir;                   // mmc_flow_direction
    int model;                 // MODEL_*
    int CH, CW, oy, ox;        // composite kernel size; first tap sits at f*R+oy, f*C+ox
    int Jr, Jc;                // low-res taps per axis of the adjoint: ceil(CH/f), ceil(CW/f)
    int MP, NP;                // zero-padded composite kernel table: f*Jr x f*Jc
    int PR, PC;                // low-res window staged per tile
    int wpad, SRp, MPq;        // dual sweep, zoom 4: rows added above the xbar window so that it starts on a row quad,
                               // its padded height (multiple of 4), and the height of the kernel table shifted by wpad
    int tiles_r, tiles_c, ntiles;
    FastDiv dv_tiles_r;        // tile -> (tile row, tile column)
    FastDiv dv_Ny;             // pixel index -> column
    FastDiv dv_q, dv_qlast;    // tiles per XCD class ((ntiles+7)/8) and of the last non-empty class
    FastDiv dv_PR;
    // Frames larger than about 2 Mpx: each XCD class walks its tile range in sub-ranges of `chunk`
    // tiles, every sub-range through all frames of the launch before the next one (0: one range).
    int chunk;
    FastDiv dv_chunk, dv_per;  // dv_per: chunk * (frames of this launch), set by the launcher
    float alpha, kappa, inv_h;
    float tau_s, sigma_s;      // step balance: every tau is multiplied by tau_s = s, every sigma by sigma_s = 1/s (1, 1 by default)
    float wsgn;                // sign of w in the temporal rows of K1: -1 (u - w, MultiframeMotionCoupling.m:490-491) or +1 (u + w, :864)
    float radius1, radius2;    // projection radii of the g1 / g2 groups
    // Away from the frame border the column abs-sum of A repeats with period f in both directions.
    // cA_margin > 0: pixels at least that far from every border take it from cA_tab (f = 4 only)
    // instead of reading the cA plane.
    int cA_margin;
    // E'p (ucorr) is zero at pixels at least this far from every border (0: nowhere known to be zero)
    int uc_margin;
    // 1: zoom 4, the composite kernel is an outer product and reaches beyond 3 x 3 low-res taps: the sweeps apply
    // its two factors one after the other (a.ksep)
    int sep;
    float cA_tab[16];          // [row mod 4 + 4 * (column mod 4)]
    int fixed;                 // 1: first frame of the sequence is pinned to (u0, w0), MultiframeMotionCoupling.m:535-572
    int dbg;                   // ablation switches for kernel tuning (env MMC_DEBUG); read only in -DMMC_ABLATION builds
};

// Every per-pixel array holds L+2 frames: slot 0 is the ghost of the previous
// block's last frame, slots 1..L the local frames, slot L+1 the ghost of the
// next block's first frame.
struct Arrays {
    float *u, *w;              // primal iterate
    float *ub, *wb;            // xbar = 2 x+ - x
    float *g1t, *g1x, *g1y;    // dual of K1 (temporal, d/dcol, d/drow)
    float *g2t, *g2x, *g2y;    // dual of K2
    float *vx, *vy;            // flow of pair t (frame t <-> t+1); NaN = masked
    float *cW;                 // column abs-sum of the motion-coupling operator
    uint8_t *idq;              // per local frame and row quad: 2 bits per row = number of unmasked -Id entries of W in that column (the part of cW that is not scattered)
    float *q1, *q2;            // duals of the first-frame constraint u_1 = u0, w_1 = w0 (one frame each)
    float *u0, *w0;            // the pinned first frame
    float *p, *f;              // data-term dual and low-res data, L*n
    float *cA;                 // column abs-sum of D*B, N
    float *sigp;               // 1 / row abs-sum of D*B, n
    float *cker;               // composite kernel, CH x CW column-major
    float *ckp;                // the same, zero-padded to MP x NP
    float *ckq;                // the same, wpad zero rows on top, zero-padded to MPq x CW
    // Geom::sep: the composite kernel is the outer product ky kx' (separable blur and sampler):
    float *ksep;               // ky zero-padded to MP | kx zero-padded to NP | ky with wpad zero rows on top, padded to MPq; MMC_KSEP_PITCH apart
    // Sparse correction E = A - (composite kernel with one replicate clamp), non-zero only on the low-res
    // pixels whose sampling rows are not the interior ones; all null when the sampler has none.
    int ne, net;               // rows of E (low-res pixels) and of E' (high-res pixels)
    int *e_row, *e_ptr, *e_col;   float *e_val;
    int *et_row, *et_ptr, *et_col; float *et_val;
    float *pcorr;              // E xbar_u, L*n (zero off the rows of E)
    float *ucorr;              // E' p, L*N (zero off the rows of E')
    // Sources of a target tile in the adjoint of the warp: a rectangle plus a list of the stragglers
    int *bbox;                 // [(L+1) pair slots][ntiles][4] = rmin,rmax,cmin,cmax of the source rectangle
    int *ovf_ptr;              // [(L+1)*ntiles + 1] offsets into ovf_idx
    int *ovf_idx;              // pixel indices of the sources outside their tile's rectangle
    unsigned long long *bstat; // [(L+1)][ntiles][3] sum of source-minus-tap rows, columns, source count (set-up only)
    int *tbox;                 // [L pair slots][ntiles][4] = rows/cols of the warped frame staged for a tile's TAPS
    double *sums;              // [MMC_NUM_SUMS_INT][MMC_SUM_SLOTS]
};

// Launchers (all asynchronous on `s`; return hipGetLastError()).
hipError_t mmc_launch_primal(const Geom &g, const Arrays &a, int t0, int t1, int mode, bool reduce, hipStream_t s);
hipError_t mmc_launch_dual(const Geom &g, const Arrays &a, int t0, int t1, int mode, bool reduce, hipStream_t s);
// `frame_words`: N 8-byte words of device memory
hipError_t mmc_launch_init_warp(const Geom &g, const Arrays &a, void *frame_words, hipStream_t s);
// counts (fill = false) or places (fill = true) the sources outside their tile's rectangle; cursor: (L+1)*ntiles ints
hipError_t mmc_launch_init_overflow(const Geom &g, const Arrays &a, bool fill, int *cursor, hipStream_t s);
hipError_t mmc_launch_sanitize(const Geom &g, const Arrays &a, hipStream_t s);
hipError_t mmc_launch_fold_occ(const Geom &g, float *vx, const uint8_t *occ, size_t count, hipStream_t s);
// Bicubic start point (MultiframeMotionCoupling.m:415-419): per output phase (index mod zoom) the six
// normalised Keys weights and the offset of the first tap relative to (index / zoom).
enum { MMC_BICUBIC_MAX_ZOOM = 8, MMC_BICUBIC_TAPS = 6 };
struct BicubicTab { float w[MMC_BICUBIC_MAX_ZOOM][MMC_BICUBIC_TAPS]; int first[MMC_BICUBIC_MAX_ZOOM]; };
// u, ub (and w, wb when the model has w) of the local frames <- imresize(f, zoom, 'bicubic')
hipError_t mmc_launch_bicubic_start(const Geom &g, const Arrays &a, const BicubicTab &tab, hipStream_t s);
hipError_t mmc_launch_h_partials(const Geom &g, const Arrays &a, hipStream_t s);
// sums[SUM_DUALLIN] += <f, p> over the local frames (+ <u0, q1> + <w0, q2> on the block that holds the pinned frame)
hipError_t mmc_launch_dual_linear(const Geom &g, const Arrays &a, hipStream_t s);
// Streams through 12 per-pixel planes (L+2 frame slots each) with the block walk and access shape of
// the sweeps and leaves their contents unchanged; times how a placement of the planes performs.
hipError_t mmc_launch_stream_probe(const Geom &g, float *const planes[12], int nf, bool dual_like, hipStream_t s);

// ---- classical single-frame solve (mmc_sf.cpp, mmc_sf_kernels.hip) ------------------------------
// Geometry and the data operator D*B (composite kernel, its column sums) come from a one-frame
// mmc_handle that the single-frame handle owns.
struct SfArrays {
    int nf;                      // low-resolution frames = data terms
    float alpha, delta;          // TV weight (ball radius of q), Huber parameter of the data terms
    float *u, *ub;               // primal iterate and 2u+ - u, N each
    float *qx, *qy;              // dual of the gradient (d/dcol, d/drow), N each
    float *vx, *vy;              // flow from the central frame to frame i, nf*N each; NaN = masked row
    float *p, *f, *sigp;         // data-term duals, low-res data, dual steps, nf*n each
    float *tau;                  // primal steps, N
    float *wimg;                 // W_i ub for every frame, nf*N (scratch of the dual half-iteration)
    unsigned long long *acc;     // W_i' A' p_i of the irregular frames in 64-bit fixed point (scale 2^40), N; zero between iterations
    int *bbox;                   // [nf][ntiles][4] = rmin,rmax,cmin,cmax of the source pixels with a tap in the tile (64 x 16 tiles)
    int *irregular;              // [nf] 1: some tile's source rectangle of this frame is too large to walk (violent flow);
                                 //      the frame is scattered with global atomics instead
    int any_irregular;
    const float *cker, *cA;      // composite kernel CH x CW and absolute column sums of D*B (of the geometry handle)
    float *ck12;                 // the composite kernel zero-padded to 12 x 12 (zoom 4, kernels up to 12 x 12), else unused
    int ck12_ok;
    double *sums;                // [MMC_NUM_SUMS_INT][MMC_SUM_SLOTS]
};
// dual half-iteration: W_i ub, p and q updates; primal half-iteration: scatter and u update.
// MODE_RAW applies K to ub (results in p, qx, qy) resp. K' to (p, q) (result in ub); MODE_STEPS
// writes the dual steps of the data rows to sigp resp. the primal steps to tau and touches nothing else.
hipError_t mmc_sf_launch_dual(const Geom &g, const SfArrays &a, int mode, bool reduce, hipStream_t s);
hipError_t mmc_sf_launch_primal(const Geom &g, const SfArrays &a, int mode, bool reduce, hipStream_t s);
// source rectangles of every (frame, tile) from the flows: a.bbox
hipError_t mmc_sf_launch_bbox(const Geom &g, const SfArrays &a, hipStream_t s);
enum { MMC_SF_BOX_MAX = 6144 };   // largest source rectangle (pixels) the tile kernel walks

// Failure of an entry point that has no handle and must not allocate: `literal` (a string literal)
// becomes what mmc_last_error(NULL) returns on this thread; returns `code`.
int mmc_static_fail(int code, const char *literal);

// Streams and events of handles and groups come from a per-device pool that lives as long as the
// process: a destroyed handle gives them back instead of destroying them, so a session that builds
// one solver per batch creates each runtime object once.  `device` must be the current device on
// acquire.  A released stream must be idle (synchronised); a released event may have been recorded.
// timing = false: hipEventDisableTiming.  Null on failure (message via hipGetLastError).
hipStream_t mmc_pool_stream_acquire(int device);
void mmc_pool_stream_release(int device, hipStream_t s);
hipEvent_t mmc_pool_event_acquire(int device, bool timing);
void mmc_pool_event_release(int device, bool timing, hipEvent_t e);

#ifdef MMC_WATCH
// debug builds only (tools/watch/mmc_watch.c)
extern "C" int mmc_watch_arm(void *addr);
extern "C" void mmc_watch_disarm(void);
extern "C" void mmc_watch_check(const char *where);
extern "C" void mmc_canary_new(void);
extern "C" void mmc_canary_check(const char *where);
extern "C" void mmc_canary_free(void);
#define WCHK(where) mmc_canary_check(where)
#else
#define WCHK(where) ((void)0)
#endif

// Used by the group driver (mmc_group.cpp), which owns one handle per frame block.
struct mmc_handle;
hipStream_t mmc_internal_stream(const mmc_handle *h);
// Geometry and device arrays of a handle; the data operator (composite kernel, cA, sigp) is built on demand.
const Geom *mmc_internal_geom(const mmc_handle *h);
const Arrays *mmc_internal_arrays(const mmc_handle *h);
int mmc_internal_build_data_operator(mmc_handle *h);
// A handle with the geometry and the data operator of a one-frame block and none of the per-pixel
// state (validates the parameters as mmc_create does); only mmc_set_blur, mmc_internal_* and
// mmc_destroy may be called on it.
int mmc_internal_create_geometry(const struct mmc_params *p, mmc_handle **out);
int mmc_internal_set_flow_planes(mmc_handle *h, const float *vcol, const float *vrow, const uint8_t *occ);
