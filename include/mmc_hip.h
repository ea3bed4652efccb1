/*
 * mmc_hip.h -- C ABI of libmmc_hip.so: the super-resolution (PDHG) solve of
 * Multiframe Motion Coupling on AMD MI355X (gfx950).
 *
 * The library replaces, for one contiguous block of frames, what the reference
 * hands to prost / flexBox inside MultiframeMotionCoupling.m:
 *   problem assembly   init_u_prost            MultiframeMotionCoupling.m:346-574
 *   the solve          solveU                  MultiframeMotionCoupling.m:977-1004
 * All operators are matrix-free; nothing here builds a sparse matrix.
 * Three groups of entry points:
 *   mmc_*        one frame block on one GPU (a whole sequence is the block [0, nc))
 *   mmc_group_*  one sequence as several blocks, one per GPU, driven by the calling thread
 *   mmc_sf_*     the reference's comparison methods: one high-resolution frame with
 *                motion-compensated data terms (classical methods/singleframeMotionSR_*_prost.m)
 *
 * Conventions
 *   - Plain C, no C++ types.  Every function returns MMC_OK (0) or a negative
 *     mmc_status; mmc_last_error() gives the message.  No exception or abort
 *     crosses this boundary.
 *   - All image-like host buffers are float32 in MATLAB column-major order
 *     (row index fastest): idx = r + Ny*c + Ny*Nx*t.  The gateway converts
 *     MATLAB doubles to float on the way in and back on the way out.
 *   - Ny = factor*ny, Nx = factor*nx are the high-resolution frame sizes.
 *   - A handle owns all device memory of one frame block [frame_begin,
 *     frame_end) of a sequence of nc_global frames, plus one ghost frame at
 *     either end when the block has a neighbour.  Calls on one handle must come
 *     from one thread at a time.  All work is queued on one HIP stream.
 */
#ifndef MMC_HIP_H
#define MMC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMC_ABI_VERSION 4

typedef enum mmc_status {
    MMC_OK = 0,
    MMC_ERR_INVALID_ARG = -1,   /* bad pointer, size or parameter combination   */
    MMC_ERR_STATE = -2,         /* call made before its prerequisites           */
    MMC_ERR_HIP = -3,           /* a HIP runtime call failed                    */
    MMC_ERR_UNSUPPORTED = -4,   /* valid in the reference, not implemented here */
    MMC_ERR_NO_DEVICE = -5      /* no usable gfx950 device                      */
} mmc_status;

/* flowDirection, MultiframeMotionCoupling.m:235, :392-398 */
typedef enum mmc_flow_direction {
    MMC_FLOW_FORWARD = 0,           /* 'forward'          -> constructWarpFF(v,'F-I') */
    MMC_FLOW_BACKWARD = 1,          /* 'backward'         -> constructWarpFF(v,'I-F') */
    MMC_FLOW_FORWARD_BACKWARD = 2   /* 'forward-backward' -> constructWarpFB(v)       */
} mmc_flow_direction;

/* interpMethod, MultiframeMotionCoupling.m:237-239, helpers/samplingOperator.m */
typedef enum mmc_sampling {
    MMC_SAMPLING_AVERAGE = 0,   /* 'average': block mean; exact only for factor 4 */
    MMC_SAMPLING_STRIDE = 1,    /* 'stride' : every factor-th pixel from top-left */
    MMC_SAMPLING_TAPS = 2       /* any separable sampler, given by mmc_set_sampling_taps */
} mmc_sampling;

/* Versioned POD.  Set size = sizeof(mmc_params), version = MMC_ABI_VERSION. */
typedef struct mmc_params {
    uint32_t size;
    uint32_t version;
    int32_t ny, nx;            /* low-resolution frame size (rows, cols)               */
    int32_t factor;            /* zoom factor, MultiframeMotionCoupling.m:126-130      */
    int32_t nc_global;         /* frames in the whole sequence                         */
    int32_t frame_begin;       /* this block is [frame_begin, frame_end), 0-based      */
    int32_t frame_end;
    float alpha;               /* MultiframeMotionCoupling.m:207                       */
    float kappa;               /* :209; NaN = additive model (:515), 1 = model :500    */
    float h;                   /* :437; <= 0: computed (see mmc_init)                  */
    int32_t flow_direction;    /* mmc_flow_direction                                   */
    int32_t sampling;          /* mmc_sampling                                         */
    int32_t device;            /* HIP device ordinal                                   */
    void *stream;              /* hipStream_t to queue on; NULL: the library makes one */
} mmc_params;

/* Result of a solve / of the last residual check.
 * The first group is reduced inside the update sweeps at every check.  The second group (the
 * "diagnostics") takes two more read-only passes over the block, about the cost of one iteration:
 * it is filled at a check when diagnostics are on (mmc_set_diagnostics) and is NaN otherwise.
 *   primal_energy  ||A u - f||_1 + alpha sum_j ||(K1(u - w))_j|| + alpha sum_j ||(K2 w)_j|| at the
 *                  iterate (the quantity prost prints, MultiframeMotionCoupling.m:357-359); the box
 *                  on u holds exactly, the equality rows of a pinned first frame are not counted
 *   dual_energy    -<f,p> - <u0,q1> - <w0,q2> - sum_j max(0, -(K'y)_u,j): the dual objective
 *                  -F*(y) - G*(-K'y) without the condition (K'y)_w = 0 that the free variable w
 *                  imposes (the duals lie inside their sets by construction)
 *   gap            primal_energy - dual_energy
 *   dual_infeasibility  sum_j |(K'y)_w,j|.  For any bound B >= max|w| at the optimum the optimal
 *                  energy lies in [dual_energy - B * dual_infeasibility, primal_energy].
 *   opt_res_primal 2-norm of the smallest element of dG(x) + K'y   (distance from primal optimality)
 *   opt_res_dual   2-norm of the smallest element of dF*(y) - K x  (distance from dual optimality)
 *   norm_KTy, norm_Kx   2-norms that scale the relative part of the stopping test
 * With diagnostics on, a solve with tol > 0 stops when
 *   opt_res_primal <= tol*sqrt(#primal) + tol*norm_KTy  and  opt_res_dual <= tol*sqrt(#dual) + tol*norm_Kx,
 * the absolute + relative residual test that the reference asks of prost with all four tolerances
 * equal to tol_pd (MultiframeMotionCoupling.m:276-280, :357-363).  With diagnostics off it stops when
 * max(res_primal, res_dual) < tol. */
typedef struct mmc_stats {
    int32_t iterations;        /* PDHG iterations executed by this call                */
    int32_t converged;         /* 1 if the tolerance test stopped the loop             */
    double res_primal;         /* ||x+ - x||_2 / max(||x+||_2, tiny) at the last check */
    double res_dual;           /* ||y+ - y||_2 / max(||y+||_2, tiny) at the last check */
    double solve_seconds;      /* host wall time of the call, device-synchronised      */
    double primal_energy, dual_energy, gap, dual_infeasibility;
    double opt_res_primal, opt_res_dual, norm_KTy, norm_Kx;
    double step_balance;       /* s of mmc_set_step_balance at the end of the call     */
} mmc_stats;

/* Raw sums behind the diagnostics of mmc_stats, for adding up over frame blocks:
 * [0] regulariser energy  [1] data energy  [2] |K x|^2  [3] dual optimality residual^2
 * [4] sum max(0,-(K'y)_u) [5] sum |(K'y)_w| [6] |K'y|^2 [7] primal optimality residual^2
 * [8] <f,p> + <u0,q1> + <w0,q2> */
#define MMC_NUM_DIAG 9
/* When the diagnostics are evaluated: never, at the checks of a solve that has a tolerance or a
 * progress callback (the default), or at every check. */
typedef enum mmc_diagnostics { MMC_DIAG_OFF = 0, MMC_DIAG_AUTO = 1, MMC_DIAG_ALWAYS = 2 } mmc_diagnostics;

/* Raw sums behind mmc_stats, for all-reducing over frame blocks:
 * [0]=sum (x+ - x)^2  [1]=sum (x+)^2  [2]=sum (y+ - y)^2  [3]=sum (y+)^2 */
#define MMC_NUM_SUMS 4

/* Progress callback, called on the calling thread after every residual check
 * (prost's 'num_cback_calls', MultiframeMotionCoupling.m:358).  Return non-zero
 * to stop the solve. */
typedef int (*mmc_progress_cb)(int iteration, const mmc_stats *stats, void *user);

/* Device addresses of the frames a block exchanges with its neighbours once per
 * sweep.  Each entry is one high-resolution frame of `frame_bytes` bytes.
 *   send_prev[2]: xbar = 2x+ - x of the FIRST local frame (u part, w part)
 *                 -> previous block, which receives it in recv_next[2]
 *   send_next[2]: temporal dual components (g1, g2) of the LAST local frame
 *                 -> next block, which receives them in recv_prev[2]
 * A pointer is NULL when the block has no neighbour on that side (or the model
 * has no w / g2). */
typedef struct mmc_halo {
    uint64_t frame_bytes;
    void *send_prev[2];
    void *recv_next[2];
    void *send_next[2];
    void *recv_prev[2];
} mmc_halo;

typedef struct mmc_handle mmc_handle;

int mmc_abi_version(void);

/* HIP release the library was compiled against and the HIP runtime it runs on in this process,
 * both encoded as major*10000000 + minor*100000 + patch (HIP_VERSION).  They differ when the
 * host program brings its own copy of the runtime under the same SONAME (PyTorch-ROCm does);
 * the pair belongs in every bug report.  Either pointer may be NULL.  No counterpart in the
 * reference (prost.list_gpus, MultiframeMotionCoupling.m:255, is its only runtime probe). */
int mmc_hip_versions(int *built_against, int *runtime);

/* Message of the last failure on `h`; with h == NULL the last mmc_create failure
 * of the calling thread. */
const char *mmc_last_error(const mmc_handle *h);

/* Validates the parameters the way MultiframeMotionCoupling.init does
 * (:228-231 positive scalars; :283-290 NaN kappa is the additive model) and
 * allocates the device state of the block.  Replaces the variable / dual-pair
 * declarations of init_u_prost, MultiframeMotionCoupling.m:445-459, :483. */
int mmc_create(const mmc_params *params, mmc_handle **out);
int mmc_destroy(mmc_handle *h);

/* Low-resolution data term f: ny x nx x (frame_end-frame_begin).
 * Replaces obj.imageSequenceSmall(:) at MultiframeMotionCoupling.m:496. */
int mmc_set_lowres(mmc_handle *h, const float *f);

/* Blur kernel k (kh x kw, column-major, both odd as helpers/RepConvMtx.m:6-8
 * demands), applied as a true convolution with replicated border; k == NULL
 * means no blur (MultiframeMotionCoupling.m:373).  Replaces RepConvMtx and its
 * composition with samplingOperator, MultiframeMotionCoupling.m:368-375. */
int mmc_set_blur(mmc_handle *h, const float *k, int kh, int kw);

/* The separable sampling operator S = kron(Sx', Sy) of helpers/samplingOperator.m:77-83 for the
 * interpMethods other than 'average'/'stride' ('nearest', 'bilinear', 'bicubic', 'lanczos2',
 * 'lanczos3', 'custom', with or without antialiasing, :99-136), handed over as its two 1-D
 * factors (Dim1Matrix, :89-146) in banded form: low-res row R takes taps_y weights
 * w_y[R*taps_y ...] from the high-res rows first_y[R], first_y[R]+1, ... (0-based; a position
 * outside the frame must carry weight 0); likewise for columns.  Rows away from the border must
 * be shifted copies of one another (they are for an integer zoom factor); the rows near the
 * border, where imresize folds the taps beyond the edge onto the edge pixel, may differ and are
 * applied exactly.  Needs params.sampling == MMC_SAMPLING_TAPS; call before mmc_init. */
int mmc_set_sampling_taps(mmc_handle *h, const int32_t *first_y, const float *w_y, int taps_y,
                          const int32_t *first_x, const float *w_x, int taps_x);

/* Flow fields of the pairs this block touches: global pairs
 * [max(frame_begin-1,0), min(frame_end, nc_global-1)), laid out
 * Ny x Nx x npairs x 2 exactly like the reference's v (helpers/constructWarpFF.m:5):
 * component 1 = displacement along columns, component 2 = along rows, in
 * high-resolution pixels.  occ (same pairs, Ny x Nx x npairs, non-zero =
 * occluded, may be NULL) is the optional occlusion map of constructWarpFF.m:31-35.
 * May be called again on a live handle to swap flows and keep the primal/dual
 * state (variants/MultiframeMotionCouplingAlternating.m:108-159); call
 * mmc_init again afterwards.  Replaces constructWarpFF / constructWarpFB. */
int mmc_set_flow(mmc_handle *h, const float *v, const uint8_t *occ);
int mmc_num_flow_pairs(const mmc_handle *h);

/* Start point: u and w for the local frames followed, when the block has a
 * next neighbour, by that neighbour's first frame (needed once, for h).
 * w may be NULL when the model has no w (kappa = 1 or NaN).  Duals start at 0.
 * Replaces u_vec.val / w_vec.val, MultiframeMotionCoupling.m:445-456.
 * u == NULL (and w == NULL): the library computes u = w = imresize(f, factor)
 * with the bicubic kernel on the device from the data given to mmc_set_lowres
 * (the u_up of MultiframeMotionCoupling.m:415-419; the start point of :686-687).
 * A block with a next neighbour then still lacks that neighbour's first frame:
 * exchange the halo once (send_prev -> recv_next) before mmc_h_partials. */
int mmc_set_state(mmc_handle *h, const float *u, const float *w);
int mmc_num_state_frames(const mmc_handle *h);

/* Pins the first frame of the SEQUENCE to (u0, w0), each Ny x Nx: the two extra
 * duals q1, q2 with identity blocks and F*(q) = <x0, q> of
 * MultiframeMotionCoupling.m:535-572, used to chain batches of a long video
 * (variants/VideoProcessor_supplementary_MMC.m:119-176).  Infimal-convolution
 * model only.  Blocks that do not hold frame 0 accept and ignore it; NULL, NULL
 * removes the constraint; one NULL is the reference's error (:190-192). */
int mmc_set_fixed_frame(mmc_handle *h, const float *u0, const float *w0);

/* The temporal rows of the first regulariser as init_u_flexBox writes them for flowDirection
 * 'backward': {Id, -Wi, Id, -Wi} on [u_i, u_i+1, w_i, w_i+1] (MultiframeMotionCoupling.m:864-868),
 * i.e. (Id - W)(u + w), where its 'forward' branch (:834) and init_u_prost (:490-491) couple
 * u - w.  enable != 0 reproduces :864 (a different model: only the sign of w in those rows
 * changes, the step sizes do not); 0, the default, is the u - w model.  Infimal-convolution
 * model only.  Takes effect at the next sweep; no mmc_init needed. */
int mmc_set_flexbox_backward_terms(mmc_handle *h, int enable);

/* Out-of-range / occlusion masks, column sums of the warp, tile source boxes,
 * step sizes and, for a block without neighbours and params.h <= 0, h from its
 * own sums.  A block of a sharded sequence holds only part of those sums: with
 * params.h <= 0 its h stays unset, and the sweeps return MMC_ERR_STATE, until
 * mmc_set_h has been called with the ratio of the all-reduced mmc_h_partials.
 * Replaces the one-off part of init_u_prost, MultiframeMotionCoupling.m:389-437. */
int mmc_init(mmc_handle *h);

/* h = (sum|W u|/M) / (sum|grad u|/M), MultiframeMotionCoupling.m:431-437:
 * out[0] = sum |W u0| over the pairs this block owns, out[1] = sum |grad u0|
 * over its frames, for the start point given to mmc_set_state. */
int mmc_h_partials(mmc_handle *h, double out[2]);
int mmc_set_h(mmc_handle *h, float value);
int mmc_get_h(const mmc_handle *h, float *value);

/* The two sums of mmc_h_partials for a whole sequence that already lives in device memory,
 * without a handle and without allocating anything (no hipMalloc / hipFree, no host allocation).
 * No counterpart in the reference.  It exists for callers that cannot allocate: code under HIP
 * graph capture, tight serving loops that recompute h for every incoming batch, harnesses that
 * forbid allocation after start-up.
 *   d_u  Ny x Nx x nc        device, the frames (the bicubic start point, :415-419)
 *   d_v  Ny x Nx x (nc-1) x 2 device, flows laid out as for mmc_set_flow; a NaN in either
 *        component masks the row (fold an occlusion map in beforehand); may be NULL when nc == 1
 *   d_out 2 doubles          device: [0] = sum |W u|, [1] = sum |grad u|
 * Both kernels are queued on `stream` (a hipStream_t, NULL = the default stream) of `device`;
 * the call does not synchronise, d_out is valid once the stream has passed it.  The caller's
 * current device is the same after the call as before it.  The first call on a device looks the
 * workspace's address up (hipGetSymbolAddress); make that call outside a stream capture.  The per-workgroup
 * partial sums live in a static device array of 8 MiB that each GPU used holds until the process
 * ends: ceil(Ny*Nx/256)*nc must not exceed MMC_STATIC_MAX_BLOCKS (30 frames of 1080p need
 * 243000, 15 frames of 2160x3840 need 486000); larger inputs are rejected with
 * MMC_ERR_INVALID_ARG before anything is launched (mmc_last_error(NULL) has the text).  All
 * calls on one GPU share that array: calls that overlap on one GPU (different streams or threads)
 * race and must be serialised by the caller.  The sums are added in a fixed order, so the result
 * is bitwise reproducible; it agrees with mmc_h_partials up to the rounding of double additions. */
#define MMC_STATIC_MAX_BLOCKS 524288
int mmc_h_partials_static(const float *d_u, const float *d_v, int Ny, int Nx, int nc,
                          int flow_direction, int device, void *stream, double *d_out);

/* The solve: prost.solve / flexBox.runAlgorithm, MultiframeMotionCoupling.m:981,
 * :989.  Runs at most max_iters PDHG iterations; every check_every iterations
 * (and after the last) the residuals are reduced, the callback runs, and the
 * loop stops if tol > 0 and the tolerance test passes (see mmc_stats).  Only for a block
 * without neighbours; blocks of a sharded sequence are driven sweep by sweep. */
int mmc_solve(mmc_handle *h, int max_iters, float tol, int check_every,
              mmc_stats *stats, mmc_progress_cb cb, void *user);

/* One half-iteration on local frames [t0, t1): primal update (needs recv_prev),
 * dual update (needs recv_next).  reduce != 0 accumulates the residual sums.
 * These are the pieces mmc_solve is made of; a sharded solve interleaves them
 * with the halo exchange. */
int mmc_sweep_primal(mmc_handle *h, int t0, int t1, int reduce);
int mmc_sweep_dual(mmc_handle *h, int t0, int t1, int reduce);
int mmc_sums_reset(mmc_handle *h);
int mmc_sums_read(mmc_handle *h, double out[MMC_NUM_SUMS]);   /* synchronises */
int mmc_halo_info(mmc_handle *h, mmc_halo *out);
int mmc_sync(mmc_handle *h);

/* Diagnostics (second group of mmc_stats) for a block driven sweep by sweep.
 * mmc_diag_partials evaluates the block's share of the sums at its current iterate and duals
 * (read-only; synchronises).  A block with a previous neighbour needs that neighbour's temporal
 * duals in its ghost slot (they are, after the exchange that follows a dual sweep); a block with a
 * next neighbour needs the neighbour's first frame of u and w -- the iterate, not xbar -- in the
 * ghost slot that mmc_state_halo_info names (send_prev -> the previous block's recv_next; the
 * other entries are NULL): exchange it before the call.  mmc_problem_size gives the block's number
 * of primal and dual unknowns.  mmc_stats_from_diag turns sums added up over all blocks into the
 * diagnostics fields of *st (the other fields are left alone) and returns 1 if they pass the
 * residual test for `tol` (0 when tol <= 0).  No counterpart in the reference: prost computes its
 * residuals and progress output inside prost.solve (MultiframeMotionCoupling.m:357-363, :981). */
int mmc_set_diagnostics(mmc_handle *h, int mode);
int mmc_diag_partials(mmc_handle *h, double out[MMC_NUM_DIAG]);
int mmc_state_halo_info(mmc_handle *h, mmc_halo *out);
int mmc_problem_size(const mmc_handle *h, double *n_primal, double *n_dual);
int mmc_stats_from_diag(const double diag[MMC_NUM_DIAG], double n_primal, double n_dual, float tol, mmc_stats *st);

/* Step balance.  The step sizes are the diagonal preconditioners T_j = 1 / sum_i |K_ij| and
 * S_i = 1 / sum_j |K_ij| with tau0 = sigma0 = 1 (MultiframeMotionCoupling.m:349-351); the reference
 * hands those start values to prost's adaptive rule ('stepsize','boyd'), which then moves weight
 * between the primal and the dual steps.  mmc_set_step_balance applies one global factor s > 0:
 * every tau is multiplied by s and every sigma divided by it, which keeps tau * sigma and with it
 * the convergence condition of the preconditioned iteration.  s = 1 is the default and the setting
 * of every fixed-iteration comparison with the oracle.  mmc_set_step_adaptation(h, 1) lets mmc_solve
 * choose s itself by residual balancing: every 50 iterations it compares the step-weighted changes
 * P = ||(x+ - x) / tau|| and D = ||(y+ - y) / sigma|| and, with Delta = 1.5, alpha = 0.5 at first
 * and alpha <- 0.8 alpha after every change, sets s <- s / (1 - alpha) when P > Delta D and
 * s <- s (1 - alpha) when P < D / Delta (mmc_step_balance_rule; the rule of adaptive PDHG, of which
 * prost's 'boyd' setting is an instance whose constants live in the absent prost submodule: own
 * definition, not a reproduction).  A block driven sweep by sweep reads its share of P^2 and D^2 with
 * mmc_step_sums_read after sweeps with reduce != 0, adds them over the blocks and applies the rule
 * to every block.  Takes effect at the next sweep; no mmc_init needed. */
int mmc_set_step_balance(mmc_handle *h, float s);
int mmc_get_step_balance(const mmc_handle *h, float *s);
int mmc_set_step_adaptation(mmc_handle *h, int enable);
int mmc_step_sums_read(mmc_handle *h, double out[2]);           /* synchronises */
/* The rule above as a function: sums = {P^2, D^2}; updates *balance and *alpha; returns 1 if it changed them */
int mmc_step_balance_rule(const double sums[2], double *balance, double *alpha);

/* Per-kernel timing of mmc_solve with HIP events recorded on the library's
 * stream around every launch (the reference only has tic/toc around run,
 * testfile.m:66-68).  With profiling enabled the next mmc_solve times up to
 * the first 512 iterations; mmc_get_profile then returns the average
 * duration in milliseconds of [0] the primal sweep, [1] the dual sweep
 * (regulariser and data term), [2] always 0 (kept for layout), and the number
 * of iterations averaged.  One event is recorded between consecutive sweeps. */
int mmc_set_profiling(mmc_handle *h, int enable);
int mmc_get_profile(mmc_handle *h, double out_ms[3], int *iterations);
/* The same for a block driven through mmc_sweep_primal / mmc_sweep_dual: summed
 * duration in milliseconds and number of the timed calls (at most the first
 * 4096 since mmc_set_profiling), [0] primal sweeps, [1] dual sweeps.
 * Synchronises the stream.  No counterpart in the reference. */
int mmc_get_sweep_profile(mmc_handle *h, double total_ms[2], int calls[2]);

/* Result extraction, MultiframeMotionCoupling.m:983-986: u and w of the local
 * frames (w may be NULL). */
int mmc_get_state(mmc_handle *h, float *u, float *w);

/* Primal objective of the current iterate restricted to this block:
 * ||A u - f||_1 + regulariser, for logging (the quantity prost prints).  Only for a block without
 * a next neighbour (use mmc_diag_partials otherwise). */
int mmc_energy(mmc_handle *h, double *value);

/* ---- One sequence on several GPUs from ONE host thread ------------------------
 * The caller of solveU (MultiframeMotionCoupling.m:977-986) is one MATLAB session,
 * i.e. one process; a group lets that process spread the sequence over the GPUs of
 * the node and keep the reference's shape of the call: one solve in, u and w out.
 * A group cuts [0, nc_global) into nblocks contiguous frame blocks (remainders to
 * the first blocks), gives each block a handle of its own on device_ids[i]
 * (device_ids == NULL: all on params.device; several blocks may share a device)
 * and runs the iteration on all of them: each block's sweeps on its own stream,
 * the two frame-sized messages per block boundary and iteration (mmc_halo) as
 * peer-to-peer copies on a copy stream per block, ordered by events so that a copy
 * overlaps with the sweeps over the frames that do not need it.  With params.h <= 0
 * mmc_group_init adds mmc_h_partials up over the blocks (:431-437).
 * params describes the whole sequence: frame_begin = 0, frame_end = nc_global,
 * stream = NULL.  Host buffers have the sizes of the whole sequence and the layouts
 * of the per-block calls of the same name: f ny x nx x nc; v Ny x Nx x (nc-1) x 2;
 * occ Ny x Nx x (nc-1); u, w Ny x Nx x nc (NULL, NULL: bicubic start point).
 * For equal h a group's iterates are bit-for-bit those of one block over the whole
 * sequence (the sweeps are order-independent); residuals are sums of doubles added
 * in another order.  Calls on one group must come from one thread at a time. */
typedef struct mmc_group mmc_group;
int mmc_group_create(const mmc_params *params, int nblocks, const int32_t *device_ids, mmc_group **out);
int mmc_group_destroy(mmc_group *g);
/* g == NULL: the last mmc_group_create failure of the calling thread */
const char *mmc_group_last_error(const mmc_group *g);
int mmc_group_num_blocks(const mmc_group *g);
int mmc_group_block_range(const mmc_group *g, int block, int32_t *frame_begin, int32_t *frame_end, int32_t *device);
/* The handle of one block, e.g. for mmc_set_profiling / mmc_get_sweep_profile / mmc_energy;
 * it stays owned by the group.  NULL for a bad index. */
mmc_handle *mmc_group_block_handle(mmc_group *g, int block);
int mmc_group_set_lowres(mmc_group *g, const float *f);
int mmc_group_set_blur(mmc_group *g, const float *k, int kh, int kw);
int mmc_group_set_sampling_taps(mmc_group *g, const int32_t *first_y, const float *w_y, int taps_y,
                                const int32_t *first_x, const float *w_x, int taps_x);
int mmc_group_set_flow(mmc_group *g, const float *v, const uint8_t *occ);
int mmc_group_set_state(mmc_group *g, const float *u, const float *w);
int mmc_group_set_fixed_frame(mmc_group *g, const float *u0, const float *w0);
int mmc_group_set_flexbox_backward_terms(mmc_group *g, int enable);
int mmc_group_init(mmc_group *g);
int mmc_group_get_h(const mmc_group *g, float *value);
int mmc_group_set_h(mmc_group *g, float value);
/* As mmc_solve; the residuals are those of the whole sequence. */
int mmc_group_solve(mmc_group *g, int max_iters, float tol, int check_every,
                    mmc_stats *stats, mmc_progress_cb cb, void *user);
int mmc_group_get_state(mmc_group *g, float *u, float *w);
/* mmc_set_diagnostics for every block of the group */
int mmc_group_set_diagnostics(mmc_group *g, int mode);
/* mmc_set_step_balance / mmc_set_step_adaptation for the whole sequence (one s for all blocks) */
int mmc_group_set_step_balance(mmc_group *g, float s);
int mmc_group_set_step_adaptation(mmc_group *g, int enable);

/* ---- Test hooks: the operators of the sweeps without the prox maps. ---------
 * K x for x = (u, w): out_p is n*L, out_g1/out_g2 are component-major
 * [temporal; dx; dy] of N*L each (additive model: g1 = temporal only,
 * g2 = [dx; dy]).  Input frames as for mmc_set_state. */
int mmc_debug_apply_K(mmc_handle *h, const float *u, const float *w,
                      float *out_p, float *out_g1, float *out_g2);
/* K' y: p is n*L, g1/g2 as above, plus the previous block's temporal duals
 * (ghost, may be NULL). */
int mmc_debug_apply_KT(mmc_handle *h, const float *p, const float *g1, const float *g2,
                       const float *ghost_g1t, const float *ghost_g2t,
                       float *out_u, float *out_w);
/* Diagonal step sizes: tau for u, w (N*L each) and sigma for p (n*L), g1, g2
 * (N*L each, one per pixel group). */
int mmc_debug_steps(mmc_handle *h, float *tau_u, float *tau_w,
                    float *sigma_p, float *sigma_g1, float *sigma_g2);
/* tau as the update sweep itself derives it: there the column sums of the motion-coupling
 * operator come out of the tile scatter of the tap weights plus two identity bits per pixel,
 * not from the plane that mmc_init wrote (which mmc_debug_steps reads).  Both must agree
 * bit for bit. */
int mmc_debug_steps_update(mmc_handle *h, float *tau_u, float *tau_w);

/* Placement of the per-pixel planes in device memory: plane i starts i * (this many bytes) into its
 * allocation, so that the planes a sweep streams through together do not all sit at the same address
 * modulo a power of two.  mmc_create picks the value for large blocks by timing a streaming probe.
 * Results do not depend on it. */
size_t mmc_debug_plane_skew(const mmc_handle *h);

/* 1 after mmc_init when the sweeps apply the composite kernel of sampler and blur as two 1-D passes: zoom 4, the
 * kernel an outer product (the samplers of helpers/samplingOperator.m:146 are, a Gaussian blur is) and larger
 * than 3 x 3 low-res taps.  Results agree with the one-pass form to float32 rounding. */
int mmc_debug_separable(const mmc_handle *h);

/* ---- Classical single-frame super-resolution (the reference's comparison methods) ------------
 * classical methods/singleframeMotionSR_mitzel_prost.m, ..._mitzel_prost_fast.m and
 * ..._unger_prost.m assemble, after their flow estimation (which stays on the reference path),
 * one problem for prost.solve (..._mitzel_prost.m:82-172):
 *     min_u  sum_i H_delta(D B W_i u - f_i) + alpha sum_j H_delta(|grad u|_j)
 * u: the central high-resolution frame (N = Ny*Nx); f_i: low-resolution frame i; W_i: bilinear
 * warp with the flow from the central frame to frame i (:64-70); D the box sampler of :75-77
 * (= 'average'), B the blur of :78-79; H_delta the Huber function, delta = 0 the absolute value
 * (huber_eps of ..._unger_prost.m:148-161).  These calls replace :82-172 of those files: problem
 * assembly and solve.  Same conventions as the calls above (float32, column-major, status
 * codes); mmc_sf_last_error(NULL) is the last mmc_sf_create failure of the calling thread. */
typedef struct mmc_sf_handle mmc_sf_handle;
typedef struct mmc_sf_params {
    uint32_t size;             /* sizeof(mmc_sf_params) */
    uint32_t version;          /* MMC_ABI_VERSION */
    int32_t ny, nx;            /* low-resolution frame size */
    int32_t factor;            /* zoom factor */
    int32_t num_frames;        /* low-resolution frames; the central one is ceil(num_frames/2), 1-based (:21) */
    float alpha;               /* weight of TV: ||q_j|| <= alpha (:163) */
    float huber_eps;           /* delta >= 0: Huber parameter of the data terms (..._unger_prost.m:148-161) and of the
                                  regulariser (dual term (delta/alpha)/2 ||q_j||^2, ..._unger_prost.m:163); 0 for Mitzel */
    int32_t sampling;          /* MMC_SAMPLING_AVERAGE (factor 4) or MMC_SAMPLING_STRIDE */
    int32_t device;
    void *stream;              /* hipStream_t to queue on; NULL: the library makes one */
} mmc_sf_params;
int mmc_sf_create(const mmc_sf_params *params, mmc_sf_handle **out);
int mmc_sf_destroy(mmc_sf_handle *h);
const char *mmc_sf_last_error(const mmc_sf_handle *h);
/* f: ny x nx x num_frames (J0 of :14-20) */
int mmc_sf_set_lowres(mmc_sf_handle *h, const float *f);
/* k: kh x kw, odd sizes, no negative entry (the step sizes take absolute row and column sums of
 * D*B*W_i as sums of products); NULL: no blur */
int mmc_sf_set_blur(mmc_sf_handle *h, const float *k, int kh, int kw);
/* v: Ny x Nx x num_frames x 2 (:31-60; component 1 along columns, 2 along rows, as warpingOperator
 * takes it at :69); the central frame's slice is ignored and taken as zero (:35-37).
 * occ: NULL or Ny x Nx x num_frames, non-zero drops that row of W_i. */
int mmc_sf_set_flow(mmc_sf_handle *h, const float *v, const uint8_t *occ);
/* Start point, Ny x Nx; NULL: zero (prost.variable(N) starts there, :97).  Duals start at zero. */
int mmc_sf_set_state(mmc_sf_handle *h, const float *u);
/* Source rectangles of the tile scatter and diagonal step sizes; needs low-res data and flow,
 * leaves the iterate and the duals as they are.  Call again after mmc_sf_set_flow / mmc_sf_set_blur. */
int mmc_sf_init(mmc_sf_handle *h);
/* As mmc_solve: at most max_iters iterations (:86-92: 7500 'accurate', 2500 otherwise); tol > 0
 * stops on the relative change of the iterates, checked every check_every iterations. */
int mmc_sf_solve(mmc_sf_handle *h, int max_iters, float tol, int check_every,
                 mmc_stats *stats, mmc_progress_cb cb, void *user);
int mmc_sf_get_state(mmc_sf_handle *h, float *u);       /* I of :172 */
/* Test hooks.  K u: out_p num_frames*n, out_q 2N ([d/dcol; d/drow]).  K' y: out_u N.
 * Steps: tau N, sigma_p num_frames*n, sigma_q N. */
int mmc_sf_debug_apply_K(mmc_sf_handle *h, const float *u, float *out_p, float *out_q);
int mmc_sf_debug_apply_KT(mmc_sf_handle *h, const float *p, const float *q, float *out_u);
int mmc_sf_debug_steps(mmc_sf_handle *h, float *tau, float *sigma_p, float *sigma_q);

#ifdef __cplusplus
}
#endif
#endif /* MMC_HIP_H */
