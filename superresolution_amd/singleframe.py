"""Classical single-frame super-resolution with motion-compensated data terms: host-side mirror
of the reference's comparison methods (classical methods/singleframeMotionSR_mitzel_prost.m,
..._mitzel_prost_fast.m, ..._unger_prost.m) with the solve on MI355X.

The three reference functions estimate flows from the central frame to every other frame
(motionEstimatorClass, :28-62) and then hand one problem to prost.solve (:82-172).  Flow
estimation stays on the reference path here: the flows are an argument.  What follows them --
the operators D, B, W_i (:64-79), the solve and the colour handling (:175-185) -- is mirrored.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import MMCError, as_f32, fptr
from .imresize import imresize_bicubic


def fspecial_gaussian(size, sigma):
    """MATLAB fspecial('gaussian', size, sigma)."""
    r = (size - 1) / 2.0
    y, x = np.mgrid[-r:r + 1, -r:r + 1]
    h = np.exp(-(x * x + y * y) / (2.0 * sigma * sigma))
    h[h < np.finfo(float).eps * h.max()] = 0
    return h / h.sum()


def central_slice(num_frames):
    """0-based index of ceil(numFrames/2) (:21)."""
    return (num_frames + 1) // 2 - 1


class SingleFrameSolver(object):
    """One mmc_sf_* handle (include/mmc_hip.h), numpy in and out."""

    def __init__(self, ny, nx, factor, num_frames, alpha=0.1, huber_eps=0.0, sampling="average", device=0, stream=None):
        self._lib = _capi.load_library()
        self._h = C.c_void_p()
        self.ny, self.nx, self.factor, self.nf = int(ny), int(nx), int(factor), int(num_frames)
        self.Ny, self.Nx = self.ny * self.factor, self.nx * self.factor
        codes = {"average": _capi.SAMPLING_AVERAGE, "stride": _capi.SAMPLING_STRIDE}
        if sampling not in codes:
            raise MMCError(-4, "the single-frame solve takes the samplers 'average' and 'stride'")
        p = _capi.mmc_sf_params()
        p.size, p.version = C.sizeof(p), _capi.MMC_ABI_VERSION
        p.ny, p.nx, p.factor, p.num_frames = self.ny, self.nx, self.factor, self.nf
        p.alpha, p.huber_eps = float(alpha), float(huber_eps)
        p.sampling, p.device, p.stream = codes[sampling], int(device), stream
        self.params = p
        rc = self._lib.mmc_sf_create(C.byref(p), C.byref(self._h))
        if rc != 0:
            raise MMCError(rc, self._lib.mmc_sf_last_error(None).decode())

    def _check(self, rc):
        if rc != 0:
            raise MMCError(rc, self._lib.mmc_sf_last_error(self._h).decode())

    def close(self):
        if self._h:
            self._lib.mmc_sf_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_lowres(self, f):
        f = as_f32(f)
        if f.shape != (self.ny, self.nx, self.nf):
            raise ValueError("low-res sequence must be %s, got %s" % ((self.ny, self.nx, self.nf), f.shape))
        self._check(self._lib.mmc_sf_set_lowres(self._h, fptr(f)))

    def set_blur(self, k):
        if k is None or np.size(k) == 0:
            self._check(self._lib.mmc_sf_set_blur(self._h, None, 0, 0))
            return
        k = as_f32(k)
        self._check(self._lib.mmc_sf_set_blur(self._h, fptr(k), k.shape[0], k.shape[1]))

    def set_flow(self, v, occ=None):
        v = as_f32(v)
        if v.shape != (self.Ny, self.Nx, self.nf, 2):
            raise ValueError("flowField must be %s, got %s" % ((self.Ny, self.Nx, self.nf, 2), v.shape))
        o = None
        if occ is not None:
            o = np.asfortranarray(np.asarray(occ) != 0, dtype=np.uint8)
            if o.shape != (self.Ny, self.Nx, self.nf):
                raise ValueError("occlusion map must be %s" % ((self.Ny, self.Nx, self.nf),))
        self._check(self._lib.mmc_sf_set_flow(self._h, fptr(v), None if o is None else o.ctypes.data_as(_capi._u8p)))

    def set_state(self, u=None):
        if u is None:
            self._check(self._lib.mmc_sf_set_state(self._h, None))
            return
        u = as_f32(u)
        if u.shape != (self.Ny, self.Nx):
            raise ValueError("start point must be %s" % ((self.Ny, self.Nx),))
        self._check(self._lib.mmc_sf_set_state(self._h, fptr(u)))

    def init(self):
        self._check(self._lib.mmc_sf_init(self._h))

    def solve(self, max_iters, tol=0.0, check_every=0, callback=None):
        st = _capi.mmc_stats()

        def tramp(it, sp, _user):
            s = sp.contents
            return int(bool(callback(it, {"res_primal": s.res_primal, "res_dual": s.res_dual, "seconds": s.solve_seconds})))

        cb = _capi.PROGRESS_CB(tramp) if callback is not None else C.cast(None, _capi.PROGRESS_CB)
        self._check(self._lib.mmc_sf_solve(self._h, int(max_iters), float(tol), int(check_every), C.byref(st), cb, None))
        return {"iterations": st.iterations, "converged": bool(st.converged), "res_primal": st.res_primal,
                "res_dual": st.res_dual, "seconds": st.solve_seconds}

    def get_state(self):
        u = np.empty((self.Ny, self.Nx), dtype=np.float32, order="F")
        self._check(self._lib.mmc_sf_get_state(self._h, fptr(u)))
        return u

    # test hooks
    def apply_K(self, u):
        u = as_f32(u)
        N, n = self.Ny * self.Nx, self.ny * self.nx
        p = np.empty(self.nf * n, dtype=np.float32)
        q = np.empty(2 * N, dtype=np.float32)
        self._check(self._lib.mmc_sf_debug_apply_K(self._h, fptr(u), fptr(p), fptr(q)))
        return p, q

    def apply_KT(self, p, q):
        p = np.ascontiguousarray(p, dtype=np.float32)
        q = np.ascontiguousarray(q, dtype=np.float32)
        out = np.empty(self.Ny * self.Nx, dtype=np.float32)
        self._check(self._lib.mmc_sf_debug_apply_KT(self._h, fptr(p), fptr(q), fptr(out)))
        return out

    def steps(self):
        N, n = self.Ny * self.Nx, self.ny * self.nx
        tau = np.empty(N, dtype=np.float32)
        sp = np.empty(self.nf * n, dtype=np.float32)
        sq = np.empty(N, dtype=np.float32)
        self._check(self._lib.mmc_sf_debug_steps(self._h, fptr(tau), fptr(sp), fptr(sq)))
        return tau, sp, sq


def singleframeMotionSR(imageSequenceSmall, factor, alpha, flowField, huber_eps=0.0, comp_mode="accurate",
                        occlusion=None, device=0, verbose=True):
    """[imgSR, timings] of singleframeMotionSR_mitzel_prost(imageSequenceSmall, factor, alpha, beta, comp_mode)
    (huber_eps = 0) and of singleframeMotionSR_unger_prost(..., huber_eps, comp_mode), with the flows
    those functions estimate at :28-62 passed in as `flowField` (Ny x Nx x numFrames x 2, from the
    central frame to frame i; the central slice is ignored).

    imageSequenceSmall: ny x nx x 3 x numFrames RGB in [0,1] (the reference's input, :11-20), or
    ny x nx x numFrames grey (returned as a grey image).  Returns (imgSR, timings) with
    timings = [0, solve seconds] (:9, :168-169; no flow estimation is timed here)."""
    from .mmc import rgb2ycbcr, ycbcr2rgb
    seq = np.asarray(imageSequenceSmall, dtype=np.float64)
    colour = seq.ndim == 4
    if colour:
        nf = seq.shape[3]
        J0 = np.stack([rgb2ycbcr(seq[:, :, :, i])[:, :, 0] for i in range(nf)], axis=2)     # :14-20
    elif seq.ndim == 3:
        nf = seq.shape[2]
        J0 = seq
    else:
        raise ValueError("Load an actual video sequence")
    ny, nx = J0.shape[:2]
    cs = central_slice(nf)                                                                   # :21
    sigm_unger = 0.25 * np.sqrt(factor ** 2 - 1)                                             # :78
    k = fspecial_gaussian(7, sigm_unger)                                                     # :79
    iters = 7500 if comp_mode == "accurate" else 2500                                        # :86-90
    with SingleFrameSolver(ny, nx, factor, nf, alpha=alpha, huber_eps=huber_eps, sampling="average", device=device) as s:
        s.set_lowres(J0)
        s.set_blur(k)
        s.set_flow(flowField, occlusion)
        s.set_state(None)
        s.init()
        if verbose:
            print("Single frame motion coupling initialized")                               # :165
        cb = None
        if verbose:
            def cb(it, d):
                print("  it %6d  res_primal %.3e  res_dual %.3e  (%.2f s)" % (it, d["res_primal"], d["res_dual"], d["seconds"]))
                return False
        st = s.solve(iters, 0.0, max(iters // 5, 1) if verbose else 0, cb)                   # num_cback_calls 5, :91
        I = s.get_state().astype(np.float64)                                                 # :172
    timings = np.array([0.0, st["seconds"]])
    if not colour:
        return I, timings
    ycc = rgb2ycbcr(seq[:, :, :, cs])
    up = np.stack([imresize_bicubic(ycc[:, :, c], factor) for c in range(3)], axis=2)        # :178
    up[:, :, 0] = I                                                                          # :179
    return ycbcr2rgb(up), timings                                                            # :185
