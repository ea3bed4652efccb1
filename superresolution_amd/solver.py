"""FrameBlockSolver: numpy-facing wrapper of one libmmc_hip handle.

One instance owns the device state of a contiguous block of frames.  For a
whole sequence on one GPU the block is [0, nc); a sharded sequence uses one
block per rank (see sharding.py).  Arrays follow the reference's MATLAB layout:
frames are (rows, cols, frame) and flows (Ny, Nx, pairs, 2) with component 0
the displacement along columns (helpers/constructWarpFF.m:5).
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import MMCError, as_f32, fptr

_DIRECTIONS = {"forward": _capi.FLOW_FORWARD, "backward": _capi.FLOW_BACKWARD,
               "forward-backward": _capi.FLOW_FORWARD_BACKWARD}
_SAMPLINGS = {"average": _capi.SAMPLING_AVERAGE, "stride": _capi.SAMPLING_STRIDE, "taps": _capi.SAMPLING_TAPS}


class FrameBlockSolver(object):
    def __init__(self, ny, nx, factor, nc_global, frame_begin=0, frame_end=None,
                 alpha=0.01, kappa=0.25, h=0.0, flow_direction="backward",
                 sampling="average", device=0, stream=None):
        self._lib = _capi.load_library()
        self._h = C.c_void_p()
        if flow_direction not in _DIRECTIONS:
            raise ValueError("Invalid flow direction.")
        if sampling not in _SAMPLINGS:
            raise MMCError(-4, "sampling must be 'average', 'stride' or 'taps' (rows given with "
                               "set_sampling_taps), got '%s'" % sampling)
        frame_end = nc_global if frame_end is None else frame_end
        p = _capi.mmc_params()
        p.size = C.sizeof(_capi.mmc_params)
        p.version = _capi.MMC_ABI_VERSION
        p.ny, p.nx, p.factor = int(ny), int(nx), int(factor)
        p.nc_global, p.frame_begin, p.frame_end = int(nc_global), int(frame_begin), int(frame_end)
        p.alpha, p.kappa, p.h = float(alpha), float(kappa), float(h)
        p.flow_direction = _DIRECTIONS[flow_direction]
        p.sampling = _SAMPLINGS[sampling]
        p.device = int(device)
        p.stream = stream
        self.params = p
        rc = self._lib.mmc_create(C.byref(p), C.byref(self._h))
        if rc != 0:
            raise MMCError(rc, self._lib.mmc_last_error(None).decode())
        self.ny, self.nx, self.factor = int(ny), int(nx), int(factor)
        self.Ny, self.Nx = self.ny * self.factor, self.nx * self.factor
        self.L = int(frame_end) - int(frame_begin)
        self.nc_global = int(nc_global)
        self.frame_begin, self.frame_end = int(frame_begin), int(frame_end)
        self.has_prev = frame_begin > 0
        self.has_next = frame_end < nc_global
        self.has_w = not (np.isnan(kappa) or kappa == 1)
        self.additive = bool(np.isnan(kappa))
        self._cb_keepalive = None

    # -- plumbing ----------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            raise MMCError(rc, self._lib.mmc_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.mmc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def num_flow_pairs(self):
        return self._lib.mmc_num_flow_pairs(self._h)

    @property
    def num_state_frames(self):
        return self._lib.mmc_num_state_frames(self._h)

    # -- inputs ------------------------------------------------------------
    def set_lowres(self, f):
        f = as_f32(f)
        if f.shape != (self.ny, self.nx, self.L):
            raise ValueError("low-res block must be %s, got %s" % ((self.ny, self.nx, self.L), f.shape))
        self._check(self._lib.mmc_set_lowres(self._h, fptr(f)))

    def set_blur(self, k):
        if k is None or np.size(k) == 0:
            self._check(self._lib.mmc_set_blur(self._h, None, 0, 0))
            return
        k = as_f32(np.atleast_2d(k))
        self._check(self._lib.mmc_set_blur(self._h, fptr(k), k.shape[0], k.shape[1]))

    def set_sampling_taps(self, first_y, w_y, first_x, w_x):
        """1-D factors of the sampling operator in banded form (see sampling.dim1_taps)."""
        fy = np.ascontiguousarray(first_y, dtype=np.int32)
        fx = np.ascontiguousarray(first_x, dtype=np.int32)
        wy = np.ascontiguousarray(w_y, dtype=np.float32)
        wx = np.ascontiguousarray(w_x, dtype=np.float32)
        if fy.shape != (self.ny,) or wy.ndim != 2 or wy.shape[0] != self.ny:
            raise ValueError("row taps must be (%d,) and (%d, T)" % (self.ny, self.ny))
        if fx.shape != (self.nx,) or wx.ndim != 2 or wx.shape[0] != self.nx:
            raise ValueError("column taps must be (%d,) and (%d, T)" % (self.nx, self.nx))
        i32 = C.POINTER(C.c_int32)
        self._check(self._lib.mmc_set_sampling_taps(self._h, fy.ctypes.data_as(i32), fptr(wy), wy.shape[1],
                                                    fx.ctypes.data_as(i32), fptr(wx), wx.shape[1]))

    def set_flow(self, v, occ=None):
        npairs = self.num_flow_pairs
        if npairs == 0:
            self._check(self._lib.mmc_set_flow(self._h, None, None))
            return
        v = as_f32(v).reshape((self.Ny, self.Nx, -1, 2), order="F")
        if v.shape[2] != npairs:
            raise ValueError("flow must hold %d pairs, got %d" % (npairs, v.shape[2]))
        o = None
        if occ is not None:
            o = np.asfortranarray(np.asarray(occ).reshape((self.Ny, self.Nx, npairs), order="F") != 0,
                                  dtype=np.uint8)
        self._check(self._lib.mmc_set_flow(self._h, fptr(v),
                                           None if o is None else o.ctypes.data_as(C.POINTER(C.c_uint8))))

    def set_state(self, u=None, w=None):
        """Start point; u=None: bicubic estimate of the low-res data, computed on the device."""
        if u is None:
            if w is not None:
                raise ValueError("w without u")
            self._check(self._lib.mmc_set_state(self._h, None, None))
            return
        nf = self.num_state_frames
        u = as_f32(u)
        if u.shape != (self.Ny, self.Nx, nf):
            raise ValueError("state must be %s, got %s" % ((self.Ny, self.Nx, nf), u.shape))
        if self.has_w:
            w = as_f32(u if w is None else w)
            if w.shape != u.shape:
                raise ValueError("w must have the shape of u")
        else:
            w = None
        self._check(self._lib.mmc_set_state(self._h, fptr(u), fptr(w)))

    def set_fixed_frame(self, u0, w0):
        """Pin frame 0 of the sequence to (u0, w0); (None, None) removes the constraint."""
        if u0 is None and w0 is None:
            self._check(self._lib.mmc_set_fixed_frame(self._h, None, None))
            return
        a = None if u0 is None else as_f32(u0)
        b = None if w0 is None else as_f32(w0)
        for x in (a, b):
            if x is not None and x.shape != (self.Ny, self.Nx):
                raise ValueError("fixed frame must be %s" % ((self.Ny, self.Nx),))
        self._check(self._lib.mmc_set_fixed_frame(self._h, fptr(a), fptr(b)))

    def set_flexbox_backward_terms(self, enable=True):
        """Temporal rows on u + w, as init_u_flexBox writes them for 'backward' (:864)."""
        self._check(self._lib.mmc_set_flexbox_backward_terms(self._h, int(bool(enable))))

    def init(self):
        self._check(self._lib.mmc_init(self._h))

    # -- h -----------------------------------------------------------------
    def h_partials(self):
        out = (C.c_double * 2)()
        self._check(self._lib.mmc_h_partials(self._h, out))
        return float(out[0]), float(out[1])

    @property
    def h(self):
        v = C.c_float()
        self._check(self._lib.mmc_get_h(self._h, C.byref(v)))
        return float(v.value)

    @h.setter
    def h(self, value):
        self._check(self._lib.mmc_set_h(self._h, float(value)))

    # -- solve -------------------------------------------------------------
    def solve(self, max_iters, tol=0.0, check_every=0, callback=None):
        st = _capi.mmc_stats()
        if callback is not None:
            def _cb(it, stp, _user):
                s = stp.contents
                return int(bool(callback(it, s.as_dict())))
            cb = _capi.PROGRESS_CB(_cb)
        else:
            cb = _capi.PROGRESS_CB()
        self._cb_keepalive = cb
        self._check(self._lib.mmc_solve(self._h, int(max_iters), float(tol), int(check_every),
                                        C.byref(st), cb, None))
        return st.as_dict()

    def sweep_primal(self, t0, t1, reduce=False):
        self._check(self._lib.mmc_sweep_primal(self._h, int(t0), int(t1), int(reduce)))

    def sweep_dual(self, t0, t1, reduce=False):
        self._check(self._lib.mmc_sweep_dual(self._h, int(t0), int(t1), int(reduce)))

    def sums_reset(self):
        self._check(self._lib.mmc_sums_reset(self._h))

    def sums_read(self):
        out = (C.c_double * _capi.MMC_NUM_SUMS)()
        self._check(self._lib.mmc_sums_read(self._h, out))
        return np.array(list(out), dtype=np.float64)

    def halo_info(self):
        hl = _capi.mmc_halo()
        self._check(self._lib.mmc_halo_info(self._h, C.byref(hl)))
        return hl

    def sync(self):
        self._check(self._lib.mmc_sync(self._h))

    # -- step balance ---------------------------------------------------------
    @property
    def step_balance(self):
        """Global factor s: every tau times s, every sigma divided by s (1 by default)."""
        v = C.c_float()
        self._check(self._lib.mmc_get_step_balance(self._h, C.byref(v)))
        return float(v.value)

    @step_balance.setter
    def step_balance(self, s):
        self._check(self._lib.mmc_set_step_balance(self._h, float(s)))

    def set_step_adaptation(self, enable=True):
        """mmc_solve chooses the step balance by residual balancing every 50 iterations."""
        self._check(self._lib.mmc_set_step_adaptation(self._h, int(bool(enable))))

    def step_sums_read(self):
        """This block's sums of ((x+ - x)/tau)^2 and ((y+ - y)/sigma)^2 of the last sweeps with reduce (synchronises)."""
        out = (C.c_double * 2)()
        self._check(self._lib.mmc_step_sums_read(self._h, out))
        return np.array(list(out), dtype=np.float64)

    # -- diagnostics: energies, gap, optimality residuals ----------------------
    def set_diagnostics(self, mode):
        """_capi.DIAG_OFF / DIAG_AUTO (default: when the solve has a tolerance or a callback) / DIAG_ALWAYS."""
        self._check(self._lib.mmc_set_diagnostics(self._h, int(mode)))

    def diag_partials(self):
        """This block's share of the MMC_NUM_DIAG sums at the current iterate and duals (synchronises)."""
        out = (C.c_double * _capi.MMC_NUM_DIAG)()
        self._check(self._lib.mmc_diag_partials(self._h, out))
        return np.array(list(out), dtype=np.float64)

    def state_halo_info(self):
        hl = _capi.mmc_halo()
        self._check(self._lib.mmc_state_halo_info(self._h, C.byref(hl)))
        return hl

    def problem_size(self):
        a, b = C.c_double(), C.c_double()
        self._check(self._lib.mmc_problem_size(self._h, C.byref(a), C.byref(b)))
        return float(a.value), float(b.value)

    def diagnostics(self, tol=0.0):
        """Energies, gap and optimality residuals of a block without neighbours, as a dict, plus
        'passes' (the residual test for tol)."""
        if self.has_prev or self.has_next:
            raise MMCError(-2, "diagnostics(): this block has neighbours; add diag_partials() up over the blocks")
        return stats_from_diag(self._lib, self.diag_partials(), *self.problem_size(), tol=tol)

    def set_profiling(self, enable=True):
        self._check(self._lib.mmc_set_profiling(self._h, int(bool(enable))))

    def get_profile(self):
        """Average kernel durations (ms) of the last profiled solve."""
        ms = (C.c_double * 3)()
        n = C.c_int()
        self._check(self._lib.mmc_get_profile(self._h, ms, C.byref(n)))
        return {"primal_ms": ms[0], "dual_reg_ms": ms[1], "dual_data_ms": ms[2], "iterations": n.value}

    def get_sweep_profile(self):
        """Summed duration (ms) and count of the profiled mmc_sweep_* calls."""
        ms = (C.c_double * 2)()
        n = (C.c_int * 2)()
        self._check(self._lib.mmc_get_sweep_profile(self._h, ms, n))
        return {"primal_ms": ms[0], "dual_ms": ms[1], "primal_calls": n[0], "dual_calls": n[1]}

    def energy(self):
        v = C.c_double()
        self._check(self._lib.mmc_energy(self._h, C.byref(v)))
        return float(v.value)

    # -- outputs -----------------------------------------------------------
    def get_state(self):
        u = np.empty((self.Ny, self.Nx, self.L), dtype=np.float32, order="F")
        w = np.empty_like(u) if self.has_w else None
        self._check(self._lib.mmc_get_state(self._h, fptr(u), fptr(w)))
        return u, w

    # -- test hooks --------------------------------------------------------
    def debug_apply_K(self, u, w=None):
        nf = self.num_state_frames
        u = as_f32(u)
        assert u.shape == (self.Ny, self.Nx, nf)
        w = as_f32(w) if self.has_w else None
        N, n, L = self.Ny * self.Nx, self.ny * self.nx, self.L
        out_p = np.empty(n * L, np.float32)
        if self.additive:
            g1 = np.empty(N * L, np.float32)
            g2 = np.empty(2 * N * L, np.float32)
        else:
            g1 = np.empty(3 * N * L, np.float32)
            g2 = np.empty(3 * N * L, np.float32) if self.has_w else None
        self._check(self._lib.mmc_debug_apply_K(self._h, fptr(u), fptr(w), fptr(out_p), fptr(g1), fptr(g2)))
        return out_p, g1, g2

    def debug_apply_KT(self, p, g1, g2=None, ghost_g1t=None, ghost_g2t=None):
        N, L = self.Ny * self.Nx, self.L
        p, g1 = as_f32(p).ravel(), as_f32(g1).ravel()
        g2 = None if g2 is None else as_f32(g2).ravel()
        gh1 = None if ghost_g1t is None else as_f32(ghost_g1t).ravel(order="F")
        gh2 = None if ghost_g2t is None else as_f32(ghost_g2t).ravel(order="F")
        out_u = np.empty(N * L, np.float32)
        out_w = np.empty(N * L, np.float32) if self.has_w else None
        self._check(self._lib.mmc_debug_apply_KT(self._h, fptr(p), fptr(g1), fptr(g2), fptr(gh1), fptr(gh2),
                                                 fptr(out_u), fptr(out_w)))
        return out_u, out_w

    def debug_steps(self):
        N, n, L = self.Ny * self.Nx, self.ny * self.nx, self.L
        tu = np.empty(N * L, np.float32)
        tw = np.empty(N * L, np.float32) if self.has_w else None
        sp = np.empty(n * L, np.float32)
        s1 = np.empty(N * L, np.float32)
        s2 = np.empty(N * L, np.float32)
        self._check(self._lib.mmc_debug_steps(self._h, fptr(tu), fptr(tw), fptr(sp), fptr(s1), fptr(s2)))
        return tu, tw, sp, s1, s2

    @property
    def plane_skew(self):
        """Bytes by which consecutive per-pixel planes are shifted against each other (placement only)."""
        return int(self._lib.mmc_debug_plane_skew(self._h))

    @property
    def separable(self):
        """True when the sweeps apply the sampler-and-blur kernel as two 1-D passes (see mmc_debug_separable)."""
        return bool(self._lib.mmc_debug_separable(self._h))

    def debug_steps_update(self):
        """tau_u, tau_w as the update sweep derives them (see mmc_debug_steps_update)."""
        N, L = self.Ny * self.Nx, self.L
        tu = np.empty(N * L, np.float32)
        tw = np.empty(N * L, np.float32) if self.has_w else None
        self._check(self._lib.mmc_debug_steps_update(self._h, fptr(tu), fptr(tw)))
        return tu, tw


def stats_from_diag(lib, diag, n_primal, n_dual, tol=0.0):
    """The diagnostics fields of mmc_stats from sums added up over all blocks (mmc_stats_from_diag)."""
    st = _capi.mmc_stats()
    d = (C.c_double * _capi.MMC_NUM_DIAG)(*[float(x) for x in diag])
    passes = lib.mmc_stats_from_diag(d, float(n_primal), float(n_dual), float(tol), C.byref(st))
    out = {k: v for k, v in st.as_dict().items()
           if k in ("primal_energy", "dual_energy", "gap", "dual_infeasibility", "opt_res_primal", "opt_res_dual",
                    "norm_KTy", "norm_Kx")}
    out["passes"] = bool(passes)
    return out


class SequenceGroupSolver(object):
    """One sequence cut into frame blocks that one host thread drives on one or several GPUs
    (mmc_group_* of the C ABI): what a single-process caller such as the MEX gateway uses to reach
    every GPU of the node while keeping the call shape of MultiframeMotionCoupling.m:977-986.
    Arrays are those of the whole sequence, in the layouts of FrameBlockSolver."""

    def __init__(self, ny, nx, factor, nc, nblocks=1, devices=None, alpha=0.01, kappa=0.25, h=0.0,
                 flow_direction="backward", sampling="average", device=0):
        self._lib = _capi.load_library()
        self._g = C.c_void_p()
        if flow_direction not in _DIRECTIONS:
            raise ValueError("Invalid flow direction.")
        if sampling not in _SAMPLINGS:
            raise MMCError(-4, "sampling must be 'average', 'stride' or 'taps', got '%s'" % sampling)
        p = _capi.mmc_params()
        p.size = C.sizeof(_capi.mmc_params)
        p.version = _capi.MMC_ABI_VERSION
        p.ny, p.nx, p.factor = int(ny), int(nx), int(factor)
        p.nc_global, p.frame_begin, p.frame_end = int(nc), 0, int(nc)
        p.alpha, p.kappa, p.h = float(alpha), float(kappa), float(h)
        p.flow_direction = _DIRECTIONS[flow_direction]
        p.sampling = _SAMPLINGS[sampling]
        p.device = int(device)
        p.stream = None
        ids = None
        if devices is not None:
            if len(devices) != nblocks:
                raise ValueError("one device ordinal per block")
            ids = (C.c_int32 * nblocks)(*[int(d) for d in devices])
        rc = self._lib.mmc_group_create(C.byref(p), int(nblocks), ids, C.byref(self._g))
        if rc != 0:
            raise MMCError(rc, self._lib.mmc_group_last_error(None).decode())
        self.ny, self.nx, self.factor, self.nc = int(ny), int(nx), int(factor), int(nc)
        self.Ny, self.Nx = self.ny * self.factor, self.nx * self.factor
        self.has_w = not (np.isnan(kappa) or kappa == 1)
        self._cb_keepalive = None

    def _check(self, rc):
        if rc != 0:
            raise MMCError(rc, self._lib.mmc_group_last_error(self._g).decode())

    def close(self):
        if getattr(self, "_g", None) is not None and self._g.value:
            self._lib.mmc_group_destroy(self._g)
            self._g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def num_blocks(self):
        return self._lib.mmc_group_num_blocks(self._g)

    def block_range(self, i):
        """(frame_begin, frame_end, device) of block i."""
        a, b, d = C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self._lib.mmc_group_block_range(self._g, int(i), C.byref(a), C.byref(b), C.byref(d)))
        return a.value, b.value, d.value

    def block_sweep_profile(self, i):
        """Summed duration (ms) and count of block i's profiled sweeps (after set_profiling)."""
        hb = self._lib.mmc_group_block_handle(self._g, int(i))
        ms = (C.c_double * 2)()
        n = (C.c_int * 2)()
        rc = self._lib.mmc_get_sweep_profile(hb, ms, n)
        if rc != 0:
            raise MMCError(rc, self._lib.mmc_last_error(hb).decode())
        return {"primal_ms": ms[0], "dual_ms": ms[1], "primal_calls": n[0], "dual_calls": n[1]}

    def set_profiling(self, enable=True):
        for i in range(self.num_blocks):
            hb = self._lib.mmc_group_block_handle(self._g, i)
            self._lib.mmc_set_profiling(hb, int(bool(enable)))

    def set_lowres(self, f):
        f = as_f32(f)
        if f.shape != (self.ny, self.nx, self.nc):
            raise ValueError("low-res sequence must be %s, got %s" % ((self.ny, self.nx, self.nc), f.shape))
        self._check(self._lib.mmc_group_set_lowres(self._g, fptr(f)))

    def set_blur(self, k):
        if k is None or np.size(k) == 0:
            self._check(self._lib.mmc_group_set_blur(self._g, None, 0, 0))
            return
        k = as_f32(np.atleast_2d(k))
        self._check(self._lib.mmc_group_set_blur(self._g, fptr(k), k.shape[0], k.shape[1]))

    def set_sampling_taps(self, first_y, w_y, first_x, w_x):
        fy = np.ascontiguousarray(first_y, dtype=np.int32)
        fx = np.ascontiguousarray(first_x, dtype=np.int32)
        wy = np.ascontiguousarray(w_y, dtype=np.float32)
        wx = np.ascontiguousarray(w_x, dtype=np.float32)
        if fy.shape != (self.ny,) or wy.ndim != 2 or wy.shape[0] != self.ny:
            raise ValueError("row taps must be (%d,) and (%d, T)" % (self.ny, self.ny))
        if fx.shape != (self.nx,) or wx.ndim != 2 or wx.shape[0] != self.nx:
            raise ValueError("column taps must be (%d,) and (%d, T)" % (self.nx, self.nx))
        i32 = C.POINTER(C.c_int32)
        self._check(self._lib.mmc_group_set_sampling_taps(self._g, fy.ctypes.data_as(i32), fptr(wy), wy.shape[1],
                                                          fx.ctypes.data_as(i32), fptr(wx), wx.shape[1]))

    def set_flow(self, v, occ=None):
        if self.nc == 1:
            self._check(self._lib.mmc_group_set_flow(self._g, None, None))
            return
        v = as_f32(v).reshape((self.Ny, self.Nx, -1, 2), order="F")
        if v.shape[2] != self.nc - 1:
            raise ValueError("flow must hold %d pairs, got %d" % (self.nc - 1, v.shape[2]))
        o = None
        if occ is not None:
            o = np.asfortranarray(np.asarray(occ).reshape((self.Ny, self.Nx, self.nc - 1), order="F") != 0,
                                  dtype=np.uint8)
        self._check(self._lib.mmc_group_set_flow(self._g, fptr(v),
                                                 None if o is None else o.ctypes.data_as(C.POINTER(C.c_uint8))))

    def set_state(self, u=None, w=None):
        if u is None:
            if w is not None:
                raise ValueError("w without u")
            self._check(self._lib.mmc_group_set_state(self._g, None, None))
            return
        u = as_f32(u)
        if u.shape != (self.Ny, self.Nx, self.nc):
            raise ValueError("state must be %s, got %s" % ((self.Ny, self.Nx, self.nc), u.shape))
        w = as_f32(u if w is None else w) if self.has_w else None
        self._check(self._lib.mmc_group_set_state(self._g, fptr(u), fptr(w)))

    def set_fixed_frame(self, u0, w0):
        a = None if u0 is None else as_f32(u0)
        b = None if w0 is None else as_f32(w0)
        self._check(self._lib.mmc_group_set_fixed_frame(self._g, fptr(a), fptr(b)))

    def set_flexbox_backward_terms(self, enable=True):
        self._check(self._lib.mmc_group_set_flexbox_backward_terms(self._g, int(bool(enable))))

    def set_diagnostics(self, mode):
        self._check(self._lib.mmc_group_set_diagnostics(self._g, int(mode)))

    def set_step_balance(self, s):
        """One global factor for every block: every tau times s, every sigma divided by s."""
        self._check(self._lib.mmc_group_set_step_balance(self._g, float(s)))

    def set_step_adaptation(self, enable=True):
        """mmc_group_solve chooses the step balance by residual balancing every 50 iterations."""
        self._check(self._lib.mmc_group_set_step_adaptation(self._g, int(bool(enable))))

    def init(self):
        self._check(self._lib.mmc_group_init(self._g))

    @property
    def h(self):
        v = C.c_float()
        self._check(self._lib.mmc_group_get_h(self._g, C.byref(v)))
        return float(v.value)

    @h.setter
    def h(self, value):
        self._check(self._lib.mmc_group_set_h(self._g, float(value)))

    def solve(self, max_iters, tol=0.0, check_every=0, callback=None):
        st = _capi.mmc_stats()
        if callback is not None:
            def _cb(it, stp, _user):
                s = stp.contents
                return int(bool(callback(it, s.as_dict())))
            cb = _capi.PROGRESS_CB(_cb)
        else:
            cb = _capi.PROGRESS_CB()
        self._cb_keepalive = cb
        self._check(self._lib.mmc_group_solve(self._g, int(max_iters), float(tol), int(check_every),
                                              C.byref(st), cb, None))
        return st.as_dict()

    def get_state(self):
        u = np.empty((self.Ny, self.Nx, self.nc), dtype=np.float32, order="F")
        w = np.empty_like(u) if self.has_w else None
        self._check(self._lib.mmc_group_get_state(self._g, fptr(u), fptr(w)))
        return u, w
