"""MultiframeMotionCoupling: host-side mirror of the reference's MATLAB class
(MultiframeMotionCoupling.m) with the super-resolution solve on MI355X.

Same public surface: constructor name/value options (:98-218), public
properties (:31-57), `init` (:221-313), `run` (:316-336), `result1`/`result2`.
The one addition is the framework value 'hip' (the default here), which takes
the place of 'prost': init builds no sparse matrix but hands the raw arrays to
libmmc_hip (init_u_hip), and solveU calls its solve.  framework = 'flexBox'
runs the same library with the presets of init_u_flexBox (:690-902).  Optical-flow estimation
(init_v0, :905-971) is not part of this package: flows are passed in with
'flowField', as variants/testfile_epicFlow.m does.
"""
import math
import numbers
import warnings

import numpy as np

from ._capi import MMCError
from .imresize import imresize_bicubic

_POS_FLOW = ("forward", "backward", "forward-backward")
_POS_FW = ("hip", "prost", "flexBox", "flexBox-vector")
_POS_IM = ("nearest", "bilinear", "bicubic", "bicubic-0", "lanczos2", "lanczos3", "pchip",
           "custom", "average", "stride")
_POS_COMP = ("accurate", "fast", "fastest")


def rgb2ycbcr(rgb):
    """MATLAB rgb2ycbcr for double RGB in [0,1] along axis 2 (:116)."""
    rgb = np.asarray(rgb, dtype=np.float64)
    T = np.array([[65.481, 128.553, 24.966], [-37.797, -74.203, 112.0], [112.0, -93.786, -18.214]]) / 255.0
    off = np.array([16.0, 128.0, 128.0]) / 255.0
    return np.einsum("kc,abc...->abk...", T, rgb) + off.reshape((1, 1, 3) + (1,) * (rgb.ndim - 3))


def ycbcr2rgb(ycc):
    """MATLAB ycbcr2rgb for doubles (:1029)."""
    ycc = np.asarray(ycc, dtype=np.float64)
    T = np.array([[65.481, 128.553, 24.966], [-37.797, -74.203, 112.0], [112.0, -93.786, -18.214]]) / 255.0
    off = np.array([16.0, 128.0, 128.0]) / 255.0
    Ti = np.linalg.inv(T)
    x = ycc - off.reshape((1, 1, 3) + (1,) * (ycc.ndim - 3))
    return np.clip(np.einsum("kc,abc...->abk...", Ti, x), 0.0, 1.0)


def _validatestring(value, allowed):
    """MATLAB validatestring: case-insensitive unique-prefix match, else error."""
    if not isinstance(value, str):
        raise ValueError("Expected input to match one of these values: %s" % ", ".join(allowed))
    hits = [a for a in allowed if a.lower().startswith(value.lower())]
    if value in allowed:
        return value
    if len(hits) == 1:
        return hits[0]
    raise ValueError("Expected input to match one of these values:\n%s\nThe input, '%s', did not match any "
                     "of the valid values." % (", ".join("'%s'" % a for a in allowed), value))


class MultiframeMotionCoupling(object):
    def __init__(self, imageSequenceSmall, zoom_factor=None, flowField=None, warpOp=None,
                 verbose=None, u0_frame=None, w0_frame=None, occlusion=None, device=0):
        seq = np.asarray(imageSequenceSmall, dtype=np.float64)
        if seq.ndim == 3:                                            # :104-108
            self._imageSequenceSmall = seq
            self._colorflag = False
            self._imageSequenceYCbCr = None
            self._numFrames = seq.shape[2]
        elif seq.ndim == 4:                                          # :110-119
            self._colorflag = True
            self._numFrames = seq.shape[3]
            ycc = np.stack([rgb2ycbcr(seq[:, :, :, i]) for i in range(seq.shape[3])], axis=3)
            self._imageSequenceYCbCr = ycc
            self._imageSequenceSmall = ycc[:, :, 0, :]
        else:
            raise ValueError("Load an actual video sequence")        # :122
        self.factor = 4 if zoom_factor is None else zoom_factor      # :126-130
        self._dimsSmall = tuple(seq.shape[:2])
        self._v = None
        self._flowCompute = True
        if flowField is not None and np.ndim(flowField) > 0:          # :145-154
            self._v = np.asarray(flowField, dtype=np.float64)
            self._flowCompute = False
        self.verbose = 1 if verbose is None else verbose             # :157-161
        if warpOp is not None and np.ndim(warpOp) > 0:               # :166-175
            raise MMCError(-4, "a prebuilt sparse 'warpOp' cannot be honoured by the matrix-free HIP path; "
                               "pass 'flowField' (and 'occlusion') instead")
        self._occ = None if occlusion is None else np.asarray(occlusion)
        self._u0_frame = u0_frame                                    # :178-188
        self._w0_frame = w0_frame
        if (u0_frame is None) != (w0_frame is None):
            raise ValueError("Give both fix frames u0 and w0.")     # :190-192
        self.framework = "hip"                                       # reference default: 'prost' (:199)
        self.comp_mode = "accurate"                                  # :200
        self.alpha = 0.01                                            # :207-211
        self.beta = 0.2
        self.kappa = 0.25
        self._h = 1.0
        self.flowDirection = "backward"
        self.interpMethod = "average"                                # :214-216
        self.interpKernel = None
        self.interpAA = False
        self.k = None
        self.result1 = None
        self.result2 = None
        self.device = device
        self._u = None
        self._w = None
        self._solver = None
        self._opts = {}
        self.check_every = 100
        # Not a property of the reference class: how the step sizes are balanced between the primal and the dual side.
        # 1.0 (default): the preconditioners as they are, tau0 = sigma0 = 1 (:349-351); a number s: every tau times s,
        # every sigma divided by s; 'adaptive': the library's residual balancing, its stand-in for prost's
        # 'stepsize','boyd' (:349), whose own rule lives in the absent prost submodule.
        self.stepBalance = 1.0
        self.flexBoxBackwardAsWritten = True     # framework 'flexBox', 'backward': the terms of :864 as they stand
        self.last_stats = None

    # ------------------------------------------------------------------
    @staticmethod
    def checkposScalar(x):
        """:1051-1062"""
        if np.ndim(x) != 0:
            raise ValueError("Input is not scalar")
        if not isinstance(x, numbers.Number) and not isinstance(x, np.generic):
            raise ValueError("Input is not numeric")
        if x <= 0:
            raise ValueError("Input must be > 0")
        return True

    @property
    def h(self):
        return self._h

    def init(self):
        """:221-313"""
        if self.verbose > 0:
            print("Initialization...")
        for x in (self.factor, self.alpha, self.beta, self.kappa):   # :228-231
            self.checkposScalar(x)
        self.flowDirection = _validatestring(self.flowDirection, _POS_FLOW)   # :242-245
        self.framework = _validatestring(self.framework, _POS_FW)
        self.comp_mode = _validatestring(self.comp_mode, _POS_COMP)
        self.interpMethod = _validatestring(self.interpMethod, _POS_IM)
        if self.comp_mode == "fastest" and self.framework not in ("prost", "hip"):    # :248-250
            raise ValueError(' "Fastest" mode can currently only be used with the "prost" framework')
        if self.framework == "prost":                                # :253-258
            raise RuntimeError("prost framework not found on path, or no GPUs found with prost")
        if self.framework == "flexBox-vector":                       # in posFW (:236) but not dispatched (:299-307)
            raise ValueError("Invalid framework string.")
        self._dimsLarge = (int(self.factor) * self._dimsSmall[0], int(self.factor) * self._dimsSmall[1])   # :273
        self._tol_pd = 1e-6 if self.comp_mode == "accurate" else 1e-4                                      # :276-280
        if self.comp_mode == "fastest":                              # :283-290
            if not (isinstance(self.kappa, float) and math.isnan(self.kappa)):
                warnings.warn('You have chosen the "fastest" computation mode.\n'
                              "As such your choice of kappa will be overwritten, \n"
                              "because this mode uses only additive regularization.")
            self.kappa = float("nan")
        if self._flowCompute:                                        # :293-296
            raise MMCError(-4, "optical-flow estimation (init_v0) stays on the reference path: "
                               "construct the object with 'flowField'")
        if self.framework == "flexBox":
            self._init_u_flexBox()
        else:
            self._init_u_hip()
        if self.verbose > 0:
            print("Initialization finished")

    def _init_u_flexBox(self):
        """The problem init_u_flexBox (:690-902) hands to flexBox.runAlgorithm, solved by the HIP path:
        iteration caps 2500 / 800 (:798-802), start point u = bicubic estimate and w = 0 (only the
        first nc of the 2nc primal variables are set, :893-896), 'average' without anti-aliasing is
        the block mean of superpixelOperator for every zoom factor (:707-708), additive model and
        'forward-backward' are the reference's errors (:699-702, :776-778).
        The 'backward' branch of the reference couples u + w in its temporal rows
        ({Id,-Wi,Id,-Wi}, :864) where its 'forward' branch (:834) and the prost path (:490-491)
        couple u - w.  With flexBoxBackwardAsWritten (the default) this path solves what :864
        states; set it to False for the u - w model in both directions."""
        if isinstance(self.kappa, float) and (math.isnan(self.kappa) or self.kappa == 0):
            raise ValueError("Additive warp and full separation not implemented yet for flexBox.")   # :699-702
        if self.flowDirection == "forward-backward":
            raise ValueError("todo")                                                                  # :776-778
        if self.kappa == 1:
            raise MMCError(-4, "init_u_flexBox keeps both variables u and w at kappa = 1 (:805-886); the library's "
                               "kappa = 1 is the reduced model of init_u_prost (:500-513).  Use framework 'hip'.")
        block_mean = self.interpMethod == "average" and not self.interpAA
        as_written = self.flowDirection == "backward" and bool(self.flexBoxBackwardAsWritten)
        self._init_u_hip(max_iters=(2500, 800), w_start_zero=True, block_mean=block_mean, w_plus=as_written)

    def _init_u_hip(self, max_iters=(7500, 2500), w_start_zero=False, block_mean=False, w_plus=False):
        """What init_u_prost (:346-574) does, without the sparse matrices."""
        from .solver import FrameBlockSolver
        ny, nx = self._dimsSmall
        Ny, Nx = self._dimsLarge
        nc = self._numFrames
        iters = max_iters[0] if self.comp_mode == "accurate" else max_iters[1]      # :352-356
        self._opts = {"max_iters": iters, "tol": self._tol_pd}
        v = np.asarray(self._v, dtype=np.float64)
        if v.shape != (Ny, Nx, max(nc - 1, 0), 2) and not (nc == 2 and v.shape == (Ny, Nx, 2)):
            raise ValueError("flowField must be %s, got %s" % ((Ny, Nx, nc - 1, 2), v.shape))
        if self._solver is not None:
            self._solver.close()
        # 'average' at zoom 4 and 'stride' are built into the library; every other sampler of
        # samplingOperator.m (:368-370) is handed over as the rows of its two 1-D factors
        builtin = self.interpMethod == "stride" or (self.interpMethod == "average" and int(self.factor) == 4
                                                    and not self.interpAA)
        s = FrameBlockSolver(ny, nx, int(self.factor), nc, alpha=self.alpha, kappa=self.kappa, h=0.0,
                             flow_direction=self.flowDirection, sampling=self.interpMethod if builtin else "taps",
                             device=self.device)
        if not builtin:
            from .sampling import dim1_taps
            if block_mean:      # superpixelOperator: mean over factor x factor blocks
                f = int(self.factor)
                fy, wy = np.arange(0, Ny, f, dtype=np.int32), np.full((ny, f), 1.0 / f, np.float32)
                fx, wx = np.arange(0, Nx, f, dtype=np.int32), np.full((nx, f), 1.0 / f, np.float32)
            else:
                fy, wy = dim1_taps(Ny, ny, self.interpMethod, self.interpKernel, bool(self.interpAA))
                fx, wx = dim1_taps(Nx, nx, self.interpMethod, self.interpKernel, bool(self.interpAA))
            s.set_sampling_taps(fy, wy, fx, wx)
        s.set_lowres(self._imageSequenceSmall)
        s.set_blur(self.k)                                           # :373-375
        s.set_flow(v, self._occ)                                     # :389-412
        # u_up = imresize(LR, factor) (:415-419) is computed on the device; h is always taken from it
        # (:432-437), whatever the start point is (:446-456)
        s.set_state()
        if nc > 1:
            warpPix, gradPix = s.h_partials()
            if not (warpPix > 0 and gradPix > 0):
                s.close()
                raise MMCError(-1, "warp or gradient energy of the bicubic estimate is zero, h is undefined")
            s.h = warpPix / gradPix
        else:
            s.h = 1.0
        if self._u is not None or self._w is not None or w_start_zero:
            u_up, _ = s.get_state()
            w_default = np.zeros_like(u_up) if w_start_zero else u_up
            s.set_state(u_up if self._u is None else self._u, w_default if self._w is None else self._w)
        if self._u0_frame is not None:                               # :535-572
            if np.isnan(self.kappa) or self.kappa == 1:
                raise MMCError(-1, "u0_frame/w0_frame need the infimal-convolution model (kappa not 1, not NaN)")
            s.set_fixed_frame(self._u0_frame, self._w0_frame)
        if w_plus:
            s.set_flexbox_backward_terms(True)
        s.init()                                                     # masks, h (:431-437), step sizes
        if isinstance(self.stepBalance, str):
            if self.stepBalance != "adaptive":
                s.close()
                raise ValueError("stepBalance must be a positive number or 'adaptive'")
            s.set_step_adaptation(True)
        else:
            s.step_balance = float(self.stepBalance)
        self._h = s.h
        if self.verbose > 0:
            print("Downsampling operator constructed")
            print("Warp operator constructed")
            print("h = %g" % self._h)
        self._solver = s

    def run(self):
        """:316-336"""
        if self._solver is None:
            raise RuntimeError('No initialization found, run the "init" method first.')
        print("Solving problem for u")
        self._solveU()
        print("-------Main Iteration finished!")
        if self._colorflag:
            self._recomputeRGB()
        else:
            self.result1 = self._u
            self.result2 = self._w
        print("Algorithm finished !!")

    def _solveU(self):
        """:977-1004"""
        cb = None
        if self.verbose > 0:
            def cb(it, d):
                # what prost prints 'num_cback_calls' = 5 times per solve (:357-359)
                print("  it %6d  energy %.6e  gap %.3e  res_primal %.3e  res_dual %.3e  (%.2f s)" %
                      (it, d["primal_energy"], d["gap"], d["opt_res_primal"], d["opt_res_dual"], d["seconds"]))
                return False
        every = max(self._opts["max_iters"] // 5, 1) if self.verbose > 0 else self.check_every   # num_cback_calls 5, :358
        every = min(every, self.check_every) if self.check_every > 0 else every
        self.last_stats = self._solver.solve(self._opts["max_iters"], self._opts["tol"], every, cb)
        u, w = self._solver.get_state()
        self._u = u.astype(np.float64)                               # :983
        if w is not None:
            self._w = w.astype(np.float64)                           # :984-986

    def _recomputeRGB(self):
        """:1022-1046"""
        Ny, Nx = self._dimsLarge
        nc = self._numFrames
        f = int(self.factor)
        out = np.zeros((Ny, Nx, 3, nc))
        for i in range(nc):
            ycc = np.stack([self._u[:, :, i],
                            imresize_bicubic(self._imageSequenceYCbCr[:, :, 1, i], f),
                            imresize_bicubic(self._imageSequenceYCbCr[:, :, 2, i], f)], axis=2)
            out[:, :, :, i] = ycbcr2rgb(ycc)
        self.result1 = out
        if self._w is not None:
            out2 = np.zeros((Ny, Nx, 3, nc))
            for i in range(nc):
                ycc = np.stack([self._u[:, :, i] - self._w[:, :, i],
                                imresize_bicubic(self._imageSequenceYCbCr[:, :, 1, i], f),
                                imresize_bicubic(self._imageSequenceYCbCr[:, :, 2, i], f)], axis=2)
                out2[:, :, :, i] = ycbcr2rgb(ycc)
            self.result2 = out2

    def close(self):
        if self._solver is not None:
            self._solver.close()
            self._solver = None
