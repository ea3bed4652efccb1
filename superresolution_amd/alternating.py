"""MultiframeMotionCouplingAlternating: host-side mirror of the reference's
variant that alternates the super-resolution solve with a re-estimation of the
optical flow on the high-resolution result
(variants/MultiframeMotionCouplingAlternating.m).

Same surface as the MATLAB class: the ground-truth sequence as second
constructor argument (:24-30), `outerIts`, `currentIt`, `psnrVals`, `run`.  The
flow re-estimation (solveV, :45-105) is motionEstimatorClass of the
motionEstimationGUI submodule and stays on the reference path: it is passed in
as a callable `flowEstimator(u, flowDirection) -> v` with u the current
(Ny, Nx, nc) result and v of shape (Ny, Nx, nc-1, 2), component 0 the
displacement along columns (:91-93).  updateMMCsolver (:108-159) swaps the
operator blocks inside the live prost problem and keeps h, the primal and the
dual variables; here that is mmc_set_flow + mmc_init on the live handle.
"""
import numpy as np

from .metrics import central_patch, psnr
from .mmc import MultiframeMotionCoupling


class MultiframeMotionCouplingAlternating(MultiframeMotionCoupling):
    def __init__(self, imageSequenceSmall, imageSequenceLarge, flowEstimator=None, **kw):
        super().__init__(imageSequenceSmall, **kw)
        self.imageSequenceLarge = np.asarray(imageSequenceLarge, dtype=np.float64)   # :26
        self.flowEstimator = flowEstimator
        self.outerIts = 5                                                            # :27
        self.currentIt = 1                                                           # :28
        self.psnrVals = 0                                                            # :29

    def init(self):
        """:33-40"""
        if self.framework != "hip":
            raise RuntimeError("Alternating only in prost for now")                 # :35-37 ('hip' stands in for 'prost')
        self.psnrVals = np.zeros((self._numFrames, int(self.outerIts)))              # :38
        super().init()

    def solveV(self):
        """:45-105: new flow fields from the current high-resolution frames."""
        if self.flowEstimator is None:
            raise RuntimeError("flow re-estimation stays on the reference path: pass flowEstimator(u, flowDirection)")
        if self.verbose > 0:
            print("Calculating new velocity fields")
        Ny, Nx = self._dimsLarge
        v = np.asarray(self.flowEstimator(self._u, self.flowDirection), dtype=np.float64)
        if v.shape != (Ny, Nx, self._numFrames - 1, 2):
            raise ValueError("flowEstimator must return %s, got %s" % ((Ny, Nx, self._numFrames - 1, 2), v.shape))
        self._v = v

    def updateMMCsolver(self):
        """:108-159: the new warp operator replaces the old one in the live solver; h stays (:130-131)."""
        h = self._solver.h
        self._solver.set_flow(self._v, self._occ)
        self._solver.init()
        assert self._solver.h == h
        if self.verbose > 0:
            print("New warp operator constructed")

    def getPSNR(self):
        """:193-202: PSNR of every frame on the patch 20:end-20 after this outer iteration."""
        if self._colorflag:
            self._recomputeRGB()
            res = self.result1
        else:
            res = self._u
        for ii in range(self._numFrames):
            out = central_patch(res[..., ii])
            ref = central_patch(self.imageSequenceLarge[..., ii])
            self.psnrVals[ii, self.currentIt - 1] = round(psnr(out, ref), 2)

    def run(self):
        """:163-189: order v0 - u - v - u - v ..."""
        if self._solver is None:
            raise RuntimeError('No initialization found, run the "init" method first.')
        for jj in range(1, int(self.outerIts) + 1):
            print("Solving problem for u")
            self.currentIt = jj
            self._solveU()
            self.solveV()
            self.updateMMCsolver()
            self.getPSNR()
            print("-------Main Iteration %d finished!" % jj)
        if self._colorflag:
            self._recomputeRGB()
        else:
            self.result1 = self._u
            self.result2 = self._w
        print("Algorithm finished !!")
