"""MI355X-native super-resolution solve of Multiframe Motion Coupling.

Host-side mirror of the reference's MultiframeMotionCoupling class over the
libmmc_hip C ABI (HIP kernels for gfx950).  Importing this package does not
load the native library; constructing a solver does, and fails loudly if the
library has not been built.
"""
from ._capi import MMCError  # noqa: F401

__all__ = ["MMCError", "FrameBlockSolver", "SequenceGroupSolver", "MultiframeMotionCoupling", "MultiframeMotionCouplingAlternating",
           "SingleFrameSolver", "singleframeMotionSR", "LoadImSequence", "LoadImSequenceForward", "LoadImSequenceReal", "readFlowFile", "writeFlowFile"]


def __getattr__(name):
    if name == "FrameBlockSolver":
        from .solver import FrameBlockSolver
        return FrameBlockSolver
    if name == "SequenceGroupSolver":
        from .solver import SequenceGroupSolver
        return SequenceGroupSolver
    if name == "MultiframeMotionCoupling":
        from .mmc import MultiframeMotionCoupling
        return MultiframeMotionCoupling
    if name == "MultiframeMotionCouplingAlternating":
        from .alternating import MultiframeMotionCouplingAlternating
        return MultiframeMotionCouplingAlternating
    if name in ("SingleFrameSolver", "singleframeMotionSR"):
        from . import singleframe
        return getattr(singleframe, name)
    if name in ("LoadImSequence", "LoadImSequenceForward", "LoadImSequenceReal"):
        from . import loaders
        return getattr(loaders, name)
    if name in ("readFlowFile", "writeFlowFile"):
        from . import flowio
        return getattr(flowio, name)
    raise AttributeError(name)
