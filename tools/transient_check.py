"""Is the slow stretch after the start of a solve tied to the iterate or to the load step?
40 iterations, an idle pause, 40 more iterations on the same state (no reset), under
rocprofv3 --kernel-trace; tools/trace_steps.sh prints the duration of every sweep."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from superresolution_amd import synthetic
from superresolution_amd.solver import FrameBlockSolver

L, pause = 30, float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
ny, nx = 270, 480
blk = synthetic.make_block(ny, nx, L, 0, L, device="cpu")
s = FrameBlockSolver(ny, nx, 4, L, alpha=0.01, kappa=0.25, h=0.0, flow_direction="backward", sampling="average")
s.set_lowres(blk["f"])
s.set_blur(synthetic.gaussian_kernel())
s.set_flow(blk["v"])
s.set_state()
s.init()
s.solve(40, 0.0, 0)
s.sync()
time.sleep(pause)
s.solve(40, 0.0, 0)
s.sync()
s.close()
