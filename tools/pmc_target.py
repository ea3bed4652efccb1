"""The bench workload with nothing but the library on the GPU (inputs rendered on the host), for
hardware-counter passes: rocprofv3 --pmc <counters> -- python3 tools/pmc_target.py [frames] [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from superresolution_amd import synthetic
from superresolution_amd.imresize import upsample_sequence
from superresolution_amd.solver import FrameBlockSolver

L = int(sys.argv[1]) if len(sys.argv) > 1 else 30
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ny, nx = 270, 480
blk = synthetic.make_block(ny, nx, L, 0, L, device="cpu")
u_up = upsample_sequence(blk["f_state"], 4).astype(np.float32)
s = FrameBlockSolver(ny, nx, 4, L, alpha=0.01, kappa=0.25, h=0.0, flow_direction="backward", sampling="average")
s.set_lowres(blk["f"])
s.set_blur(synthetic.gaussian_kernel())
s.set_flow(blk["v"])
s.set_state(u_up, u_up)
s.init()
st = s.solve(steps, 0.0, 0)
print("iterations", st["iterations"], "h", s.h, flush=True)
s.close()
