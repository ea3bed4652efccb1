"""What the per-launch HIP events of mmc_set_profiling cost: the headline workload, 200 iterations with and
without them, alternating.  usage: python tools/profiling_overhead.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superresolution_amd import synthetic
from superresolution_amd.solver import FrameBlockSolver

L, ny, nx = 30, 270, 480
blk = synthetic.make_block(ny, nx, L, 0, L, device="cpu")
with FrameBlockSolver(ny, nx, 4, L) as s:
    s.set_lowres(blk["f"]); s.set_blur(synthetic.gaussian_kernel()); s.set_flow(blk["v"]); s.set_state(); s.init()
    s.solve(100, 0.0, 0); s.sync()
    for rep in range(3):
        for prof in (False, True):
            s.set_profiling(prof)
            s.sync()
            t0 = time.perf_counter()
            s.solve(200, 0.0, 0)
            s.sync()
            print("events %-3s %.4f ms per iteration" % ("on" if prof else "off", (time.perf_counter() - t0) / 200 * 1e3), flush=True)
