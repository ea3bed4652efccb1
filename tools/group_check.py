"""The single-process group driver (mmc_group_*) on one device: the bench workload as 1, 2, 4 and 8 frame
blocks that share GPU 0.  What it shows is the driver's own cost (events, copy streams, peer copies
that are plain device copies here) and that the iterates do not depend on the cut.
usage: python tools/group_check.py [iters]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from superresolution_amd import synthetic
from superresolution_amd.solver import SequenceGroupSolver

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ny, nx, nc = 270, 480, 32
blk = synthetic.make_block(ny, nx, nc, 0, nc, device="cpu")
ref = None
for nb in (1, 2, 4, 8):
    with SequenceGroupSolver(ny, nx, 4, nc, nblocks=nb, h=0.3) as g:
        g.set_lowres(blk["f"]); g.set_blur(synthetic.gaussian_kernel()); g.set_flow(blk["v"]); g.set_state(); g.init()
        g.solve(20)
        t0 = time.perf_counter()
        st = g.solve(iters)
        dt = time.perf_counter() - t0
        u, w = g.get_state()
    same = ""
    if ref is None:
        ref = (u, w)
    else:
        same = "  iterates equal to one block bit for bit: %s" % (np.array_equal(u, ref[0]) and np.array_equal(w, ref[1]))
    print("%d block(s) of %d frames: %.3f ms per iteration (%.0f HR Mpx*iter/s)%s" % (nb, nc // nb, dt / iters * 1e3, 16 * ny * nx * nc * iters / dt / 1e6, same), flush=True)
