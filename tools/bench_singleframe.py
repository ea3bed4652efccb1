"""Throughput of the classical single-frame solve (mmc_sf_*) on the reference's sizes:
13 low-resolution frames of 144x176 ('city', 576x704 high-resolution) and of 270x480 (1080p),
synthetic scene and analytic flows from the central frame.  usage: python tools/bench_singleframe.py [iters]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from superresolution_amd import synthetic
from superresolution_amd.singleframe import SingleFrameSolver, central_slice, fspecial_gaussian

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for ny, nx, nf in ((144, 176, 13), (270, 480, 13), (270, 480, 5)):
    blk = synthetic.make_block(ny, nx, nf, 0, nf, device="cpu")
    Ny, Nx = 4 * ny, 4 * nx
    cs = central_slice(nf)
    # flow from the central frame to frame i: the scene translates by (0.7, -0.4) px per frame
    v = np.zeros((Ny, Nx, nf, 2), dtype=np.float32)
    for i in range(nf):
        v[:, :, i, 0] = 0.7 * (i - cs)
        v[:, :, i, 1] = -0.4 * (i - cs)
    with SingleFrameSolver(ny, nx, 4, nf, alpha=0.1, huber_eps=0.01) as s:
        s.set_lowres(blk["f"]); s.set_blur(fspecial_gaussian(7, 0.25 * np.sqrt(15.0))); s.set_flow(v); s.set_state(None)
        t0 = time.perf_counter(); s.init(); t_init = time.perf_counter() - t0
        s.solve(20)
        st = s.solve(iters)
    ms = st["seconds"] / iters * 1e3
    # algorithmic bytes per high-res pixel and iteration: both half-iterations read the flows of every
    # frame (2 nf each) and q (2), the primal one reads p (nf/16), u (1) and writes u, ub (2), the dual one
    # reads ub (1), p, f (2 nf/16) and writes p (nf/16), q (2)
    floats = 4 * nf + 4 * nf / 16.0 + 10
    print("%dx%dx%d: %.3f ms per iteration, %.0f HR Mpx*iter/s, %.0f B/px/iter algorithmic -> %.2f TB/s (%.1f %% of 8 TB/s); init %.3f s; 7500 iterations %.1f s"
          % (Ny, Nx, nf, ms, Ny * Nx / ms / 1e3, 4 * floats, 4 * floats * Ny * Nx / ms / 1e9, 4 * floats * Ny * Nx / ms / 1e9 / 8.0 * 100.0, t_init, ms * 7.5))
