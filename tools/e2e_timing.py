"""What a caller with host buffers pays around the solve: create, upload of the low-res data and the
flows, device-side start point, mmc_init, N iterations, download of u and w.  The bench workload
(30 frames of 1080x1920, IC model).  usage: python tools/e2e_timing.py [iterations]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from superresolution_amd import synthetic
from superresolution_amd.solver import FrameBlockSolver

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 500
L, ny, nx = 30, 270, 480
blk = synthetic.make_block(ny, nx, L, 0, L, device="cpu")
f, v = np.asfortranarray(blk["f"], dtype=np.float32), np.asfortranarray(blk["v"], dtype=np.float32)   # column-major float32, what a MATLAB gateway hands over
k = synthetic.gaussian_kernel()
t = {}
for rep in range(2):           # the second repetition is reported (first: library load, allocator warm-up)
    t0 = time.perf_counter()
    s = FrameBlockSolver(ny, nx, 4, L, alpha=0.01, kappa=0.25, h=0.0, flow_direction="backward", sampling="average")
    t["create"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    s.set_lowres(f); s.set_blur(k); s.set_flow(v); s.set_state()
    s.sync()
    t["upload"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    s.init(); s.sync()
    t["init"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    s.solve(iters, 0.0, 0); s.sync()
    t["solve"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    u, w = s.get_state()
    t["download"] = time.perf_counter() - t0
    s.close()
M = 16 * ny * nx * L
tot = sum(t.values())
print({k_: round(x, 4) for k_, x in t.items()}, "total %.3f s" % tot)
print("host bytes in: %.2f GB (%s), out: %.2f GB" % ((f.nbytes + v.nbytes) / 1e9, v.dtype, (u.nbytes + w.nbytes) / 1e9))
print("solve only      : %.0f HR Mpx*iter/s" % (M * iters / t["solve"] / 1e6))
print("host to host    : %.0f HR Mpx*iter/s at %d iterations" % (M * iters / tot / 1e6, iters))
for n in (2500, 7500):
    tt = tot - t["solve"] + t["solve"] / iters * n
    print("host to host    : %.0f HR Mpx*iter/s at %d iterations (solve time scaled)" % (M * n / tt / 1e6, n))
