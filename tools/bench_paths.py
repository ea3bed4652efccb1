"""Iteration time of the code paths beside the headline one, on 1080p-sized frames (10 frames each): other
zoom factors, a row count that is not a multiple of 4 (short last row quads, unaligned columns), samplers
handed over as rows.  usage: python tools/bench_paths.py [iters] [case number, to run that one only]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from superresolution_amd import synthetic
from superresolution_amd.sampling import dim1_taps
from superresolution_amd.solver import FrameBlockSolver

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
L = 10
CASES = [("zoom 4, 'average' (headline path)", 4, 270, 480, "average", None, False),
         ("zoom 4, 'stride'", 4, 270, 480, "stride", None, False),
         ("zoom 4, 'bicubic' with anti-aliasing, as rows", 4, 270, 480, "bicubic", None, True),
         ("zoom 4, 'lanczos3', as rows", 4, 270, 480, "lanczos3", None, False),
         ("zoom 2, 'stride'", 2, 540, 960, "stride", None, False),
         ("zoom 2, 'average' (4-tap mean), as rows", 2, 540, 960, "average", None, False),
         ("zoom 3, 'stride', 1080 rows", 3, 360, 640, "stride", None, False),
         ("zoom 3, 'stride', 1083 rows (no row quads)", 3, 361, 640, "stride", None, False),
         ("zoom 8, 'stride'", 8, 135, 240, "stride", None, False),
         ("zoom 4, 'bicubic' without anti-aliasing, as rows", 4, 270, 480, "bicubic", None, False)]
if len(sys.argv) > 2:
    CASES = [CASES[int(sys.argv[2])]]
for name, factor, ny, nx, method, kernel, aa in CASES:
    Ny, Nx = factor * ny, factor * nx
    blk = synthetic.make_block(ny, nx, L, 0, L, factor=factor, device="cpu")
    builtin = method == "stride" or (method == "average" and factor == 4)
    s = FrameBlockSolver(ny, nx, factor, L, sampling=method if builtin else "taps")
    if not builtin:
        fy, wy = dim1_taps(Ny, ny, method, kernel, aa)
        fx, wx = dim1_taps(Nx, nx, method, kernel, aa)
        s.set_sampling_taps(fy, wy, fx, wx)
    s.set_lowres(blk["f"]); s.set_blur(synthetic.gaussian_kernel()); s.set_flow(blk["v"]); s.set_state(); s.init()
    s.solve(20, 0.0, 0)
    st = s.solve(iters, 0.0, 0)
    ms = st["seconds"] / iters * 1e3
    s.set_profiling(True)
    s.solve(iters, 0.0, 0)
    pr = s.get_profile()
    print("%-48s %4dx%4dx%d  %.3f ms per iteration, %6.0f HR Mpx*iter/s  (primal %.3f, dual %.3f)"
          % (name, Ny, Nx, L, ms, Ny * Nx * L / ms / 1e3, pr["primal_ms"], pr["dual_reg_ms"] + pr["dual_data_ms"]))
    sys.stdout.flush()
    s.close()
