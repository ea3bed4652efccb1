"""The whole solve at the headline size: 30 frames of 1080 x 1920 (synthetic scene with known ground truth), the
infimal-convolution model, 7500 iterations as the 'accurate' preset asks (MultiframeMotionCoupling.m:352-356), once
with the fixed step balance 1 and once with residual balancing of it.  Every 500 iterations: primal energy, gap,
optimality residuals, step balance and the PSNR of the central frame against the truth on the 20:end-20 crop
(testfile.m:71-75).  What it shows: float32 sweeps stay stable over the full length (energy falls monotonically at
the checks, the gap closes), and what the step balancing buys at this size.
usage: python tools/long_run.py [iterations]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from superresolution_amd import synthetic
from superresolution_amd.metrics import psnr
from superresolution_amd.solver import FrameBlockSolver

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 7500
L, ny, nx = 30, 270, 480
blk = synthetic.make_block(ny, nx, L, 0, L, device="cpu", with_truth=True)
truth = blk["truth"][20:-20, 20:-20, L // 2].astype(np.float64)
for adapt in (False, True):
    with FrameBlockSolver(ny, nx, 4, L, alpha=0.01, kappa=0.25, h=0.0, flow_direction="backward", sampling="average") as s:
        s.set_lowres(blk["f"]); s.set_blur(synthetic.gaussian_kernel()); s.set_flow(blk["v"]); s.set_state(); s.init()
        s.set_step_adaptation(adapt)
        print("step balancing %s, h = %.4f" % ("on" if adapt else "off", s.h), flush=True)
        t0 = time.perf_counter()
        last = [None]

        def cb(it, st):
            u, _ = s.get_state()
            p = psnr(u[20:-20, 20:-20, L // 2].astype(np.float64), truth)
            flag = "" if last[0] is None or st["primal_energy"] <= last[0] else "  (energy rose)"
            last[0] = st["primal_energy"]
            print("  it %5d  energy %.4f  gap %.4f  res %.3e / %.3e  balance %.3f  PSNR %.3f dB%s"
                  % (it, st["primal_energy"], st["gap"], st["opt_res_primal"], st["opt_res_dual"], st.get("step_balance", 1.0), p, flag), flush=True)
            return False

        st = s.solve(iters, 0.0, 500, cb)
        print("  %d iterations in %.2f s (with the checks and the downloads for the PSNR)" % (st["iterations"], time.perf_counter() - t0), flush=True)
