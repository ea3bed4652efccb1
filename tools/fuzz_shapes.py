"""Small random configurations of both solves against the scipy oracles: extreme frame shapes (1 x 1
low-resolution pixels upwards), zoom factors 1..5, 1..4 frames, every model and flow direction, masks,
with and without blur.  Prints the worst relative error; exits non-zero on a miss.
usage: python tools/fuzz_shapes.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import warnings

import numpy as np

warnings.filterwarnings('ignore')
import mmc_oracle as orc
import sf_oracle as sfo
from helpers import make_case, make_sf_case, rel_l2
from superresolution_amd.singleframe import SingleFrameSolver
from superresolution_amd.solver import FrameBlockSolver

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst, bad = 0.0, 0
for n in range(cases):
    factor = int(rng.choice([1, 2, 3, 4, 4, 4, 5]))
    big = rng.random() < 0.3                          # some frames span several 64 x 16 tiles
    ny, nx = int(rng.integers(1, 30 if big else 7)), int(rng.integers(1, 30 if big else 7))
    nc = int(rng.integers(1, 5))
    sampling = "average" if factor == 4 and rng.random() < 0.7 else "stride"
    ksz = int(rng.choice([0, 1, 3, 5, 7]))
    k = None if ksz == 0 else orc.fspecial_gaussian(ksz, 0.8)
    with_occ = bool(rng.random() < 0.4)
    flow = str(rng.choice(["smooth", "random", "integer", "zero"]))
    scale = float(rng.choice([0.5, 2.0, 6.0]))
    iters = 25
    desc = "ny=%d nx=%d nc=%d f=%d %s k=%s occ=%d flow=%s/%g" % (ny, nx, nc, factor, sampling, ksz, with_occ, flow, scale)
    if rng.random() < 0.7:
        kappa = float(rng.choice([0.25, 1.0, float("nan"), 0.6]))
        direction = str(rng.choice(["backward", "forward", "forward-backward"]))
        f, v, occ = make_case(int(rng.integers(1 << 30)), ny, nx, nc, factor=factor, flow=flow, flow_scale=scale, with_occ=with_occ)
        P = orc.build_problem(f, v, factor=factor, kappa=kappa, flowDirection=direction, k=k, occ=occ, interpMethod=sampling)
        if not (np.isfinite(P.h) and P.h > 0):          # degenerate start point (constant frames): h is undefined, fix it
            P = orc.build_problem(f, v, factor=factor, kappa=kappa, flowDirection=direction, k=k, occ=occ, interpMethod=sampling, h=1.0)
        x, _, _ = orc.pdhg(P, iters)
        u_ref, w_ref = orc.split_primal(P, x)
        with FrameBlockSolver(ny, nx, factor, nc, alpha=P.alpha, kappa=kappa, h=P.h, flow_direction=direction, sampling=sampling) as s:
            s.set_lowres(f); s.set_blur(k); s.set_flow(v, occ); s.set_state(P.u_up, P.u_up); s.init()
            s.solve(iters)
            u, w = s.get_state()
        err = rel_l2(u, u_ref)
        desc = "mmc kappa=%g %s " % (kappa, direction) + desc
    else:
        nf = nc
        f, v, occ = make_sf_case(int(rng.integers(1 << 30)), ny, nx, nf, factor=factor, flow="smooth" if flow in ("smooth", "zero", "integer") else "random",
                                 flow_scale=scale, with_occ=with_occ)
        delta = float(rng.choice([0.0, 0.01]))
        P = sfo.build_problem(f, v, factor=factor, alpha=0.1, huber_eps=delta, k=k, sampling=sampling, occ=occ)
        x, _, _ = sfo.pdhg(P, iters)
        with SingleFrameSolver(ny, nx, factor, nf, alpha=0.1, huber_eps=delta, sampling=sampling) as s:
            s.set_lowres(f); s.set_blur(k); s.set_flow(v, occ); s.set_state(None); s.init()
            s.solve(iters)
            err = rel_l2(s.get_state(), sfo.image(P, x))
        desc = "sf delta=%g " % delta + desc
    worst = max(worst, err)
    if not err < 1e-4:
        bad += 1
        print("MISS %.3e  %s" % (err, desc), flush=True)
print("cases %d, worst relative error %.3e, misses %d" % (cases, worst, bad))
sys.exit(1 if bad else 0)
