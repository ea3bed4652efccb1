"""Random medium-sized configurations of the coupled solve (tens to hundreds of 64 x 16 tiles per frame,
2 to 6 frames) against the C restatement (oracle/mmc_oracle_c.c): every model and flow direction, zoom 4 with
'average' or 'stride', zoom 2, 3, 5, 6 and 8 with 'stride' (row counts that are and are not multiples of 4), smooth / random / integer flows with and without a few
far outliers, occlusion maps, with and without blur.  Prints the worst relative error; exits non-zero on a miss.
usage: python tools/fuzz_medium.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np

import mmc_oracle as orc
import mmc_oracle_c as occ
from helpers import make_case, rel_l2
from superresolution_amd.imresize import upsample_sequence
from superresolution_amd.solver import FrameBlockSolver

if not occ.available():
    sys.exit("C oracle not built")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst, bad = 0.0, 0
for n in range(cases):
    factor = int(rng.choice([4, 4, 4, 2, 2, 3, 3, 5, 6, 8]))
    ny, nx = int(rng.integers(40, 160)) * 4 // factor, int(rng.integers(40, 200)) * 4 // factor
    if factor in (3, 5, 6) and rng.random() < 0.75:
        ny = (ny + 3) // 4 * 4                       # a row count the row-quad sweeps take (zoom factors that do not divide the tile)
    nc = int(rng.integers(2, 7))
    sampling = "average" if factor == 4 and rng.random() < 0.6 else "stride"
    k = orc.fspecial_gaussian(int(rng.choice([3, 5, 7])), 0.8) if rng.random() < 0.8 else None
    kappa = float(rng.choice([0.25, 1.0, float("nan"), 0.6]))
    direction = str(rng.choice(["backward", "forward", "forward-backward"]))
    flow = str(rng.choice(["smooth", "random", "integer"]))
    scale = float(rng.choice([0.7, 2.5, 8.0]))
    with_occ = bool(rng.random() < 0.4)
    f, v, oc = make_case(int(rng.integers(1 << 30)), ny, nx, nc, factor=factor, flow=flow, flow_scale=scale, with_occ=with_occ)
    outliers = rng.random() < 0.4
    if outliers:                                     # a few vectors that leave their tile's neighbourhood by far
        m = rng.random(v.shape[:3]) < 5e-4
        v[m] += rng.choice([-150.0, 90.0, 200.0])
    v = v.astype(np.float32).astype(np.float64)
    u_up = upsample_sequence(f, factor).astype(np.float32)
    iters = 12
    desc = "kappa=%g %s ny=%d nx=%d nc=%d f=%d %s k=%s occ=%d flow=%s/%g outliers=%d" % (
        kappa, direction, ny, nx, nc, factor, sampling, None if k is None else k.shape[0], with_occ, flow, scale, outliers)
    with FrameBlockSolver(ny, nx, factor, nc, kappa=kappa, h=0.0, flow_direction=direction, sampling=sampling) as s:
        s.set_lowres(f); s.set_blur(k); s.set_flow(v, oc); s.set_state(u_up, u_up); s.init()
        h = s.h
        s.solve(iters)
        u, w = s.get_state()
    u_ref, w_ref, _ = occ.solve(f, v, u_up, u_up, iters, k=k, factor=factor, kappa=kappa, flowDirection=direction,
                                sampling=sampling, occ=oc, h=h)
    err = rel_l2(u, u_ref)
    worst = max(worst, err)
    if not err < 2e-5:
        bad += 1
        print("MISS %.3e  %s" % (err, desc))
        sys.stdout.flush()
print("cases %d, worst relative error %.3e, misses %d" % (cases, worst, bad))
sys.exit(1 if bad else 0)
