"""Random configurations with the samplers of helpers/samplingOperator.m that the library receives as rows of the
two 1-D factors ('bicubic', 'bilinear', 'nearest', 'lanczos2/3', 'pchip', 'bicubic-0', 'average' at zoom 2 and 3, a
custom kernel; with and without anti-aliasing), zoom 2, 3 and 4, with and without blur, frames of a few to a
few dozen tiles: K x, K' y and the step sizes against the explicit sparse operators of oracle/mmc_oracle.py, then
15 iterations against oracle.pdhg().  Prints the worst relative errors; exits non-zero on a miss.
usage: python tools/fuzz_samplers.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np

import mmc_oracle as orc
from helpers import make_case, rel_l2
from superresolution_amd.sampling import dim1_taps
from superresolution_amd.solver import FrameBlockSolver

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
TRI = (lambda x: np.maximum(1 - np.abs(x), 0), 2.0)
METHODS = ["bicubic", "bilinear", "nearest", "lanczos2", "lanczos3", "pchip", "bicubic-0", "average", "custom"]


def split_y(P, y):
    out, off = {}, 0
    for name, size, kind, par in P.dual_blocks:
        out[name] = y[off:off + size]
        off += size
    return out


worst = {"K": 0.0, "KT": 0.0, "tau": 0.0, "solve": 0.0}
bad = 0
for n in range(cases):
    method = str(rng.choice(METHODS))
    factor = int(rng.choice([2, 3]) if method == "average" else rng.choice([2, 3, 4, 4]))
    aa = bool(rng.random() < 0.5) and method not in ("pchip", "bicubic-0", "average")
    kernel = TRI if method == "custom" else None
    ny, nx = int(rng.integers(12, 72)) * 4 // factor, int(rng.integers(12, 60)) * 4 // factor
    nc = int(rng.integers(2, 4))
    k = orc.fspecial_gaussian(int(rng.choice([3, 5, 7])), 0.8) if rng.random() < 0.7 else None
    kappa = float(rng.choice([0.25, 1.0, float("nan")]))
    direction = str(rng.choice(["backward", "forward", "forward-backward"]))
    f, v, _ = make_case(int(rng.integers(1 << 30)), ny, nx, nc, factor=factor, flow="smooth", flow_scale=float(rng.choice([0.7, 2.5])))
    desc = "%s aa=%d f=%d ny=%d nx=%d nc=%d k=%s kappa=%g %s" % (method, aa, factor, ny, nx, nc, None if k is None else k.shape[0], kappa, direction)
    P = orc.build_problem(f, v, factor=factor, k=k, kappa=kappa, flowDirection=direction, interpMethod=method, interpKernel=kernel, interpAA=aa)
    Ny, Nx = P.dims[:2]
    with FrameBlockSolver(ny, nx, factor, nc, alpha=P.alpha, kappa=kappa, h=P.h, flow_direction=direction, sampling="taps") as s:
        fy, wy = dim1_taps(Ny, ny, method, kernel, aa)
        fx, wx = dim1_taps(Nx, nx, method, kernel, aa)
        s.set_sampling_taps(fy, wy, fx, wx)
        s.set_lowres(f); s.set_blur(k); s.set_flow(v); s.set_state(P.u_up, P.u_up); s.init()
        eK = eKT = eT = 0.0
        if kappa == 0.25:                    # the operator checks use the block layout of the infimal-convolution model
            x = rng.normal(size=P.K.shape[1])
            u = x[:P.M].reshape((Ny, Nx, nc), order="F")
            w = x[P.M:].reshape((Ny, Nx, nc), order="F")
            p, g1, g2 = s.debug_apply_K(u, w)
            eK = rel_l2(p, split_y(P, P.K @ x)["p"])
            y = rng.normal(size=P.K.shape[0])
            y[np.asarray(abs(P.K).sum(axis=1)).ravel() == 0] = 0
            yb = split_y(P, y)
            ku, kw = s.debug_apply_KT(yb["p"], yb["g1"], yb["g2"])
            eKT = rel_l2(ku, (P.KT @ y)[:P.M])
            eT = rel_l2(s.debug_steps()[0], P.T[:P.M])
        s.solve(15, 0.0, 0)
        uu, _ = s.get_state()
    xr, _, _ = orc.pdhg(P, 15)
    eS = rel_l2(uu, orc.split_primal(P, xr)[0])
    for name, e in (("K", eK), ("KT", eKT), ("tau", eT), ("solve", eS)):
        worst[name] = max(worst[name], e)
    if not (eK < 3e-6 and eKT < 3e-6 and eT < 2e-6 and eS < 2e-5):
        bad += 1
        print("MISS K %.2e KT %.2e tau %.2e solve %.2e  %s" % (eK, eKT, eT, eS, desc))
        sys.stdout.flush()
print("cases %d, worst relative errors %s, misses %d" % (cases, {k_: "%.2e" % e for k_, e in worst.items()}, bad))
sys.exit(1 if bad else 0)
