"""Optimal energies of small problems, reached by a route that shares nothing with the update rule
of the HIP sweeps and of oracle.pdhg(): plain Chambolle-Pock PDHG with one scalar step pair
(tau * sigma * |K|^2 < 1, |K| from a power iteration), dual step first, restarted from the running
average, float64, on the explicit sparse K of oracle/mmc_oracle.py.  Each case runs until its
primal-dual gap is below a fraction TARGET of the energy or its iteration budget is spent; the gap is
taken with the free variable w confined to a box twice as wide as the solution found, so that the
dual objective is a valid lower bound of the optimal energy: the optimum lies in [lower, energy].

Writes tests/golden/converged.npz:
    <case>_energy     primal energy at the end (upper bound of the optimum)
    <case>_lower      dual objective at the end (lower bound)
    <case>_iters      iterations taken
    <case>_psnr       (tube3 cases) PSNR of the central frame on the 20:end-20 crop, testfile.m:71-75
    <case>_u_mid      (tube3 cases) the central frame of u, float32
Cases: the synthetic fixture (tests/golden/synthetic_small.npz) with the three models, and the central
128 x 128 pixels of the tube3 fixture (tests/golden/tube3_crop.npz) with the infimal-convolution and
the additive model.  The tube3 cases are total-variation problems with flat minima: 1.5 million
iterations (about an hour each; the cases run side by side, one process per case) close the gap to
about 1e-5 of the energy.

    python tests/golden/make_converged.py
"""
import multiprocessing
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import mmc_oracle as orc      # noqa: E402

TARGET = 1e-6
# case -> (model kappa, tau/sigma ratio, iteration budget)
CASES = {"synthetic_ic": (0.25, 1.0, 400000), "synthetic_k1": (1.0, 1.0, 400000), "synthetic_add": (float("nan"), 1.0, 400000),
         "tube3_ic": (0.25, 0.01, 1500000), "tube3_add": (float("nan"), 0.01, 1500000)}


def tube3_center():
    """Central 128 x 128 high-res pixels of the tube3 fixture: truth, low-res frames, flows."""
    g = np.load(os.path.join(HERE, "tube3_crop.npz"))
    lo, hi = 16, 48
    truth = g["truth_u8"][4 * lo:4 * hi, 4 * lo:4 * hi].astype(np.float64) / 255.0
    lr = g["lr"][lo:hi, lo:hi].astype(np.float64)
    fl = g["flow_lr"][lo:hi, lo:hi].astype(np.float64)
    v = np.stack([np.stack([orc.imresize(fl[:, :, i, c], 4) * 4 for c in range(2)], axis=2) for i in range(2)], axis=2)
    return truth, lr, v


def problem(case):
    kappa = CASES[case][0]
    if case.startswith("synthetic"):
        s = np.load(os.path.join(HERE, "synthetic_small.npz"))
        return orc.build_problem(s["f"], s["v"], k=s["k"], kappa=kappa), None
    truth, lr, v = tube3_center()
    return orc.build_problem(lr, v, k=orc.fspecial_gaussian(7, np.sqrt(0.6)), kappa=kappa), truth


def bounds(P, x, y):
    wb = 2.0 * np.abs(x[P.M:]).max() + 1.0 if len(P.primal) == 2 else 0.0
    return orc.primal_energy(P, x), orc.dual_energy(P, y, wb)


def run(case):
    P, truth = problem(case)
    _, ratio, budget = CASES[case]
    L = orc.operator_norm(P, iters=300)
    t0 = time.time()
    state = {}

    def gap(x, y):
        e, d = bounds(P, x, y)
        return e - d

    def check(it, x, y):
        e, d = bounds(P, x, y)
        state.update(it=it, e=e, d=d)
        if it % 50000 == 0 or (e - d) <= TARGET * e:
            print("%s  it %7d  energy %.9f  lower %.9f  rel gap %.2e  (%.0f s)" % (case, it, e, d, (e - d) / e, time.time() - t0), flush=True)
        return (e - d) <= TARGET * e

    x, y = orc.pdhg_scalar(P, budget, norm=L, ratio=ratio, callback=check, every=2000, restart_every=2000, score=gap)
    out = {case + "_energy": state["e"], case + "_lower": state["d"], case + "_iters": state["it"]}
    if truth is not None:
        u, _ = orc.split_primal(P, x)
        out[case + "_psnr"] = orc.psnr_central(u, truth)
        out[case + "_u_mid"] = u[:, :, 1].astype(np.float32)
    return out


def main():
    with multiprocessing.Pool(len(CASES)) as pool:
        results = pool.map(run, list(CASES))
    out = {}
    for r in results:
        out.update(r)
    np.savez_compressed(os.path.join(HERE, "converged.npz"), **out)
    for key in sorted(out):
        if np.ndim(out[key]) == 0:
            print(key, out[key])


if __name__ == "__main__":
    main()
