"""Optimal energies of small problems, reached by a route that shares nothing with the update rule
of the HIP sweeps and of oracle.pdhg(): plain Chambolle-Pock PDHG with one scalar step pair
(tau * sigma * |K|^2 < 1, |K| from a power iteration), dual step first, float64, on the explicit
sparse K of oracle/mmc_oracle.py.  Run to a primal-dual gap below 1e-6 of the energy, with the free
variable w confined to a box that is twice as wide as the solution found (so the dual objective is
a valid lower bound of the optimal energy).

Writes tests/golden/converged.npz:
    <case>_energy     primal energy at the end (an upper bound of the optimum)
    <case>_lower      dual objective at the end (a lower bound)
    <case>_iters      iterations taken
    <case>_psnr       (tube3 cases) PSNR of the central frame on the 20:end-20 crop, testfile.m:71-75
    <case>_u_mid      (tube3 cases) the central frame of u, float32
Cases: the synthetic fixture (tests/golden/synthetic_small.npz) with the three models, and the central
128 x 128 pixels of the tube3 fixture (tests/golden/tube3_crop.npz) with the infimal-convolution and
the additive model.  Takes about a quarter of an hour on one core.

    python tests/golden/make_converged.py
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import mmc_oracle as orc      # noqa: E402

REL_GAP = 1e-6


def tube3_center():
    """Central 128 x 128 high-res pixels of the tube3 fixture: truth, low-res frames, flows."""
    g = np.load(os.path.join(HERE, "tube3_crop.npz"))
    lo, hi = 16, 48
    truth = g["truth_u8"][4 * lo:4 * hi, 4 * lo:4 * hi].astype(np.float64) / 255.0
    lr = g["lr"][lo:hi, lo:hi].astype(np.float64)
    fl = g["flow_lr"][lo:hi, lo:hi].astype(np.float64)
    v = np.stack([np.stack([orc.imresize(fl[:, :, i, c], 4) * 4 for c in range(2)], axis=2) for i in range(2)], axis=2)
    return truth, lr, v


def converge(P, label, max_iters=600000, every=5000):
    L = orc.operator_norm(P, iters=300)
    t0 = time.time()
    state = {}

    def check(it, x, y):
        wb = 2.0 * np.abs(x[P.M:]).max() + 1.0 if len(P.primal) == 2 else 0.0
        e = orc.primal_energy(P, x)
        d = orc.dual_energy(P, y, wb)
        state.update(it=it, e=e, d=d)
        print("%s  it %7d  energy %.9f  lower %.9f  rel gap %.2e  (%.0f s)" % (label, it, e, d, (e - d) / e, time.time() - t0), flush=True)
        return (e - d) <= REL_GAP * e

    x, y = orc.pdhg_scalar(P, max_iters, norm=L, callback=check, every=every)
    assert (state["e"] - state["d"]) <= REL_GAP * state["e"], "not converged: raise max_iters"
    return x, state


def main():
    out = {}
    k = orc.fspecial_gaussian(7, np.sqrt(0.6))
    s = np.load(os.path.join(HERE, "synthetic_small.npz"))
    for name, kappa in (("ic", 0.25), ("k1", 1.0), ("add", float("nan"))):
        P = orc.build_problem(s["f"], s["v"], k=s["k"], kappa=kappa)
        x, st = converge(P, "synthetic_" + name)
        out["synthetic_%s_energy" % name] = st["e"]
        out["synthetic_%s_lower" % name] = st["d"]
        out["synthetic_%s_iters" % name] = st["it"]
    truth, lr, v = tube3_center()
    for name, kappa in (("ic", 0.25), ("add", float("nan"))):
        P = orc.build_problem(lr, v, k=k, kappa=kappa)
        x, st = converge(P, "tube3_" + name)
        u, _ = orc.split_primal(P, x)
        out["tube3_%s_energy" % name] = st["e"]
        out["tube3_%s_lower" % name] = st["d"]
        out["tube3_%s_iters" % name] = st["it"]
        out["tube3_%s_psnr" % name] = orc.psnr_central(u, truth)
        out["tube3_%s_u_mid" % name] = u[:, :, 1].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "converged.npz"), **out)
    for key in sorted(out):
        if np.ndim(out[key]) == 0:
            print(key, out[key])


if __name__ == "__main__":
    main()
