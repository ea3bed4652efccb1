"""The model pinned at convergence, independently of the update rule the HIP sweeps and oracle.pdhg()
share (diagonal preconditioning, primal step first): tests/golden/converged.npz holds, per case, an
upper and a lower bound of the optimal energy that plain scalar-step Chambolle-Pock PDHG with restarts
reached in float64 on the explicit sparse matrices (tests/golden/make_converged.py).  This is the
comparison a MATLAB user would make against prost (compare at convergence, not iterate by iterate:
prost balances its steps adaptively, MultiframeMotionCoupling.m:349-363), made here against the
nearest thing available.  Energies of the HIP iterates are evaluated in float64 by the oracle.

What is asserted, and what is not: run long enough, the HIP path reaches the optimal energy to 1e-4
and the PSNR of the independent solution to 0.005 dB.  At the reference's iteration caps
(MultiframeMotionCoupling.m:352-356: 7500 'accurate', 2500 otherwise) the small synthetic problems are
within 1e-4 as well; the tube3 scene (flat total-variation minimum) is within 0.02 dB in PSNR
(testfile.m:71-75) but 3e-3 above the optimal energy with the infimal-convolution model."""
import os
import sys

import numpy as np
import pytest

import mmc_oracle as orc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD)
KBLUR = orc.fspecial_gaussian(7, np.sqrt(0.6))


def _energy(P, u, w):
    x = np.asarray(u, np.float64).ravel(order="F")
    if w is not None and len(P.primal) == 2:
        x = np.concatenate([x, np.asarray(w, np.float64).ravel(order="F")])
    return orc.primal_energy(P, x)


def _solve(P, f, v, k, kappa, checkpoints):
    from superresolution_amd.solver import FrameBlockSolver
    ny, nx, nc = f.shape
    out = []
    with FrameBlockSolver(ny, nx, 4, nc, kappa=kappa, h=P.h) as s:
        s.set_lowres(f); s.set_blur(k); s.set_flow(v); s.set_state(P.u_up, P.u_up); s.init()
        done = 0
        for n in checkpoints:
            s.solve(n - done, 0.0, 0)
            done = n
            u, w = s.get_state()
            assert u.min() >= 0.0 and u.max() <= 1.0        # feasible: its energy cannot be below the optimum
            out.append((u.astype(np.float64), None if w is None else w.astype(np.float64)))
    return out


@pytest.mark.parametrize("name,kappa,cap", [("ic", 0.25, 7500), ("k1", 1.0, 7500), ("add", float("nan"), 2500)])
def test_synthetic_fixture_reaches_the_optimal_energy_at_the_reference_caps(name, kappa, cap):
    gold = np.load(os.path.join(GOLD, "converged.npz"))
    upper, lower = float(gold["synthetic_%s_energy" % name]), float(gold["synthetic_%s_lower" % name])
    assert 0.0 <= upper - lower <= 1e-6 * upper                     # the independent bracket itself
    s = np.load(os.path.join(GOLD, "synthetic_small.npz"))
    P = orc.build_problem(s["f"], s["v"], k=s["k"], kappa=kappa)
    (u, w), = _solve(P, s["f"], s["v"], s["k"], kappa, [cap])
    e = _energy(P, u, w)
    assert lower * (1 - 1e-6) <= e <= upper * (1 + 1e-4), (lower, e, upper)


@pytest.mark.parametrize("name,kappa,cap,cap_tol,long_run", [("ic", 0.25, 7500, 5e-3, 100000), ("add", float("nan"), 2500, 2e-3, 50000)])
def test_tube3_scene_against_the_independent_optimum(name, kappa, cap, cap_tol, long_run):
    import make_converged as mc
    gold = np.load(os.path.join(GOLD, "converged.npz"))
    upper, lower = float(gold["tube3_%s_energy" % name]), float(gold["tube3_%s_lower" % name])
    psnr_star, u_star = float(gold["tube3_%s_psnr" % name]), gold["tube3_%s_u_mid" % name].astype(np.float64)
    assert 0.0 <= upper - lower <= 3e-5 * upper
    truth, lr, v = mc.tube3_center()
    P = orc.build_problem(lr, v, k=KBLUR, kappa=kappa)
    (u_cap, w_cap), (u_long, w_long) = _solve(P, lr, v, KBLUR, kappa, [cap, long_run])
    e_cap, e_long = _energy(P, u_cap, w_cap), _energy(P, u_long, w_long)
    # at the reference's iteration cap
    assert lower * (1 - 1e-6) <= e_cap <= upper * (1 + cap_tol), (lower, e_cap, upper)
    assert abs(orc.psnr_central(u_cap, truth) - psnr_star) < 0.02
    # run on: the optimal energy of the independent solver to 1e-4, its image to 0.005 dB
    assert lower * (1 - 1e-6) <= e_long <= upper * (1 + 1e-4), (lower, e_long, upper)
    assert e_long < e_cap
    assert abs(orc.psnr_central(u_long, truth) - psnr_star) < 0.005
    assert np.sqrt(np.mean((u_long[:, :, 1] - u_star) ** 2)) < 2e-3
