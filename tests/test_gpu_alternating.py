"""The alternating variant (variants/MultiframeMotionCouplingAlternating.m:163-189): solve, new
flows from a caller-supplied estimator, operator swap on the live solver with h and the
primal/dual state kept, PSNR per frame and outer iteration.  Checked against the oracle run through
the same sequence of problems."""
import numpy as np
import pytest

import mmc_oracle as orc
from helpers import make_case, rel_l2

pytestmark = pytest.mark.gpu

KBLUR = orc.fspecial_gaussian(7, np.sqrt(0.6))


def test_outer_loop_matches_oracle_sequence():
    from superresolution_amd import MultiframeMotionCouplingAlternating
    ny, nx, nc, iters = 12, 11, 3, 40
    f, v0, _ = make_case(81, ny, nx, nc, flow_scale=1.0)
    flows = [make_case(82 + k, ny, nx, nc, flow_scale=1.0)[1] for k in range(2)]
    truth = np.clip(np.stack([orc.imresize(f[:, :, i], 4) for i in range(nc)], axis=2), 0, 1)
    calls = []

    def estimator(u, direction):
        assert u.shape == (4 * ny, 4 * nx, nc) and direction == "backward"
        calls.append(u.copy())
        return flows[len(calls) - 1]

    M = MultiframeMotionCouplingAlternating(f, truth, flowEstimator=estimator, flowField=v0, verbose=0)
    M.k = KBLUR
    M.comp_mode = "fast"
    M.outerIts = 2
    M.init()
    M._opts["max_iters"] = iters
    M._opts["tol"] = 0.0
    h0 = M.h
    M.run()
    assert len(calls) == 2 and M.currentIt == 2 and M.psnrVals.shape == (nc, 2)
    assert np.all(np.isfinite(M.psnrVals)) and np.all(M.psnrVals == np.round(M.psnrVals, 2))
    assert M._solver.h == h0                                        # :130-131

    # oracle: iters on v0, then iters on flows[0] from the same primal/dual state and h
    P = orc.build_problem(f, v0, k=KBLUR)
    x, y, _ = orc.pdhg(P, iters)
    assert rel_l2(calls[0], orc.split_primal(P, x)[0]) < 1e-4       # what the estimator saw first
    P2 = orc.build_problem(f, flows[0], k=KBLUR, h=P.h)
    y = np.where(np.asarray(abs(P2.K).sum(axis=1)).ravel() == 0, 0.0, y)
    x2, y2, _ = orc.pdhg(P2, iters, x=x, y=y)
    assert rel_l2(M.result1, orc.split_primal(P2, x2)[0]) < 1e-4
    assert rel_l2(calls[1], orc.split_primal(P2, x2)[0]) < 1e-4
    M.close()


def test_needs_an_estimator():
    from superresolution_amd import MultiframeMotionCouplingAlternating
    f, v0, _ = make_case(83, 8, 8, 2)
    M = MultiframeMotionCouplingAlternating(f, np.zeros((32, 32, 2)), flowField=v0, verbose=0)
    M.k = KBLUR
    M.outerIts = 1
    M.init()
    M._opts["max_iters"] = 5
    with pytest.raises(RuntimeError):
        M.run()
    M.close()
