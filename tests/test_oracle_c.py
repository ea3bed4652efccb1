"""The C/OpenMP restatement (oracle/mmc_oracle_c.c, float32) against the explicit
sparse oracle (oracle/mmc_oracle.py, float64): same iterate to float32 accuracy
for every model and flow direction."""
import numpy as np
import pytest

import mmc_oracle as orc
import mmc_oracle_c as occ_c
from helpers import make_case, rel_l2

KBLUR = orc.fspecial_gaussian(7, np.sqrt(0.6))
pytestmark = pytest.mark.skipif(not occ_c.available(), reason="C oracle could not be built (no gcc)")


@pytest.mark.parametrize("direction", ["backward", "forward", "forward-backward"])
@pytest.mark.parametrize("kappa", [0.25, 1.0, float("nan")])
def test_c_restatement_matches_sparse_oracle(direction, kappa):
    f, v, occ = make_case(51, 7, 9, 4, flow="smooth", flow_scale=2.0, with_occ=True)
    P = orc.build_problem(f, v, kappa=kappa, flowDirection=direction, k=KBLUR, occ=occ)
    x, y, chg = orc.pdhg(P, 80)
    u_ref, w_ref = orc.split_primal(P, x)
    u, w, info = occ_c.solve(f, v, P.u_up, P.u_up, 80, k=KBLUR, kappa=kappa, flowDirection=direction, occ=occ, threads=2)
    assert abs(info["h"] - P.h) < 1e-5 * P.h
    assert rel_l2(u, u_ref) < 2e-5
    if w_ref is not None:
        assert np.abs(w - w_ref).max() < 2e-5
    s = info["sums"]
    assert abs(np.sqrt(s[0] / s[1]) - chg[0] / chg[1]) < 2e-2 * chg[0] / chg[1] + 1e-7


def test_c_restatement_variants_and_threads():
    f, v, _ = make_case(52, 6, 6, 3, flow="random", flow_scale=2.0)
    k = np.random.default_rng(0).random((3, 5))
    P = orc.build_problem(f, v, k=k, interpMethod="stride")
    x, _, _ = orc.pdhg(P, 40)
    u1, w1, _ = occ_c.solve(f, v, P.u_up, P.u_up, 40, k=k, sampling="stride", threads=1)
    u4, w4, i4 = occ_c.solve(f, v, P.u_up, P.u_up, 40, k=k, sampling="stride", threads=4)
    assert rel_l2(u1, orc.split_primal(P, x)[0]) < 2e-5
    assert np.array_equal(u1, u4) and np.array_equal(w1, w4)       # gather form: no order dependence
    with pytest.raises(RuntimeError):
        occ_c.solve(f, v, P.u_up, P.u_up, 1, k=np.ones((4, 4)))


@pytest.mark.parametrize("direction,kappa", [("backward", 0.25), ("forward-backward", 0.25), ("forward", float("nan")), ("backward", 1.0)])
def test_matrix_free_mode_matches_the_stored_rows(direction, kappa):
    """The mode that stores no operator (taps recomputed, W' scattered with atomic adds, D*B through
    the composite kernel with the replicate border folded in) against the sparse oracle and the
    stored-row mode; its scatter order depends on the thread schedule, hence a tolerance."""
    f, v, occ = make_case(53, 6, 7, 3, flow="smooth", flow_scale=2.0, with_occ=True)
    P = orc.build_problem(f, v, kappa=kappa, flowDirection=direction, k=KBLUR, occ=occ)
    x, _, _ = orc.pdhg(P, 60)
    u_ref, w_ref = orc.split_primal(P, x)
    u0, w0, i0 = occ_c.solve(f, v, P.u_up, P.u_up, 60, k=KBLUR, kappa=kappa, flowDirection=direction, occ=occ, threads=2)
    u1, w1, i1 = occ_c.solve(f, v, P.u_up, P.u_up, 60, k=KBLUR, kappa=kappa, flowDirection=direction, occ=occ, threads=3,
                             matrix_free=True)
    assert abs(i1["h"] - i0["h"]) < 1e-6 * i0["h"]
    assert rel_l2(u1, u_ref) < 2e-5 and rel_l2(u1, u0) < 1e-5
    if w_ref is not None:
        assert np.abs(w1 - w_ref).max() < 2e-5
    # 'stride' sampling without blur: a one-tap composite kernel
    P2 = orc.build_problem(f, v, kappa=kappa, flowDirection=direction, interpMethod="stride")
    x2, _, _ = orc.pdhg(P2, 30)
    u2, _, _ = occ_c.solve(f, v, P2.u_up, P2.u_up, 30, kappa=kappa, flowDirection=direction, sampling="stride", matrix_free=True)
    assert rel_l2(u2, orc.split_primal(P2, x2)[0]) < 2e-5
