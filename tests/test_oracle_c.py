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
