"""CPU checks of the restatement of the classical single-frame solve (oracle/sf_oracle.py)
against what the reference's text fixes: the problem structure of
classical methods/singleframeMotionSR_mitzel_prost.m:97-163, the central slice (:21, :35-37),
the preconditioning bound, and the behaviour of the iteration."""
import numpy as np
import scipy.sparse.linalg as spla

import mmc_oracle as orc
import sf_oracle as sfo
from helpers import make_sf_case

K7 = orc.fspecial_gaussian(7, 0.25 * np.sqrt(15.0))      # sigm_Unger at factor 4, :78-79


def test_central_slice_is_ceil_half():
    assert [sfo.central_slice(n) for n in (1, 2, 3, 5, 13)] == [0, 0, 1, 2, 6]      # ceil(n/2) - 1


def test_structure_and_central_identity():
    f, v, _ = make_sf_case(1, 5, 6, 3)
    P = sfo.build_problem(f, v, alpha=0.1, k=K7)
    N, n = 20 * 24, 30
    assert P.K.shape == (3 * n + 2 * N, N)
    # the central frame's rows are D*B itself, whatever its flow slice holds
    A = (orc.samplingOperator((20, 24), (5, 6), 'average') @ orc.RepConvMtx(K7, (20, 24))).tocsr()
    assert abs(P.K[n:2 * n] - A).max() < 1e-15
    # the box sampler of :75-77 is interpMethod 'average'
    box = (lambda x: (np.abs(x) <= 2).astype(float) / 4.0, 7.0)
    assert abs(orc.samplingOperator((20, 24), (5, 6), 'custom', box) - orc.samplingOperator((20, 24), (5, 6), 'average')).max() == 0
    # no entry of K's data rows is negative: absolute sums are K 1 and K' 1
    assert P.K[:3 * n].min() >= 0


def test_preconditioning_bound():
    f, v, occ = make_sf_case(2, 5, 5, 3, with_occ=True)
    P = sfo.build_problem(f, v, alpha=0.1, k=K7, occ=occ)
    import scipy.sparse as sp
    M = sp.diags(np.sqrt(P.S)) @ P.K @ sp.diags(np.sqrt(P.T))
    s = spla.svds(M, k=1, return_singular_vectors=False)[0]
    assert s <= 1.0 + 1e-9


def test_energy_decreases_and_huber_smooths():
    f, v, _ = make_sf_case(3, 6, 6, 5, flow_scale=1.0)
    for delta in (0.0, 0.01):
        P = sfo.build_problem(f, v, alpha=0.1, huber_eps=delta, k=K7)
        x0, _, _ = sfo.pdhg(P, 1)
        x1, _, _ = sfo.pdhg(P, 100)
        x2, y2, chg = sfo.pdhg(P, 400)
        e = [sfo.primal_energy(P, x) for x in (x0, x1, x2)]
        assert e[2] < e[1] < e[0]
        assert chg[0] / chg[1] < 5e-3
        m = P.n * P.nf
        assert np.abs(y2[:m]).max() <= 1 + 1e-12
        q = y2[m:].reshape(2, P.N)
        assert np.sqrt((q * q).sum(axis=0)).max() <= P.alpha * (1 + 1e-9)


def test_one_frame_zero_flow_is_tv_deblurring_of_that_frame():
    """nf = 1: the only frame is the central one; the minimiser's down-sampled blur is close to f."""
    f, v, _ = make_sf_case(4, 6, 6, 1)
    P = sfo.build_problem(f, v, alpha=0.005, k=K7)
    x, _, _ = sfo.pdhg(P, 1500)
    r = P.K[:P.n] @ x - P.f
    assert np.abs(r).mean() < 0.01


def test_huber_parameter_also_smooths_the_regulariser():
    """singleframeMotionSR_unger_prost.m:163 passes huber_eps/alpha as the quadratic coefficient of the
    gradient dual: q <- proj_alpha(q~ / (1 + s delta/alpha)), the conjugate of alpha * Huber_delta(|grad u|)."""
    f, v, _ = make_sf_case(9, 6, 6, 3, flow_scale=1.0)
    P = sfo.build_problem(f, v, alpha=0.1, huber_eps=0.05, k=K7)
    m = P.n * P.nf
    rng = np.random.default_rng(0)
    y = rng.normal(0, 0.2, P.K.shape[0])
    S = rng.random(P.K.shape[0]) + 0.1
    S[m:] = np.tile(S[m:m + P.N], 2)
    out = sfo.prox_Fstar(P, y, S)
    # brute force per group: argmin_q  |q - y|^2 / (2 s) + (delta/alpha)/2 |q|^2  s.t. |q| <= alpha
    q = y[m:].reshape(2, P.N) / (1.0 + S[m:m + P.N] * 0.05 / 0.1)
    nrm = np.sqrt((q * q).sum(axis=0))
    q = q * np.minimum(1.0, 0.1 / np.maximum(nrm, 1e-300))
    assert np.allclose(out[m:].reshape(2, P.N), q, atol=1e-14)
    # it changes the solution, and the energy it decreases is the Huber-TV one
    x_h, _, _ = sfo.pdhg(P, 300)
    P0 = sfo.build_problem(f, v, alpha=0.1, huber_eps=0.0, k=K7)
    x_0, _, _ = sfo.pdhg(P0, 300)
    assert np.abs(x_h - x_0).max() > 1e-3
    assert sfo.primal_energy(P, x_h) < sfo.primal_energy(P, x_0) + 1e-12
    g = (P.K @ x_h)[m:].reshape(2, P.N)
    assert sfo.primal_energy(P, x_h) < sfo.primal_energy(P0, x_h)        # Huber <= absolute value, term by term
