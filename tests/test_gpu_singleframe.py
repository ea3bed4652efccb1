"""GPU parity of the classical single-frame solve (mmc_sf_*) against oracle/sf_oracle.py:
operators, adjoint, step sizes, iterates at equal iteration counts, masks, Huber term,
the mirror function, reproducibility.  Bars as for the multiframe solve: relative L2 <= 1e-4.
Parity is unpinned against real prost output (the oracle restates the reference's text)."""
import numpy as np
import pytest

import mmc_oracle as orc
import sf_oracle as sfo
from helpers import make_sf_case, rel_l2

pytestmark = pytest.mark.gpu
K7 = orc.fspecial_gaussian(7, 0.25 * np.sqrt(15.0))


def _solver(f, v, occ=None, alpha=0.1, delta=0.0, k=K7, sampling="average", factor=4):
    from superresolution_amd.singleframe import SingleFrameSolver
    ny, nx, nf = f.shape
    s = SingleFrameSolver(ny, nx, factor, nf, alpha=alpha, huber_eps=delta, sampling=sampling)
    s.set_lowres(f); s.set_blur(k); s.set_flow(v, occ); s.set_state(None); s.init()
    return s


@pytest.mark.parametrize("flow,with_occ", [("smooth", False), ("random", True)])
def test_operators_adjoint_and_steps(flow, with_occ):
    f, v, occ = make_sf_case(11, 9, 11, 5, flow=flow, flow_scale=2.0, with_occ=with_occ)
    P = sfo.build_problem(f, v, alpha=0.1, k=K7, occ=occ)
    rng = np.random.default_rng(0)
    with _solver(f, v, occ) as s:
        u = rng.normal(size=(36, 44))
        p, q = s.apply_K(u)
        ref = P.K @ u.ravel(order="F")
        m = P.n * P.nf
        assert rel_l2(p, ref[:m]) < 1e-5 and rel_l2(q, ref[m:]) < 1e-5
        y = rng.normal(size=P.K.shape[0])
        kt = s.apply_KT(y[:m], y[m:])
        assert rel_l2(kt, P.KT @ y) < 1e-5
        # <K u, y> = <u, K' y> with the device operators alone
        lhs = float(np.dot(np.concatenate([p, q]).astype(np.float64), y.astype(np.float32).astype(np.float64)))
        rhs = float(np.dot(u.ravel(order="F").astype(np.float32).astype(np.float64), kt.astype(np.float64)))
        assert abs(lhs - rhs) < 1e-4 * max(abs(lhs), 1.0)
        tau, sp_, sq = s.steps()
        assert rel_l2(tau, P.T) < 1e-5
        assert rel_l2(sp_, P.S[:m]) < 1e-5
        assert rel_l2(sq, P.S[m:m + P.N]) < 1e-6


@pytest.mark.parametrize("nf,delta,sampling,k", [(5, 0.0, "average", K7), (13, 0.01, "average", K7),
                                                 (3, 0.0, "stride", K7), (4, 0.01, "average", None),
                                                 (1, 0.0, "average", K7), (2, 0.01, "average", K7)])
def test_solve_parity(nf, delta, sampling, k):
    f, v, occ = make_sf_case(20 + nf, 8, 10, nf, flow_scale=1.5, with_occ=(nf == 5))
    P = sfo.build_problem(f, v, alpha=0.1, huber_eps=delta, k=k, sampling=sampling, occ=occ)
    x, _, chg = sfo.pdhg(P, 80)
    with _solver(f, v, occ, delta=delta, k=k, sampling=sampling) as s:
        st = s.solve(80, 0.0, 0)
        u = s.get_state()
    assert st["iterations"] == 80
    assert rel_l2(u, sfo.image(P, x)) < 1e-4
    assert abs(st["res_primal"] - chg[0] / chg[1]) < 1e-3 * chg[0] / chg[1] + 1e-7
    assert abs(st["res_dual"] - chg[2] / chg[3]) < 1e-3 * chg[2] / chg[3] + 1e-7


def test_other_zoom_factor_and_ragged_frame():
    f, v, _ = make_sf_case(31, 7, 9, 3, factor=2)
    k = orc.fspecial_gaussian(5, 0.7)
    P = sfo.build_problem(f, v, factor=2, alpha=0.05, k=k, sampling="stride")
    x, _, _ = sfo.pdhg(P, 60)
    with _solver(f, v, alpha=0.05, k=k, sampling="stride", factor=2) as s:
        s.solve(60)
        assert rel_l2(s.get_state(), sfo.image(P, x)) < 1e-4


def test_violent_flow_and_outliers_take_the_global_scatter():
    """Frames whose source rectangles are too large to walk per tile (random flow of +-12 px, a few
    vectors of +-60 px) go through global atomics; the others through the tile kernel; mixed here."""
    f, v, occ = make_sf_case(34, 16, 24, 5, flow_scale=1.0)
    rng = np.random.default_rng(3)
    v[:, :, 0, :] = rng.normal(0.0, 12.0, v[:, :, 0, :].shape)             # violent everywhere
    out = rng.random(v[:, :, 4, 0].shape) < 0.002                          # outliers in an otherwise smooth frame
    v[:, :, 4, 0][out] += 60.0
    v[:, :, 4, 1][out] -= 40.0
    v = v.astype(np.float32).astype(np.float64)
    P = sfo.build_problem(f, v, alpha=0.1, k=K7)
    x, _, _ = sfo.pdhg(P, 60)
    rngy = np.random.default_rng(1)
    with _solver(f, v) as s:
        y = rngy.normal(size=P.K.shape[0])
        m = P.n * P.nf
        assert rel_l2(s.apply_KT(y[:m], y[m:]), P.KT @ y) < 1e-5
        tau, _, _ = s.steps()
        assert rel_l2(tau, P.T) < 1e-5
        s.solve(60)
        assert rel_l2(s.get_state(), sfo.image(P, x)) < 1e-4


def test_window_too_large_for_lds_takes_the_scratch_plane():
    """Zoom 8: the 16 x 16 low-resolution tile needs a 134 x 134 window, more than the fused dual kernel
    stages; the two-kernel path through the scratch plane runs instead."""
    f, v, _ = make_sf_case(35, 5, 6, 3, factor=8, flow_scale=2.0)
    P = sfo.build_problem(f, v, factor=8, alpha=0.1, k=K7, sampling="stride")
    x, _, _ = sfo.pdhg(P, 50)
    with _solver(f, v, k=K7, sampling="stride", factor=8) as s:
        u = np.random.default_rng(2).normal(size=(40, 48))
        p, q = s.apply_K(u)
        assert rel_l2(p, (P.K @ u.ravel(order="F"))[:P.n * P.nf]) < 1e-5
        s.solve(50)
        assert rel_l2(s.get_state(), sfo.image(P, x)) < 1e-4


def test_start_point_tolerance_stop_and_reproducibility():
    f, v, _ = make_sf_case(32, 8, 8, 5)
    P = sfo.build_problem(f, v, alpha=0.1, k=K7)
    u0 = np.asfortranarray(orc.imresize(f[:, :, 2], 4))
    x, _, _ = sfo.pdhg(P, 50, x=u0.ravel(order="F"))
    outs = []
    for _ in range(2):
        with _solver(f, v) as s:
            s.set_state(u0)
            s.solve(50)
            outs.append(s.get_state())
    assert np.array_equal(outs[0], outs[1])
    assert rel_l2(outs[0], sfo.image(P, x)) < 1e-4
    seen = []
    with _solver(f, v) as s:
        st = s.solve(5000, 1e-3, 50, lambda it, d: seen.append(it) or False)
        assert st["converged"] and st["iterations"] < 5000 and st["iterations"] % 50 == 0
        assert seen and seen[-1] == st["iterations"]
        e0 = sfo.primal_energy(P, np.zeros(P.N))
        assert sfo.primal_energy(P, s.get_state().ravel(order="F").astype(np.float64)) < e0


def test_init_on_a_live_handle_keeps_iterate_and_duals():
    """A second mmc_sf_init (needed after a new flow or blur) recomputes rectangles and steps only."""
    f, v, _ = make_sf_case(36, 8, 8, 5)
    P = sfo.build_problem(f, v, alpha=0.1, k=K7)
    x, _, _ = sfo.pdhg(P, 60)
    with _solver(f, v) as s:
        s.solve(30)
        s.init()
        s.solve(30)
        assert rel_l2(s.get_state(), sfo.image(P, x)) < 1e-4


def test_mirror_function_colour_and_grey():
    from superresolution_amd.singleframe import singleframeMotionSR
    from superresolution_amd.mmc import rgb2ycbcr
    rng = np.random.default_rng(5)
    f, v, _ = make_sf_case(33, 8, 8, 5)
    rgb = np.clip(np.stack([f, 0.9 * f + 0.05, 0.8 * f + 0.1], axis=2) + 0.01 * rng.normal(size=(8, 8, 3, 5)), 0, 1)
    img, timings = singleframeMotionSR(rgb, 4, 0.1, v, comp_mode="fast", verbose=False)
    assert img.shape == (32, 32, 3) and timings.shape == (2,) and timings[1] > 0
    J0 = np.stack([rgb2ycbcr(rgb[:, :, :, i])[:, :, 0] for i in range(5)], axis=2)
    P = sfo.build_problem(J0, v, alpha=0.1, k=K7)
    # the luma of the result is the oracle's iterate after the 'fast' cap of 2500 iterations (:86-90)
    x, _, _ = sfo.pdhg(P, 2500)
    assert rel_l2(rgb2ycbcr(img)[:, :, 0], sfo.image(P, x)) < 2e-3       # through two colour conversions and a clip
    grey, _ = singleframeMotionSR(J0, 4, 0.1, v, huber_eps=0.01, comp_mode="fast", verbose=False)
    Ph = sfo.build_problem(J0, v, alpha=0.1, huber_eps=0.01, k=K7)
    xh, _, _ = sfo.pdhg(Ph, 2500)
    assert rel_l2(grey, sfo.image(Ph, xh)) < 1e-4


def test_argument_errors():
    from superresolution_amd import MMCError
    from superresolution_amd.singleframe import SingleFrameSolver
    with pytest.raises(MMCError):
        SingleFrameSolver(8, 8, 4, 0)
    with pytest.raises(MMCError, match="65535"):
        SingleFrameSolver(8, 8, 4, 70000)
    with pytest.raises(MMCError):
        SingleFrameSolver(8, 8, 4, 3, huber_eps=-1.0)
    with pytest.raises(MMCError):
        SingleFrameSolver(8, 8, 3, 3, sampling="average")         # a block mean only at factor 4
    with SingleFrameSolver(8, 8, 4, 3) as s:
        with pytest.raises(MMCError):
            s.init()                                              # no data yet
        with pytest.raises(MMCError):
            s.set_blur(np.array([[0.5, -0.1, 0.6]]).T @ np.array([[0.2, 0.6, 0.2]]))
        with pytest.raises(MMCError):
            s.set_blur(np.ones((2, 2)))
        with pytest.raises(ValueError):
            s.set_flow(np.zeros((32, 32, 2, 2)))


def test_city_size_13_frames_against_the_oracle():
    """The size the reference runs its classical methods on (test_Mitzel_prost.m:6-14: 'city', 13
    frames, factor 4 -> 576x704 high-resolution), parameters of testfile_full_set_SR_prost.m:52-54
    (alpha 0.1, huber_eps 0.01); synthetic scene, analytic flows from the central frame plus a smooth
    perturbation.  30 iterations at equal counts, then properties of a longer run."""
    from superresolution_amd import synthetic
    from helpers import smooth_field
    ny, nx, nf = 144, 176, 13
    Ny, Nx = 4 * ny, 4 * nx
    blk = synthetic.make_block(ny, nx, nf, 0, nf, device="cpu")
    f = np.asarray(blk["f"], dtype=np.float64)
    rng = np.random.default_rng(7)
    cs = sfo.central_slice(nf)
    v = np.zeros((Ny, Nx, nf, 2))
    for i in range(nf):
        v[:, :, i, 0] = 0.7 * (i - cs) + smooth_field(rng, (Ny, Nx), 0.3)
        v[:, :, i, 1] = -0.4 * (i - cs) + smooth_field(rng, (Ny, Nx), 0.3)
    v = v.astype(np.float32).astype(np.float64)
    P = sfo.build_problem(f, v, alpha=0.1, huber_eps=0.01, k=K7)
    x, _, _ = sfo.pdhg(P, 30)
    with _solver(f, v, delta=0.01) as s:
        s.solve(30)
        u30 = s.get_state()
        assert rel_l2(u30, sfo.image(P, x)) < 1e-4
        assert float(np.abs(u30 - sfo.image(P, x)).max()) < 1e-4
        e30 = sfo.primal_energy(P, u30.ravel(order="F").astype(np.float64))
        st = s.solve(300, 0.0, 100)
        e330 = sfo.primal_energy(P, s.get_state().ravel(order="F").astype(np.float64))
        assert e330 < e30 and st["res_primal"] < 5e-3


def test_full_hd_13_frames_properties():
    """1080 x 1920 x 13, the largest size tools/bench_singleframe.py times, where the scipy oracle is too slow:
    properties that do not depend on the size.  <K u, y> = <u, K' y> for random u, y; the steps are the
    reciprocal absolute row and column sums, i.e. K 1 and K' 1 (no entry of the data rows is negative);
    the primal energy sum_i H_delta(D B W_i u - f_i) + alpha TV(u), formed on the host from K u, falls;
    two runs agree bit for bit."""
    from superresolution_amd import synthetic
    from helpers import smooth_field
    ny, nx, nf = 270, 480, 13
    Ny, Nx = 4 * ny, 4 * nx
    N, n = Ny * Nx, ny * nx
    blk = synthetic.make_block(ny, nx, nf, 0, nf, device="cpu")
    f = np.asarray(blk["f"], dtype=np.float32)
    rng = np.random.default_rng(17)
    cs = sfo.central_slice(nf)
    v = np.zeros((Ny, Nx, nf, 2), dtype=np.float32)
    for i in range(nf):
        v[:, :, i, 0] = 0.7 * (i - cs) + smooth_field(rng, (Ny, Nx), 0.3)
        v[:, :, i, 1] = -0.4 * (i - cs) + smooth_field(rng, (Ny, Nx), 0.3)
    alpha, delta = 0.1, 0.01
    fvec = np.concatenate([f[:, :, i].ravel(order="F") for i in range(nf)]).astype(np.float64)

    def energy(s, u):
        p, q = s.apply_K(u)
        r = np.abs(p.astype(np.float64) - fvec)
        data = np.where(r <= delta, r * r / (2 * delta), r - delta / 2).sum()
        g = q.astype(np.float64).reshape(2, N)
        return data + alpha * np.sqrt((g * g).sum(axis=0)).sum()

    with _solver(f, v, alpha=alpha, delta=delta) as s:
        u = rng.standard_normal(N).astype(np.float32)
        yp = rng.standard_normal(nf * n).astype(np.float32)
        yq = rng.standard_normal(2 * N).astype(np.float32)
        Kp, Kq = s.apply_K(u)
        KTy = s.apply_KT(yp, yq)
        lhs = np.dot(Kp.astype(np.float64), yp) + np.dot(Kq.astype(np.float64), yq)
        rhs = np.dot(u.astype(np.float64), KTy)
        assert abs(lhs - rhs) < 1e-5 * (np.linalg.norm(Kp) * np.linalg.norm(yp) + np.linalg.norm(Kq) * np.linalg.norm(yq))
        tau, sp, sq = s.steps()
        ones_p, _ = s.apply_K(np.ones(N, dtype=np.float32))
        rows = np.where(ones_p > 0, 1.0 / np.where(ones_p > 0, ones_p, 1.0), 1.0)
        assert np.abs(sp - rows).max() < 1e-5
        cols = s.apply_KT(np.ones(nf * n, dtype=np.float32), np.zeros(2 * N, dtype=np.float32)).reshape((Ny, Nx), order="F")
        cc, rr = np.meshgrid(np.arange(Nx), np.arange(Ny))
        cols = cols + (cc > 0) + (cc < Nx - 1) + (rr > 0) + (rr < Ny - 1)       # the gradient's column sums
        assert np.abs(tau.reshape((Ny, Nx), order="F") * cols - 1.0).max() < 1e-5
        s.set_state(None)
        e0 = energy(s, s.get_state())
        s.solve(60)
        u60 = s.get_state()
        e60 = energy(s, u60)
        st = s.solve(240, 0.0, 120)
        e300 = energy(s, s.get_state())
        assert e300 < e60 < e0 and st["res_primal"] < 1e-2
        s.set_state(None)
        s.solve(60)
        assert np.array_equal(s.get_state(), u60)
