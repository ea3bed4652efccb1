"""CPU tests of the oracle itself (oracle/mmc_oracle.py): known answers the
reference's sources imply, structural checks of the restated helpers, and the
committed golden vectors.  PARITY UNPINNED against real prost/flexBox output:
the reference ships no expected values (see oracle header)."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
from scipy.ndimage import convolve

import mmc_oracle as orc
from helpers import make_case, rel_l2

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KBLUR = orc.fspecial_gaussian(7, np.sqrt(0.6))


def test_default_blur_known_values():
    # fspecial('gaussian',7,sqrt(0.6)) of testfile.m:58: centre-row values worked out by hand
    assert np.allclose(KBLUR[3, :4], [1.46706e-4, 9.46258e-3, 0.115278, 0.265252], rtol=2e-5)
    assert abs(KBLUR.sum() - 1) < 1e-15 and np.allclose(KBLUR, KBLUR.T)
    # separable factor
    g = np.exp(-np.arange(-3, 4) ** 2 / 1.2)
    g /= g.sum()
    assert np.allclose(KBLUR, np.outer(g, g), atol=1e-15)
    assert np.allclose(g[:4], [2.85e-4, 0.018373, 0.223829, 0.515026], rtol=2e-3)


def test_average_sampling_is_block_mean_only_for_factor_4():
    # helpers/samplingOperator.m:70-73: every row of the 1-D matrix is [1/4 1/4 1/4 1/4] on pixels 4j-3..4j
    S = orc.samplingOperator([16, 24], [4, 6], "average")
    assert S.shape == (24, 384) and set(np.round(S.data, 12)) == {0.0625}
    X = np.random.default_rng(0).random((16, 24))
    d = (S @ X.ravel(order="F")).reshape(4, 6, order="F")
    assert np.allclose(d, X.reshape(4, 4, 6, 4).mean(axis=(1, 3)))
    # factor 2: a 4-tap overlapping mean with edge replication, not a 2x2 block mean
    S2 = orc._Dim1Matrix(8, 4, "custom", (lambda x: (np.abs(x) <= 2) / 4.0, 7.0), False).toarray().T
    assert np.allclose(S2[0], [0.5, 0.25, 0.25, 0, 0, 0, 0, 0])
    assert np.allclose(S2[1], [0, 0.25, 0.25, 0.25, 0.25, 0, 0, 0])


def test_stride_sampling():
    S = orc.samplingOperator([8, 12], [2, 3], "stride")
    X = np.arange(96, dtype=float).reshape(8, 12)
    assert np.array_equal((S @ X.ravel(order="F")).reshape(2, 3, order="F"), X[::4, ::4])


def test_repconvmtx_is_true_convolution_with_replicate_border():
    rng = np.random.default_rng(1)
    X = rng.random((9, 13))
    k = rng.random((5, 3))          # non-symmetric, non-square
    B = orc.RepConvMtx(k, [9, 13])
    y = (B @ X.ravel(order="F")).reshape(9, 13, order="F")
    assert np.allclose(y, convolve(X, k, mode="nearest"))
    assert np.allclose(np.asarray(B.sum(axis=1)).ravel(), k.sum())
    with pytest.raises(ValueError):
        orc.RepConvMtx(np.ones((4, 3)), [9, 13])       # RepConvMtx.m:6-8


def test_gradient_structure():
    G = orc.spmat_gradient2d(5, 4, 2)         # Nx=5, Ny=4, nc=2
    N, M = 20, 40
    assert G.shape == (2 * M, M)
    X = np.random.default_rng(2).random((4, 5, 2))
    y = G @ X.ravel(order="F")
    dx = y[:M].reshape((4, 5, 2), order="F")
    dy = y[M:].reshape((4, 5, 2), order="F")
    assert np.allclose(dx[:, :-1], X[:, 1:] - X[:, :-1]) and np.all(dx[:, -1] == 0)
    assert np.allclose(dy[:-1], X[1:] - X[:-1]) and np.all(dy[-1] == 0)


def test_warping_operator_rules():
    Ny, Nx = 6, 7
    flow = np.zeros((Ny, Nx, 2))
    W = orc.warpingOperator([Ny, Nx], flow)
    assert (W != sp.identity(Ny * Nx)).nnz == 0           # zero flow = identity, last row/col included
    flow[:, :, 0] = 1.0                                   # +1 column
    W = orc.warpingOperator([Ny, Nx], flow)
    X = np.random.default_rng(3).random((Ny, Nx))
    y = (W @ X.ravel(order="F")).reshape(Ny, Nx, order="F")
    assert np.allclose(y[:, :-1], X[:, 1:]) and np.all(y[:, -1] == 0)     # last column leaves the image
    flow[:, :, 0] = 0.25
    flow[:, :, 1] = -0.5                                  # half a row up
    W = orc.warpingOperator([Ny, Nx], flow)
    rs = np.asarray(W.sum(axis=1)).ravel().reshape(Ny, Nx, order="F")
    assert np.allclose(rs[1:, :-1], 1.0) and np.all(rs[0] == 0) and np.all(rs[:, -1] == 0)
    flow[2, 3, 0] = np.nan
    W = orc.warpingOperator([Ny, Nx], flow)
    assert W[2 + Ny * 3].nnz == 0


@pytest.mark.parametrize("kind", ["I-F", "F-I", "FB"])
def test_construct_warp_block_structure(kind):
    f, v, occ = make_case(5, 3, 4, 4, flow="smooth", flow_scale=1.5, with_occ=True)
    Ny, Nx, nc = 12, 16, 4
    N = Ny * Nx
    W = orc.constructWarpFB(v, occ) if kind == "FB" else orc.constructWarpFF(v, kind, occ)
    assert W.shape == (N * nc, N * nc)
    assert W[N * (nc - 1):].nnz == 0                       # last row block is empty (constructWarpFF.m:20,:61)
    for i in range(nc - 1):
        is_if = kind == "I-F" or (kind == "FB" and (i + 1) % 2 == 1)
        Wi = orc.warpingOperator([Ny, Nx], v[:, :, i, :]).toarray()
        marker = (np.abs(Wi).sum(axis=1) == 0) | (occ[:, :, i].ravel(order="F") != 0)
        Wi[marker] = 0
        Id = np.diag((~marker).astype(float))
        blk = W[N * i:N * (i + 1)].toarray()
        if is_if:
            assert np.allclose(blk[:, N * (i + 1):N * (i + 2)], Wi) and np.allclose(blk[:, N * i:N * (i + 1)], -Id)
        else:
            assert np.allclose(blk[:, N * i:N * (i + 1)], Wi) and np.allclose(blk[:, N * (i + 1):N * (i + 2)], -Id)
        other = np.delete(blk, np.r_[N * i:N * (i + 2)], axis=1)
        assert not other.any()


def test_imresize_properties():
    rng = np.random.default_rng(4)
    assert orc.imresize(np.ones((5, 6)), 4).shape == (20, 24)
    assert np.allclose(orc.imresize(np.full((5, 6), 0.3), 4), 0.3)
    # bicubic reproduces a linear ramp away from the replicated border
    ramp = np.outer(np.arange(12.0), np.ones(9)) * 0.1
    up = orc.imresize(ramp, 4)
    rows = (np.arange(1, 49) / 4.0 + 0.375 - 1) * 0.1      # output row x sits at x/4 + 0.375 (1-based)
    assert np.allclose(up[8:-8, 4], rows[8:-8])
    # anti-aliased shrink of a constant, and agreement of matrix and image forms
    assert np.allclose(orc.imresize(np.full((16, 16), 0.7), 0.25), 0.7)
    X = rng.random((8, 8))
    R = orc.imresize_matrix(8, 32, 4.0)
    assert np.allclose(orc.imresize(X, 4), R @ X @ R.T)


@pytest.mark.parametrize("kappa", [0.25, 1.0, float("nan")])
def test_problem_assembly_and_pdhg_sanity(kappa):
    f, v, occ = make_case(6, 6, 7, 3)
    P = orc.build_problem(f, v, kappa=kappa, k=KBLUR)
    Ny, Nx, nc, ny, nx = P.dims
    M, n = P.M, P.n
    nprim = 2 if len(P.primal) == 2 else 1
    ndual = {0.25: n * nc + 6 * M, 1.0: n * nc + 3 * M}.get(kappa, n * nc + 3 * M)
    assert P.K.shape == (ndual, nprim * M)                 # MultiframeMotionCoupling.m:445-459, :517-518
    assert abs(P.h - P.warpPix / P.gradPix) < 1e-15
    # step sizes satisfy the diagonal-preconditioning bound ||S^1/2 K T^1/2|| <= 1
    A = sp.diags(np.sqrt(P.S)) @ P.K @ sp.diags(np.sqrt(P.T))
    s = sp.linalg.svds(A, k=1, return_singular_vectors=False)[0]
    assert s <= 1.0 + 1e-9
    e0 = orc.primal_energy(P, P.x0)
    x, y, chg = orc.pdhg(P, 200)
    assert orc.primal_energy(P, x) < e0
    u, w = orc.split_primal(P, x)
    assert u.min() >= 0 and u.max() <= 1                   # box constraint, :495
    # float32 iteration stays within the solve tolerance of the float64 one
    x32, _, _ = orc.pdhg(P, 200, dtype=np.float32)
    assert rel_l2(x32[:M], x[:M]) < 1e-4


def test_fixed_first_frame_model():
    f, v, _ = make_case(7, 5, 5, 3)
    u0 = np.full((20, 20), 0.4)
    w0 = np.full((20, 20), 0.1)
    P = orc.build_problem(f, v, k=KBLUR, u0_frame=u0, w0_frame=w0)
    x, y, _ = orc.pdhg(P, 3000)
    u, w = orc.split_primal(P, x)
    assert np.abs(u[:, :, 0] - u0).max() < 5e-3 and np.abs(w[:, :, 0] - w0).max() < 5e-3   # :535-572


def test_golden_synthetic_small():
    g = np.load(os.path.join(GOLD, "synthetic_small.npz"))
    f, v, occ, k = g["f"].astype(np.float64), g["v"].astype(np.float64), g["occ"], g["k"]
    for name, kappa in (("ic", 0.25), ("k1", 1.0), ("add", float("nan"))):
        for tag, direction in (("b", "backward"), ("fb", "forward-backward")):
            P = orc.build_problem(f, v, kappa=kappa, flowDirection=direction, k=k, occ=occ)
            x, y, _ = orc.pdhg(P, 60)
            u, w = orc.split_primal(P, x)
            key = "%s_%s" % (name, tag)
            assert abs(P.h - float(g["h_" + key])) < 1e-12
            assert rel_l2(u, g["u_" + key]) < 1e-6
            if w is not None:
                assert np.abs(w - g["w_" + key]).max() < 1e-6
            assert abs(orc.primal_energy(P, x) - float(g["e_" + key])) < 1e-9 * abs(float(g["e_" + key])) + 1e-9


def test_golden_tube3_crop_super_resolution_beats_bicubic():
    """Config 1 in miniature: tube3 frames 1-3 (256x256 luma crop), zoom 4, IC model."""
    g = np.load(os.path.join(GOLD, "tube3_crop.npz"))
    truth = g["truth_u8"].astype(np.float64) / 255.0
    lr = g["lr"].astype(np.float64)
    v = np.stack([np.stack([orc.imresize(g["flow_lr"][:, :, i, c].astype(np.float64), 4) * 4
                            for c in range(2)], axis=2) for i in range(2)], axis=2)      # :958-959
    assert lr.shape == (64, 64, 3) and v.shape == (256, 256, 2, 2)
    P = orc.build_problem(lr, v, k=KBLUR)
    x, _, _ = orc.pdhg(P, 300)
    u, w = orc.split_primal(P, x)
    psnr_sr = orc.psnr_central(u, truth)
    psnr_bic = orc.psnr_central(P.u_up, truth)
    assert psnr_sr > psnr_bic + 0.5, (psnr_sr, psnr_bic)


def test_preconditioned_iteration_reaches_the_independent_optimum():
    """oracle.pdhg (diagonal preconditioning, primal step first: the rule the HIP sweeps follow) at the
    reference's iteration cap against the bounds of the optimal energy that scalar-step PDHG with restarts
    reached on the same matrices (tests/golden/converged.npz, tests/golden/make_converged.py)."""
    gold = np.load(os.path.join(GOLD, "converged.npz"))
    s = np.load(os.path.join(GOLD, "synthetic_small.npz"))
    for name, kappa, cap in (("ic", 0.25, 7500), ("add", float("nan"), 2500)):
        upper, lower = float(gold["synthetic_%s_energy" % name]), float(gold["synthetic_%s_lower" % name])
        assert 0.0 <= upper - lower <= 1e-6 * upper
        P = orc.build_problem(s["f"], s["v"], k=s["k"], kappa=kappa)
        x, y, _ = orc.pdhg(P, cap)
        e = orc.primal_energy(P, x)
        assert lower * (1 - 1e-9) <= e <= upper * (1 + 1e-4), (lower, e, upper)
        # and the dual side: the gap of the preconditioned iterate brackets the same optimum
        wb = 2.0 * np.abs(x[P.M:]).max() + 1.0 if len(P.primal) == 2 else 0.0
        assert orc.dual_energy(P, y, wb) <= upper * (1 + 1e-9)
