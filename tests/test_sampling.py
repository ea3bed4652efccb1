"""The package's 1-D sampling rows (superresolution_amd/sampling.py) against the oracle's
restatement of helpers/samplingOperator.m, and the banded form handed to the library."""
import numpy as np
import pytest

import mmc_oracle as orc
from superresolution_amd import sampling

CASES = [("nearest", False), ("nearest", True), ("bilinear", False), ("bilinear", True),
         ("bicubic", False), ("bicubic", True), ("lanczos2", False), ("lanczos3", True),
         ("average", False), ("stride", False), ("pchip", False), ("bicubic-0", False)]


@pytest.mark.parametrize("method,aa", CASES)
@pytest.mark.parametrize("factor,n_out", [(4, 12), (2, 9), (3, 7)])
def test_rows_match_oracle(method, aa, factor, n_out):
    n_in = factor * n_out
    S = sampling.dim1_matrix(n_in, n_out, method, None, aa)
    m = method
    k = None
    if method == "bilinear" and not aa:
        m = "linear"
    if method == "bicubic-0":
        m = "v5cubic"
    if method == "average":
        m, k = "custom", (lambda x: (np.abs(x) <= 2).astype(np.float64) / 4.0, 7.0)
    ref = orc._Dim1Matrix(n_in, n_out, m, k, aa).toarray().T
    assert np.abs(S - ref).max() < 1e-14
    first, w = sampling.banded(S)
    back = np.zeros_like(S)
    for R in range(n_out):
        for t in range(w.shape[1]):
            back[R, first[R] + t] += w[R, t]
    assert np.abs(back - S).max() < 1e-7
    assert (first >= 0).all() and (first + w.shape[1] <= n_in).all()


def test_average_is_block_mean_only_at_factor_4():
    S4 = sampling.dim1_matrix(16, 4, "average")
    assert np.allclose(S4[1, 4:8], 0.25) and np.count_nonzero(S4[1]) == 4      # samplingOperator.m:70-73
    S2 = sampling.dim1_matrix(16, 8, "average")
    assert np.count_nonzero(S2[3]) == 4 and np.allclose(S2[3].sum(), 1.0)


def test_custom_kernel_and_errors():
    tri = (lambda x: np.maximum(1 - np.abs(x), 0), 2.0)
    S = sampling.dim1_matrix(12, 4, "custom", tri, True)
    assert np.allclose(S.sum(axis=1), 1.0)
    with pytest.raises(ValueError):
        sampling.dim1_matrix(12, 4, "custom")
    with pytest.raises(ValueError):
        sampling.dim1_matrix(12, 4, "pchip", None, True)
    with pytest.raises(ValueError):
        sampling.dim1_matrix(12, 4, "bicubic-0", None, True)
    with pytest.raises(ValueError):
        sampling.dim1_matrix(12, 4, "spline")


def test_interp1_samplers_known_rows():
    """samplingOperator.m:99-106 for 'pchip' and 'bicubic-0' (MATLAB's 'v5cubic').  At zoom 4 the
    samples sit halfway between grid points: pchip of a unit vector has zero slopes there, so its
    rows are the bilinear ones; the cubic convolution gives (-1, 9, 9, -1)/16.  At zoom 2 the first
    sample lies in the end interval, where the one-sided derivatives come in."""
    P = sampling.dim1_matrix(16, 4, "pchip")
    assert np.abs(P - sampling.dim1_matrix(16, 4, "bilinear")).max() < 1e-15
    C = sampling.dim1_matrix(16, 4, "bicubic-0")
    assert np.allclose(C[1, 4:8], np.array([-1, 9, 9, -1]) / 16.0) and np.count_nonzero(C[1]) == 4
    assert np.allclose(C.sum(axis=1), 1.0)
    P2 = sampling.dim1_matrix(8, 4, "pchip")
    assert np.allclose(P2[0, :3], [0.5 + 0.125 * -1.5, 0.5 + 0.125 * 2.0, 0.0])     # slopes -3/2, 2, 0 at the first point
    assert np.allclose(P2[1, 2:4], [0.5, 0.5])
    C2 = sampling.dim1_matrix(8, 4, "bicubic-0")
    # first sample at 1.5 uses y(0) = 3y1 - 3y2 + y3
    w = np.array([-1, 9, 9, -1]) / 16.0
    assert np.allclose(C2[0, :3], [w[1] + 3 * w[0], w[2] - 3 * w[0], w[3] + w[0]])
    # the oracle evaluates interp1 on every column of the identity; a linear ramp is reproduced exactly
    ramp = np.arange(1.0, 9.0)
    assert np.allclose(C2 @ ramp, np.arange(1.5, 8.0, 2.0))


@pytest.mark.parametrize("method,aa", [("bicubic", True), ("lanczos2", False), ("pchip", False), ("nearest", False)])
def test_factors_recovered_from_the_kronecker_product(method, aa):
    """samplingOperator returns S = kron(Sx', Sy) (:77-83) only.  The rows of either 1-D factor sum
    to 1 (imresize normalises, interp1 interpolates constants exactly at these positions), so a caller
    that holds S gets the factors back as block sums: Sy = S(1:ny,:) * kron(1_Nx, I_Ny) and
    Sx' = S(1:ny:end,:) * kron(I_Nx, 1_Ny).  This is what the MATLAB-side patch hands to the gateway."""
    import scipy.sparse as sp
    ny, nx, f = 6, 5, 4
    Ny, Nx = f * ny, f * nx
    S = orc.samplingOperator((Ny, Nx), (ny, nx), method, None, aa).tocsr()
    Sy = (S[:ny, :] @ sp.kron(np.ones((Nx, 1)), sp.identity(Ny))).toarray()
    Sxt = (S[::ny, :] @ sp.kron(sp.identity(Nx), np.ones((Ny, 1)))).toarray()
    assert np.abs(Sy - sampling.dim1_matrix(Ny, ny, method, None, aa)).max() < 1e-12
    assert np.abs(Sxt - sampling.dim1_matrix(Nx, nx, method, None, aa)).max() < 1e-12
