"""The package's 1-D sampling rows (superresolution_amd/sampling.py) against the oracle's
restatement of helpers/samplingOperator.m, and the banded form handed to the library."""
import numpy as np
import pytest

import mmc_oracle as orc
from superresolution_amd import sampling

CASES = [("nearest", False), ("nearest", True), ("bilinear", False), ("bilinear", True),
         ("bicubic", False), ("bicubic", True), ("lanczos2", False), ("lanczos3", True),
         ("average", False), ("stride", False)]


@pytest.mark.parametrize("method,aa", CASES)
@pytest.mark.parametrize("factor,n_out", [(4, 12), (2, 9), (3, 7)])
def test_rows_match_oracle(method, aa, factor, n_out):
    n_in = factor * n_out
    S = sampling.dim1_matrix(n_in, n_out, method, None, aa)
    m = method
    k = None
    if method == "bilinear" and not aa:
        m = "linear"
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
    with pytest.raises(NotImplementedError):
        sampling.dim1_matrix(12, 4, "pchip")
    with pytest.raises(ValueError):
        sampling.dim1_matrix(12, 4, "spline")
