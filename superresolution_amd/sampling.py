"""1-D factors of the reference's separable sampling operator in banded form.

helpers/samplingOperator.m builds S = kron(Sx', Sy) (:77-83) from two 1-D
matrices (Dim1Matrix, :89-146): through interp1 for 'nearest'/'bilinear'/'pchip'/'bicubic-0'
without anti-aliasing (:99-106), through imresize column by column for
'nearest'/'bilinear'/'bicubic'/'lanczos2'/'lanczos3'/'custom' (:108-136), by
index for 'stride' (:137-141); 'average' is the custom kernel
{@(x) double(abs(x)<=2)/4, 7} (:70-73).  `dim1_taps` returns, per low-res
index, the first high-res index and the weights of a fixed number of
consecutive taps, which is what mmc_set_sampling_taps takes.
"""
import numpy as np

METHODS = ("nearest", "bilinear", "bicubic", "bicubic-0", "lanczos2", "lanczos3", "pchip",
           "custom", "average", "stride")


def _cubic(x):
    ax = np.abs(x)
    return np.where(ax <= 1, (1.5 * ax - 2.5) * ax * ax + 1.0,
                    np.where(ax <= 2, ((-0.5 * ax + 2.5) * ax - 4.0) * ax + 2.0, 0.0))


def _triangle(x):
    return np.where((x >= -1) & (x < 0), x + 1.0, np.where((x >= 0) & (x <= 1), 1.0 - x, 0.0))


def _box(x):
    return ((x >= -0.5) & (x < 0.5)).astype(np.float64)


def _lanczos(a):
    def f(x):
        eps = np.finfo(np.float64).eps
        return (a * np.sin(np.pi * x) * np.sin(np.pi * x / a) + eps) / (np.pi ** 2 * x * x + eps) * (np.abs(x) < a)
    return f


_KERNELS = {"nearest": (_box, 1.0), "bilinear": (_triangle, 2.0), "bicubic": (_cubic, 4.0),
            "lanczos2": (_lanczos(2), 4.0), "lanczos3": (_lanczos(3), 6.0)}


def _dense_rows_imresize(n_in, n_out, kernel, width, antialiasing):
    """Rows of imresize(., n_out/n_in) along one axis as a dense (n_out, n_in) matrix: sample x
    (1-based) sits at u = x/scale + (1 - 1/scale)/2, the window starts at floor(u - width/2) and
    holds ceil(width)+2 taps, weights are normalised, indices beyond the edge replicate it."""
    scale = n_out / float(n_in)
    if scale < 1 and antialiasing:
        width = width / scale
        h = lambda x: scale * kernel(scale * x)
    else:
        h = kernel
    x = np.arange(1, n_out + 1, dtype=np.float64)
    u = x / scale + 0.5 * (1.0 - 1.0 / scale)
    left = np.floor(u - width / 2.0)
    idx = left[:, None] + np.arange(int(np.ceil(width)) + 2)[None, :]
    w = h(u[:, None] - idx)
    w = w / w.sum(axis=1, keepdims=True)
    idx = np.clip(idx, 1, n_in).astype(np.int64) - 1
    S = np.zeros((n_out, n_in))
    np.add.at(S, (np.repeat(np.arange(n_out), idx.shape[1]), idx.ravel()), w.ravel())
    return S


def _unit_end_slope(i, n):
    """pchip derivative at the first grid point for the unit vector e_i (0-based i) on n >= 3 points:
    the three-point end formula with its two shape-preserving corrections."""
    y = [1.0 if i == m else 0.0 for m in range(3)]
    d1, d2 = y[1] - y[0], y[2] - y[1]
    e = (3.0 * d1 - d2) / 2.0
    if np.sign(e) != np.sign(d1):
        return 0.0
    if np.sign(d1) != np.sign(d2) and abs(e) > abs(3.0 * d1):
        return 3.0 * d1
    return e


def _interp1_identity_rows(n_in, n_out, factor, method):
    """Rows of interp1(1:n_in, eye(n_in), targetPos, method) for 'pchip' and 'bicubic-0'
    (MATLAB's 'v5cubic'), samplingOperator.m:99-106, targetPos = (1+f/2 : f : n_in+1) - 0.5.
    pchip of a unit vector has zero derivatives at every interior grid point (its secant slopes
    change sign or vanish there), so away from the first and last interval a row is the Hermite
    blend (1-3s^2+2s^3, 3s^2-2s^3) of the two neighbours; the end intervals add the one-sided
    three-point derivative.  'v5cubic' is the cubic convolution of Keys (a = -1/2) with the data
    extended by 3y1 - 3y2 + y3 beyond either end."""
    pos = np.arange(1 + factor / 2.0, n_in + 1 + 1e-12, factor) - 0.5
    S = np.zeros((len(pos), n_in))
    for j, x in enumerate(pos):
        k = min(max(int(np.floor(x)), 1), n_in - 1)          # interval [k, k+1], 1-based
        s = x - k
        if method == "pchip":
            S[j, k - 1] += (1 + 2 * s) * (1 - s) ** 2
            S[j, k] += s * s * (3 - 2 * s)
            if n_in >= 3 and k == 1:                          # derivative at grid point 1
                for i in range(3):
                    S[j, i] += s * (1 - s) ** 2 * _unit_end_slope(i, n_in)
            if n_in >= 3 and k == n_in - 1:                   # derivative at grid point n (mirror image)
                for i in range(3):
                    S[j, n_in - 1 - i] -= s * s * (s - 1) * _unit_end_slope(i, n_in)
            if n_in == 2:                                     # two points: both derivatives are the secant
                S[j, 0] += -(s * (1 - s) ** 2 + s * s * (s - 1))
                S[j, 1] += s * (1 - s) ** 2 + s * s * (s - 1)
        else:
            w = np.array([((2 - s) * s - 1) * s, (3 * s - 5) * s * s + 2, ((4 - 3 * s) * s + 1) * s, (s - 1) * s * s]) / 2.0
            for m, wm in zip(range(k - 1, k + 3), w):         # 1-based grid points k-1 .. k+2
                if 1 <= m <= n_in:
                    S[j, m - 1] += wm
                elif m == 0:                                  # y(0) = 3 y(1) - 3 y(2) + y(3)
                    S[j, 0] += 3 * wm; S[j, 1] -= 3 * wm; S[j, 2] += wm
                else:                                         # y(n+1) = 3 y(n) - 3 y(n-1) + y(n-2)
                    S[j, n_in - 1] += 3 * wm; S[j, n_in - 2] -= 3 * wm; S[j, n_in - 3] += wm
    return S


def dim1_matrix(n_in, n_out, method, kernel=None, antialiasing=False):
    """Dense (n_out, n_in) matrix of Dim1Matrix (samplingOperator.m:89-146), transposed."""
    if method not in METHODS:
        raise ValueError("Expected input to match one of these values: %s" % ", ".join(METHODS))
    if method == "custom" and kernel is None:
        raise ValueError("No custom kernel given")                                   # :57-58
    factor = n_in / float(n_out)
    if method in ("bicubic-0", "pchip"):
        if antialiasing:
            raise ValueError(" The chosen method is not possible together with antialiasing")   # :65-67
        return _interp1_identity_rows(n_in, n_out, factor, method)
    if method == "average":                                                          # :70-73
        method, kernel = "custom", (lambda x: (np.abs(x) <= 2).astype(np.float64) / 4.0, 7.0)
    if method in ("nearest", "bilinear") and not antialiasing:                       # :49-50, :99-106
        pos = np.arange(1 + factor / 2.0, n_in + 1 + 1e-12, factor) - 0.5
        S = np.zeros((len(pos), n_in))
        for j, tp in enumerate(pos):
            if method == "nearest":
                S[j, int(np.floor(tp + 0.5)) - 1] = 1.0
            else:
                i0 = int(np.floor(tp))
                a = tp - i0
                S[j, i0 - 1] += 1.0 - a
                if a > 0:
                    S[j, i0] += a
        return S
    if method == "stride":                                                           # :137-141
        S = np.zeros((n_out, n_in))
        S[np.arange(n_out), np.arange(0, n_in, int(round(factor)))[:n_out]] = 1.0
        return S
    if method == "custom":
        fn, width = kernel
    else:
        fn, width = _KERNELS[method]
    return _dense_rows_imresize(n_in, n_out, fn, float(width), antialiasing)


def banded(S):
    """(first, weights) with weights[R, m] = S[R, first[R] + m] covering every non-zero of row R."""
    n_out, n_in = S.shape
    nz = S != 0
    lo = np.where(nz.any(axis=1), nz.argmax(axis=1), 0)
    hi = np.where(nz.any(axis=1), n_in - 1 - nz[:, ::-1].argmax(axis=1), 0)
    T = int((hi - lo).max()) + 1
    first = np.minimum(lo, n_in - T).astype(np.int32)
    w = np.zeros((n_out, T), dtype=np.float32)
    for m in range(T):
        w[:, m] = S[np.arange(n_out), first + m]
    return first, w


def dim1_taps(n_in, n_out, method, kernel=None, antialiasing=False):
    return banded(dim1_matrix(n_in, n_out, method, kernel, antialiasing))
