"""CPU oracle for the Multiframe Motion Coupling super-resolution solve.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package may import this file;
it is the checker used by tests/, by __graft_entry__.smoke() and by the
cpu_baseline leg of bench.py.

PARITY UNPINNED: the reference (MATLAB) cannot be executed here and holds no
golden vectors, and the arithmetic of its solve lives in three un-vendored
submodules (prost, flexBox, motionEstimationGUI).  This file restates, with
explicit sparse matrices, exactly what the in-tree MATLAB assembles, and fixes
every rule the absent submodules would supply with an explicit definition
(marked DECISION below).  Vectors are MATLAB column-major stacks of frames:
index = r + Ny*c + Ny*Nx*t (0-based), i.e. numpy order='F'.

Restated reference files (relative to the reference checkout):
  helpers/constructWarpFF.m, helpers/constructWarpFB.m, helpers/samplingOperator.m,
  helpers/RepConvMtx.m, MultiframeMotionCoupling.m:346-574 (init_u_prost),
  :415-437 (u_up and h), testfile.m:58,71-75 (default blur, PSNR crop rule).
"""
import numpy as np
import scipy.sparse as sp

F32 = np.float32

# --------------------------------------------------------------------------
# MATLAB toolbox functions the reference calls (restated from their documented
# behaviour; needed to build D, the default blur and the bicubic start point)
# --------------------------------------------------------------------------


def fspecial_gaussian(size, sigma):
    """fspecial('gaussian', size, sigma) as called at testfile.m:58.

    h(i,j) = exp(-(i^2+j^2)/(2 sigma^2)) on the centred integer grid, entries
    below eps*max zeroed, normalised to unit sum.
    """
    if np.isscalar(size):
        size = (int(size), int(size))
    sy, sx = (size[0] - 1) / 2.0, (size[1] - 1) / 2.0
    y, x = np.mgrid[-sy:sy + 1, -sx:sx + 1]
    h = np.exp(-(x * x + y * y) / (2.0 * sigma * sigma))
    h[h < np.finfo(np.float64).eps * h.max()] = 0.0
    s = h.sum()
    if s != 0:
        h = h / s
    return h


def _cubic(x):
    """Keys cubic convolution kernel, a = -0.5 (imresize 'bicubic')."""
    ax = np.abs(x)
    ax2, ax3 = ax * ax, ax * ax * ax
    return ((1.5 * ax3 - 2.5 * ax2 + 1.0) * (ax <= 1) +
            (-0.5 * ax3 + 2.5 * ax2 - 4.0 * ax + 2.0) * ((ax > 1) & (ax <= 2)))


def _triangle(x):
    return (x + 1) * ((x >= -1) & (x < 0)) + (1 - x) * ((x >= 0) & (x <= 1))


def _box(x):
    return ((x >= -0.5) & (x < 0.5)).astype(np.float64)


def _lanczos(a):
    def f(x):
        eps = np.finfo(np.float64).eps
        v = (a * np.sin(np.pi * x) * np.sin(np.pi * x / a) + eps) / ((np.pi ** 2) * x * x + eps)
        return v * (np.abs(x) < a)
    return f


_IMRESIZE_KERNELS = {
    'nearest': (_box, 1.0), 'box': (_box, 1.0),
    'bilinear': (_triangle, 2.0), 'triangle': (_triangle, 2.0),
    'bicubic': (_cubic, 4.0), 'cubic': (_cubic, 4.0),
    'lanczos2': (_lanczos(2), 4.0), 'lanczos3': (_lanczos(3), 6.0),
}


def imresize_contributions(in_length, out_length, scale, kernel, kernel_width, antialiasing):
    """Weights and (0-based, clamped) indices of imresize along one dimension.

    Follows the contribution rule of MATLAB's imresize: output sample x (1-based)
    sits at u = x/scale + 0.5*(1 - 1/scale) in input coordinates, the window
    starts at floor(u - kernel_width/2) and holds ceil(kernel_width)+2 taps,
    weights are normalised to unit sum and out-of-range indices are clamped
    (edge replication).
    """
    if scale < 1 and antialiasing:
        h = lambda x: scale * kernel(scale * x)
        kernel_width = kernel_width / scale
    else:
        h = kernel
    x = np.arange(1, out_length + 1, dtype=np.float64)
    u = x / scale + 0.5 * (1.0 - 1.0 / scale)
    left = np.floor(u - kernel_width / 2.0)
    P = int(np.ceil(kernel_width)) + 2
    indices = left[:, None] + np.arange(P)[None, :]
    weights = h(u[:, None] - indices)
    weights = weights / weights.sum(axis=1, keepdims=True)
    indices = np.clip(indices, 1, in_length).astype(np.int64) - 1
    return weights, indices


def imresize_matrix(in_length, out_length, scale, method='bicubic', antialiasing=True):
    """Dense out_length x in_length matrix of imresize along one dimension."""
    if isinstance(method, (tuple, list)):
        kernel, width = method
    else:
        kernel, width = _IMRESIZE_KERNELS[method]
    w, idx = imresize_contributions(in_length, out_length, scale, kernel, width, antialiasing)
    R = np.zeros((out_length, in_length))
    for j in range(w.shape[1]):
        np.add.at(R, (np.arange(out_length), idx[:, j]), w[:, j])
    return R


def imresize(img, scale, method='bicubic', antialiasing=True):
    """imresize(img, scale[, method]) for a 2-D (or 2-D x channels) array.

    Output size is ceil(scale*size) as in MATLAB.  Used at
    MultiframeMotionCoupling.m:417-418 (u_up = imresize(LR, factor), bicubic).
    """
    img = np.asarray(img, dtype=np.float64)
    ny, nx = img.shape[:2]
    oy, ox = int(np.ceil(ny * scale)), int(np.ceil(nx * scale))
    Ry = imresize_matrix(ny, oy, scale, method, antialiasing)
    Rx = imresize_matrix(nx, ox, scale, method, antialiasing)
    out = np.tensordot(Ry, img, axes=(1, 0))
    out = np.moveaxis(np.tensordot(Rx, out, axes=(1, 1)), 0, 1)
    return out


# --------------------------------------------------------------------------
# helpers/samplingOperator.m
# --------------------------------------------------------------------------


def _pchip_slopes(y):
    """Derivatives of MATLAB's pchip on the grid 1..n (Fritsch-Carlson; unit spacing): zero where
    the secant slopes change sign or vanish, their harmonic mean otherwise; the three-point,
    shape-preserving end formula of pchip.m (pchipend)."""
    n = len(y)
    del_ = np.diff(y)
    d = np.zeros(n)
    if n == 2:
        d[:] = del_[0]
        return d
    for k in range(1, n - 1):
        if del_[k - 1] * del_[k] > 0:
            d[k] = 2.0 * del_[k - 1] * del_[k] / (del_[k - 1] + del_[k])

    def end(d1, d2):
        e = (3.0 * d1 - d2) / 2.0
        if np.sign(e) != np.sign(d1):
            e = 0.0
        elif np.sign(d1) != np.sign(d2) and abs(e) > abs(3.0 * d1):
            e = 3.0 * d1
        return e
    d[0] = end(del_[0], del_[1])
    d[-1] = end(del_[-1], del_[-2])
    return d


def _interp1_uniform(y, xi, method):
    """interp1(1:n, y, xi, method) for xi inside [1, n], method 'pchip' or 'v5cubic'
    (samplingOperator.m:99-106 calls it on the columns of the identity).
    'v5cubic': cubic convolution of Keys (a = -1/2) on the data extended at either end by
    3*y(1) - 3*y(2) + y(3) (and its mirror image), as MATLAB documents for uniformly spaced points."""
    n = len(y)
    xi = np.asarray(xi, dtype=np.float64)
    k = np.clip(np.floor(xi).astype(int), 1, n - 1)          # interval [k, k+1], 1-based
    s = xi - k
    if method == 'pchip':
        d = _pchip_slopes(np.asarray(y, dtype=np.float64))
        y0, y1, d0, d1 = y[k - 1], y[k], d[k - 1], d[k]
        h00 = (1 + 2 * s) * (1 - s) ** 2
        h10 = s * (1 - s) ** 2
        h01 = s * s * (3 - 2 * s)
        h11 = s * s * (s - 1)
        return h00 * y0 + h10 * d0 + h01 * y1 + h11 * d1
    if method == 'v5cubic':
        ye = np.concatenate([[3 * y[0] - 3 * y[1] + y[2]], y, [3 * y[-1] - 3 * y[-2] + y[-3]]])
        # ye[m] = y(m) in 1-based numbering of y, m = 0 .. n+1
        return (((2 - s) * s - 1) * s * ye[k - 1] + ((3 * s - 5) * s * s + 2) * ye[k]
                + ((4 - 3 * s) * s + 1) * s * ye[k + 1] + (s - 1) * s * s * ye[k + 2]) / 2.0
    raise ValueError(method)


def _Dim1Matrix(inputDim, targetDim, method, kernel, antialiasing):
    """samplingOperator.m:89-146.  Returns S with shape inputDim x targetDim."""
    factor = inputDim / targetDim
    if method in ('nearest', 'linear', 'pchip', 'v5cubic') and not antialiasing:
        # samplingOperator.m:99-106 (interp1 with the identity matrix)
        targetPos = np.arange(1 + factor / 2.0, inputDim + 1 + 1e-12, factor) - 0.5
        if method in ('pchip', 'v5cubic'):
            # interp1(1:n, eye(n), targetPos, method): every column of the identity on its own
            Y = np.eye(inputDim)
            cols = [_interp1_uniform(Y[:, i], targetPos, method) for i in range(inputDim)]
            return sp.csr_matrix(np.stack(cols, axis=0))
        S = np.zeros((inputDim, len(targetPos)))
        for j, tp in enumerate(targetPos):
            if method == 'nearest':
                S[int(np.floor(tp + 0.5)) - 1, j] = 1.0
            else:
                i0 = int(np.floor(tp))
                a = tp - i0
                S[i0 - 1, j] += 1 - a
                if a > 0:
                    S[i0, j] += a
        return sp.csr_matrix(S)
    if (method in ('nearest', 'bilinear', 'bicubic', 'lanczos2', 'lanczos3')
            or (antialiasing and method != 'custom')):
        # samplingOperator.m:108-121: row ii = imresize(e_ii, 1/factor, method)
        R = imresize_matrix(inputDim, targetDim, 1.0 / factor, method, antialiasing)
        return sp.csr_matrix(R.T)
    if method == 'custom':
        # samplingOperator.m:123-136
        R = imresize_matrix(inputDim, targetDim, 1.0 / factor, kernel, antialiasing)
        return sp.csr_matrix(R.T)
    if method == 'stride':
        # samplingOperator.m:137-141
        m = np.arange(0, inputDim, int(round(factor)))[:targetDim]
        return sp.csr_matrix((np.ones(len(m)), (m, np.arange(len(m)))), shape=(inputDim, targetDim))
    raise ValueError(method)


def samplingOperator(inputDims, targetDims, method, kernel=None, antialiasing=False):
    """helpers/samplingOperator.m:1-85.  S = kron(Sx', Sy), shape n x N."""
    valid = ('nearest', 'bilinear', 'bicubic', 'bicubic-0', 'lanczos2', 'lanczos3',
             'pchip', 'custom', 'average', 'stride')
    if method not in valid:
        raise ValueError("unknown sampling method '%s'" % method)
    if method == 'bilinear' and not antialiasing:
        method = 'linear'                      # :49-50
    elif method == 'bicubic-0' and not antialiasing:
        method = 'v5cubic'                     # :51-52
    if method == 'custom' and kernel is None:
        raise ValueError('No custom kernel given')          # :57-58
    if antialiasing and method in ('v5cubic', 'pchip'):
        raise ValueError('The chosen method is not possible together with antialiasing')  # :65-67
    if method == 'average':
        # :70-73  kernel = {@(x) double(abs(x)<=2)/4, 7}
        method = 'custom'
        kernel = (lambda x: (np.abs(x) <= 2).astype(np.float64) / 4.0, 7.0)
    Sx = _Dim1Matrix(inputDims[1], targetDims[1], method, kernel, antialiasing)      # :78
    Sy = _Dim1Matrix(inputDims[0], targetDims[0], method, kernel, antialiasing).T    # :79
    return sp.kron(Sx.T, Sy, format='csr')                                           # :83


# --------------------------------------------------------------------------
# helpers/RepConvMtx.m
# --------------------------------------------------------------------------


def RepConvMtx(k, dimsLarge):
    """helpers/RepConvMtx.m:1-41: true 2-D convolution, replicate border.

    (B u)(r,c) = sum_{i,j} k[kc_r - i, kc_c - j] * u(clamp(r+i), clamp(c+j)).
    Note that inside the reference function `Nx` is the ROW count (:10).
    """
    k = np.asarray(k, dtype=np.float64)
    if k.shape[0] % 2 != 1 or k.shape[1] % 2 != 1:
        raise ValueError('kernel has even dimensions (which is not implemented)')   # :6-8
    Nr, Ncol = int(dimsLarge[0]), int(dimsLarge[1])
    N = Nr * Ncol
    kx, ky = k.shape
    kex, key = kx // 2, ky // 2
    r = np.arange(N) % Nr
    c = np.arange(N) // Nr
    rows, cols, vals = [], [], []
    for i in range(-kex, kex + 1):
        for j in range(-key, key + 1):
            rr = np.clip(r + i, 0, Nr - 1)          # :29 north/south border
            cc = np.clip(c + j, 0, Ncol - 1)        # :30-32 west/east border
            rows.append(np.arange(N))
            cols.append(rr + Nr * cc)
            vals.append(np.full(N, k[kex - i, key - j]))    # :33 k(kcx-i, kcy-j)
    K = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                      shape=(N, N))
    return K.tocsr()                                 # duplicates summed, :38


# --------------------------------------------------------------------------
# Operators from the absent submodules, fixed by explicit DECISIONS
# --------------------------------------------------------------------------


def warp_taps(Ny, Nx, flowField):
    """Bilinear sampling taps for warpingOperator (see below).

    Returns (valid, r0, r1, c0, c1, a, b) per pixel, all of shape (Ny, Nx).
    Position arithmetic is float32 so that an fp32 device kernel reaches the
    same floor()/mask decisions.
    """
    fx = np.asarray(flowField[:, :, 0], dtype=F32)
    fy = np.asarray(flowField[:, :, 1], dtype=F32)
    rr, cc = np.meshgrid(np.arange(Ny, dtype=F32), np.arange(Nx, dtype=F32), indexing='ij')
    y = rr + fy
    x = cc + fx
    with np.errstate(invalid='ignore'):
        valid = (y >= 0) & (y <= F32(Ny - 1)) & (x >= 0) & (x <= F32(Nx - 1))
    y = np.where(valid, y, F32(0))
    x = np.where(valid, x, F32(0))
    y0 = np.floor(y)
    x0 = np.floor(x)
    a = (y - y0).astype(F32)
    b = (x - x0).astype(F32)
    r0 = y0.astype(np.int64)
    c0 = x0.astype(np.int64)
    r1 = np.minimum(r0 + 1, Ny - 1)
    c1 = np.minimum(c0 + 1, Nx - 1)
    return valid, r0, r1, c0, c1, a, b


def warpingOperator(dims, flowField):
    """warpingOperator(dims, flowField) [motionEstimationGUI, not in checkout].

    Call sites: helpers/constructWarpFF.m:25, helpers/constructWarpFB.m:20,
    MultiframeMotionCoupling.m:737.

    DECISION (interpolation): bilinear.  Row (r,c) samples the image at
    (r + flow(:,:,2), c + flow(:,:,1)): flow(:,:,1) is the displacement along
    columns, flow(:,:,2) along rows (variants/testfile_gt_sintel.m:36-38,
    MultiframeMotionCoupling.m:955-959).
    DECISION (range): a row whose sample position leaves the closed pixel
    domain [0,Ny-1]x[0,Nx-1] (or is NaN) is entirely zero; this is what the
    `sum(abs(warp),2)==0` marker of constructWarpFF.m:29 detects.  At the far
    edge the +1 tap index is clamped (its weight is exactly 0 there).
    Weights are the float32 products (1-a)(1-b), a(1-b), (1-a)b, ab.
    """
    Ny, Nx = int(dims[0]), int(dims[1])
    N = Ny * Nx
    valid, r0, r1, c0, c1, a, b = warp_taps(Ny, Nx, flowField)
    one = F32(1)
    w00 = ((one - a) * (one - b)).astype(np.float64)
    w10 = (a * (one - b)).astype(np.float64)
    w01 = ((one - a) * b).astype(np.float64)
    w11 = (a * b).astype(np.float64)
    row = (np.arange(Ny)[:, None] + Ny * np.arange(Nx)[None, :])
    vm = valid
    rows = np.concatenate([row[vm]] * 4)
    cols = np.concatenate([(r0 + Ny * c0)[vm], (r1 + Ny * c0)[vm], (r0 + Ny * c1)[vm], (r1 + Ny * c1)[vm]])
    vals = np.concatenate([w00[vm], w10[vm], w01[vm], w11[vm]])
    W = sp.coo_matrix((vals, (rows, cols)), shape=(N, N)).tocsr()
    W.eliminate_zeros()
    return W


def spmat_gradient2d(nx, ny, nc):
    """spmat_gradient2d(Nx, Ny, nc) [prost, not in checkout].

    Call sites: MultiframeMotionCoupling.m:434, :490-492, :626-627.
    DECISION: forward differences with a zero row at the last column / last
    image row; block order [Dx for all frames ; Dy for all frames] (the
    reference itself splits it that way at :626-627).  Dx differences along
    columns (stride Ny), Dy along rows (stride 1).
    """
    def d1(n):
        D = sp.diags([-np.ones(n), np.ones(n - 1)], [0, 1], shape=(n, n), format='lil')
        D[n - 1, :] = 0
        return D.tocsr()
    Dx = sp.kron(d1(nx), sp.identity(ny), format='csr')
    Dy = sp.kron(sp.identity(nx), d1(ny), format='csr')
    I = sp.identity(nc, format='csr')
    return sp.vstack([sp.kron(I, Dx, format='csr'), sp.kron(I, Dy, format='csr')], format='csr')


# --------------------------------------------------------------------------
# helpers/constructWarpFF.m, helpers/constructWarpFB.m
# --------------------------------------------------------------------------


def _pair_blocks(v, i, occ):
    """Shared body of the two constructors: constructWarpFF.m:22-38."""
    Ny, Nx = v.shape[0], v.shape[1]
    N = Ny * Nx
    warp = warpingOperator([Ny, Nx], v[:, :, i, :]).tolil()
    marker = np.asarray(abs(warp.tocsr()).sum(axis=1)).ravel() == 0          # :29
    if occ is not None:
        marker = marker | (np.asarray(occ[:, :, i]).ravel(order='F') != 0)   # :32-35
    warp = warp.tocsr()
    keep = sp.diags((~marker).astype(np.float64))
    warp = keep @ warp                                                       # :37
    idd = (~marker).astype(np.float64)                                       # :38
    return warp.tocoo(), idd, N


def _assemble(v, occ, pair_is_IF):
    Ny, Nx, n4 = v.shape[0], v.shape[1], v.shape[2]
    nc = n4 + 1
    N = Ny * Nx
    spX, spY, spA = [], [], []
    for i in range(nc - 1):
        warp, idd, _ = _pair_blocks(v, i, occ)
        if pair_is_IF(i):
            # constructWarpFF.m:40-48: -Id on frame i, warp on frame i+1
            spX.append(N * i + np.arange(N)); spY.append(N * i + np.arange(N)); spA.append(-idd)
            spX.append(N * i + warp.row); spY.append(N * (i + 1) + warp.col); spA.append(warp.data)
        else:
            # constructWarpFF.m:49-57: warp on frame i, -Id on frame i+1
            spX.append(N * i + warp.row); spY.append(N * i + warp.col); spA.append(warp.data)
            spX.append(N * i + np.arange(N)); spY.append(N * (i + 1) + np.arange(N)); spA.append(-idd)
    if not spX:
        return sp.csr_matrix((N * nc, N * nc))
    W = sp.coo_matrix((np.concatenate(spA), (np.concatenate(spX), np.concatenate(spY))),
                      shape=(N * nc, N * nc)).tocsr()                        # :61
    W.eliminate_zeros()
    return W


def constructWarpFF(v, type_, occ=None):
    """helpers/constructWarpFF.m:1-62.  v is Ny x Nx x (nc-1) x 2."""
    if type_ not in ('I-F', 'F-I'):
        raise ValueError(type_)
    return _assemble(np.asarray(v), occ, lambda i: type_ == 'I-F')


def constructWarpFB(v, occ=None):
    """helpers/constructWarpFB.m:1-57: odd pairs (1-based) as I-F, even as F-I."""
    return _assemble(np.asarray(v), occ, lambda i: (i + 1) % 2 == 1)


# --------------------------------------------------------------------------
# Problem assembly: MultiframeMotionCoupling.m:346-574 (init_u_prost)
# --------------------------------------------------------------------------


class Problem(object):
    """Saddle-point problem  min_x max_y <Kx,y> + G(x) - F*(y)  as assembled by
    init_u_prost.  Holds the stacked sparse K, the prox parameters and the
    diagonal step sizes."""
    pass


def build_problem(f_lr, v, factor=4, alpha=0.01, kappa=0.25, flowDirection='backward',
                  k=None, interpMethod='average', interpKernel=None, interpAA=False,
                  occ=None, u_init=None, w_init=None, h=None, u0_frame=None, w0_frame=None,
                  flexbox_backward_terms=False):
    """init_u_prost restated.  f_lr: ny x nx x nc;  v: Ny x Nx x (nc-1) x 2."""
    f_lr = np.asarray(f_lr, dtype=np.float64)
    ny, nx, nc = f_lr.shape
    Ny, Nx = factor * ny, factor * nx
    N, M, n = Ny * Nx, Ny * Nx * nc, ny * nx
    P = Problem()
    P.dims = (Ny, Nx, nc, ny, nx)
    P.alpha, P.kappa, P.factor = float(alpha), float(kappa), factor

    dsOp = samplingOperator([Ny, Nx], [ny, nx], interpMethod, interpKernel, interpAA)   # :368-370
    if k is not None and np.size(k) > 0:
        dsOp = dsOp @ RepConvMtx(k, [Ny, Nx])                                            # :373-375
    P.dsOp = dsOp.tocsr()

    v = np.asarray(v, dtype=np.float64).reshape(Ny, Nx, max(nc - 1, 0), 2)
    if flowDirection == 'forward':                                                       # :392-398
        W = constructWarpFF(v, 'F-I', occ)
    elif flowDirection == 'backward':
        W = constructWarpFF(v, 'I-F', occ)
    elif flowDirection == 'forward-backward':
        W = constructWarpFB(v, occ)
    else:
        raise ValueError('Invalid flow direction.')
    P.W = W

    u_up = np.zeros((Ny, Nx, nc))                                                        # :415-419
    for i in range(nc):
        u_up[:, :, i] = imresize(f_lr[:, :, i], factor)
    P.u_up = u_up

    G = spmat_gradient2d(Nx, Ny, nc)                                                     # :434
    P.G = G
    uu = u_up.ravel(order='F')
    P.warpPix = np.abs(W @ uu).sum() / uu.size                                           # :432
    P.gradPix = np.abs(G @ uu).sum() / uu.size                                           # :435
    if h is not None and h > 0:
        P.h = float(h)
    elif nc == 1:
        P.h = 1.0           # no temporal rows at all; the reference's 0/x would only scale an empty W
    else:
        P.h = P.warpPix / P.gradPix                                                      # :437
    hh = P.h

    A = sp.kron(sp.identity(nc), P.dsOp, format='csr')                                   # :487 id_kron_sparse
    fvec = f_lr.ravel(order='F')
    Z = lambda r, c: sp.csr_matrix((r, c))

    fixed = u0_frame is not None
    if not np.isnan(kappa) and kappa != 1:
        # model (i) / (iv): :480-498, :535-571
        K1 = sp.vstack([W / hh, kappa * G], format='csr')                                # :490
        K2 = sp.vstack([kappa * W / hh, G], format='csr')                                # :492
        K1w = -K1                                                                        # :490-491
        if flexbox_backward_terms:
            # init_u_flexBox, 'backward': temporal row {Id,-Wi,Id,-Wi} = (Id - W)(u + w) on
            # [u_i,u_i+1,w_i,w_i+1], spatial rows kappa*D on u and -kappa*D on w (:864-866).  The
            # sign of a whole row does not matter under the norm, the relative sign of w does.
            K1w = sp.vstack([W / hh, -kappa * G], format='csr')
        rows = [sp.hstack([A, Z(n * nc, M)]),
                sp.hstack([K1, K1w]),
                sp.hstack([Z(3 * M, M), K2])]
        P.primal = ['u', 'w']
        P.dual_blocks = [('p', n * nc, 'box_lin', (1.0, fvec)),
                         ('g1', 3 * M, 'ball', (alpha, 3, M)),                           # :497
                         ('g2', 3 * M, 'ball', (alpha, 3, M))]                           # :498
        if fixed:
            S1 = sp.hstack([sp.identity(N), Z(N, M - N)], format='csr')                  # :542,:560
            rows.append(sp.hstack([S1, Z(N, M)]))
            rows.append(sp.hstack([Z(N, M), S1]))                                        # :561
            P.dual_blocks.append(('q1', N, 'lin', (np.asarray(u0_frame, float).ravel(order='F'),)))   # :563
            P.dual_blocks.append(('q2', N, 'lin', (np.asarray(w0_frame, float).ravel(order='F'),)))   # :564
        K = sp.vstack(rows, format='csr')
    elif kappa == 1:
        # model (ii): :500-513, ball radius 2*alpha
        K1 = sp.vstack([W / hh, G], format='csr')                                        # :509
        K = sp.vstack([A, K1], format='csr')
        P.primal = ['u']
        P.dual_blocks = [('p', n * nc, 'box_lin', (1.0, fvec)),
                         ('g1', 3 * M, 'ball', (2.0 * alpha, 3, M))]                     # :513
    else:
        # model (iii): :515-533, additive TV + warp
        K = sp.vstack([A, W / hh, G], format='csr')                                      # :524-528
        P.primal = ['u']
        P.dual_blocks = [('p', n * nc, 'box_lin', (1.0, fvec)),
                         ('g1', M, 'box', (alpha,)),                                     # :532
                         ('g2', 2 * M, 'ball', (alpha, 2, M))]                           # :533
    P.K = K.tocsr()
    P.KT = P.K.T.tocsr()
    P.M, P.N, P.n = M, N, n

    # DECISION (step sizes): diagonal preconditioning T_j = 1/sum_i|K_ij|,
    # S_i = 1/sum_j|K_ij| with tau0 = sigma0 = 1 (:349-351), theta = 1; zero
    # sums are replaced by 1.  Rows that share one isotropic norm (a 'ball'
    # group) use ONE step, the smallest of the group (= 1/largest row sum), so
    # that the prox stays the Euclidean projection and the step-size condition
    # of diagonal preconditioning still holds.
    absK = abs(P.K)
    col = np.asarray(absK.sum(axis=0)).ravel()
    row = np.asarray(absK.sum(axis=1)).ravel()
    off = 0
    for name, size, kind, par in P.dual_blocks:
        if kind == 'ball':
            _, gdim, glen = par
            blk = row[off:off + size].reshape(gdim, glen)
            row[off:off + size] = np.tile(blk.max(axis=0), gdim)
        off += size
    P.T = np.where(col > 0, 1.0 / np.where(col > 0, col, 1.0), 1.0)
    P.S = np.where(row > 0, 1.0 / np.where(row > 0, row, 1.0), 1.0)

    # DECISION (start point): caller supplies u, w; default u = w = u_up
    # (the flexBox_vector choice, MultiframeMotionCoupling.m:686-687); duals 0.
    u0 = uu if u_init is None else np.asarray(u_init, float).ravel(order='F')
    if len(P.primal) == 2:
        w0 = uu if w_init is None else np.asarray(w_init, float).ravel(order='F')
        P.x0 = np.concatenate([u0, w0])
    else:
        P.x0 = u0.copy()
    P.y0 = np.zeros(P.K.shape[0])
    return P


def prox_G(P, x):
    """G(u) = indicator of [0,1] (:495); w is free."""
    x = x.copy()
    M = P.M
    np.clip(x[:M], 0.0, 1.0, out=x[:M])
    return x


def prox_Fstar(P, y, S):
    """Dual prox maps, MultiframeMotionCoupling.m:496-498, :513, :532-533, :563-564.

    sum_1d('ind_box01',0.5,-0.5,1,f,0)   -> ind{|p|<=1} + <f,p>
    sum_norm2(d,false,'ind_leq0',1/r,1,1,0,0) -> ind{||g_j||_2 <= r}, groups are
        (j, j+len, j+2len) i.e. not interleaved
    sum_1d('ind_box01',0.5/a,-0.5,1,0,0) -> ind{|g|<=a}
    sum_1d('zero',1,0,1,u0,0)            -> <u0,q>
    """
    out = np.empty_like(y)
    off = 0
    for name, size, kind, par in P.dual_blocks:
        yy = y[off:off + size]
        ss = S[off:off + size]
        if kind == 'box_lin':
            bound, lin = par
            out[off:off + size] = np.clip(yy - ss * lin, -bound, bound)
        elif kind == 'box':
            out[off:off + size] = np.clip(yy, -par[0], par[0])
        elif kind == 'lin':
            out[off:off + size] = yy - ss * par[0]
        elif kind == 'ball':
            r, gdim, glen = par
            g = yy.reshape(gdim, glen)
            nrm = np.sqrt((g * g).sum(axis=0))
            scale = np.where(nrm > r, r / np.where(nrm > 0, nrm, 1.0), 1.0)
            out[off:off + size] = (g * scale[None, :]).ravel()
        off += size
    return out


STEP_ADAPT_EVERY, STEP_DELTA, STEP_ETA = 50, 1.5, 0.8


def step_balance_rule(P2, D2, balance, alpha):
    """Residual balancing of the step sizes (the rule of adaptive PDHG; prost's 'stepsize','boyd' setting,
    MultiframeMotionCoupling.m:349-351, is of this family, its constants live in the absent submodule):
    P2 = ||(x+ - x)/tau||^2, D2 = ||(y+ - y)/sigma||^2.  Returns the new (balance, alpha)."""
    Pn, Dn = np.sqrt(max(P2, 0.0)), np.sqrt(max(D2, 0.0))
    if not (Pn > 0 and Dn > 0 and np.isfinite(Pn) and np.isfinite(Dn)):
        return balance, alpha
    if Pn > STEP_DELTA * Dn:
        return min(max(balance / (1.0 - alpha), 1e-4), 1e4), alpha * STEP_ETA
    if Pn < Dn / STEP_DELTA:
        return min(max(balance * (1.0 - alpha), 1e-4), 1e4), alpha * STEP_ETA
    return balance, alpha


def pdhg(P, iters, x=None, y=None, dtype=np.float64, callback=None, balance=1.0, adapt=False, trace=None):
    """DECISION (iteration): preconditioned PDHG, primal step first:
        x+ = prox_G(x - T .* K'y);  xb = 2 x+ - x;  y+ = prox_F*(y + S .* K xb)
    run for a fixed number of iterations.  Returns (x, y, last_change) where
    last_change = (||x+ - x||, ||x+||, ||y+ - y||, ||y+||) of the final step.
    balance: one global factor s, T <- s T and S <- S / s (the product of the steps is unchanged).
    adapt: every STEP_ADAPT_EVERY iterations s is moved by step_balance_rule; trace (a list) receives
    (iteration, s, P2, D2) at those points.
    """
    K = P.K.astype(dtype)
    KT = P.KT.astype(dtype)
    T0 = P.T.astype(dtype)
    S0 = P.S.astype(dtype)
    s_bal, alpha = float(balance), 0.5
    T = (T0 * dtype(s_bal)).astype(dtype)
    S = (S0 / dtype(s_bal)).astype(dtype)
    x = (P.x0 if x is None else x).astype(dtype)
    y = (P.y0 if y is None else y).astype(dtype)
    Pd = P
    if dtype != np.float64:
        import copy
        Pd = copy.copy(P)
        Pd.dual_blocks = [(nm, sz, kd, tuple(np.asarray(q, dtype=dtype) if isinstance(q, np.ndarray) else q
                                                for q in par)) for nm, sz, kd, par in P.dual_blocks]
    chg = (0.0, 0.0, 0.0, 0.0)
    for it in range(iters):
        xn = prox_G(Pd, x - T * (KT @ y))
        xb = (2 * xn - x).astype(dtype)
        yn = prox_Fstar(Pd, (y + S * (K @ xb)).astype(dtype), S).astype(dtype)
        if it == iters - 1 or callback is not None:
            chg = (float(np.linalg.norm(xn - x)), float(np.linalg.norm(xn)),
                   float(np.linalg.norm(yn - y)), float(np.linalg.norm(yn)))
        if adapt and (it + 1) % STEP_ADAPT_EVERY == 0:
            P2 = float((((xn - x) / T).astype(np.float64) ** 2).sum())
            D2 = float((((yn - y) / S).astype(np.float64) ** 2).sum())
            s_bal, alpha = step_balance_rule(P2, D2, s_bal, alpha)
            if trace is not None:
                trace.append((it + 1, s_bal, P2, D2))
            T = (T0 * dtype(s_bal)).astype(dtype)
            S = (S0 / dtype(s_bal)).astype(dtype)
        x, y = xn, yn
        if callback is not None:
            callback(it, x, y, chg)
    return x, y, chg


def primal_energy(P, x):
    """||Au-f||_1 + alpha sum_j ||(K1(u-w))_j|| + ... : the primal objective
    F(Kx) evaluated from the dual block structure (support functions)."""
    Kx = P.K @ x
    e = 0.0
    off = 0
    for name, size, kind, par in P.dual_blocks:
        z = Kx[off:off + size]
        if kind == 'box_lin':
            e += par[0] * np.abs(z - par[1]).sum()
        elif kind == 'box':
            e += par[0] * np.abs(z).sum()
        elif kind == 'ball':
            r, gdim, glen = par
            g = z.reshape(gdim, glen)
            e += r * np.sqrt((g * g).sum(axis=0)).sum()
        elif kind == 'lin':
            pass    # hard constraint z == par[0]; not part of the finite energy
        off += size
    return float(e)


# --------------------------------------------------------------------------
# Optimality certificates: dual objective, primal-dual gap, residuals.
# These depend on the MODEL only (K, f, the radii), not on any step-size DECISION.
# --------------------------------------------------------------------------


def dual_terms(P, y):
    """Pieces of the dual objective  D(y) = -F*(y) - G*(-K'y)  for a dual point inside its
    constraint sets (what every prox_Fstar returns):
      lin        <f,p> (+ <u0,q1> + <w0,q2>): the linear parts of F*  (:496, :563-564)
      box_u      G*(-K'y) on u, G = indicator of [0,1] (:495):  sum_j max(0, -(K'y)_u,j)
      infeas_w   sum_j |(K'y)_w,j|: w is free (G = 0), so D is finite only where this is 0
    """
    KTy = P.KT @ y
    lin = 0.0
    off = 0
    for name, size, kind, par in P.dual_blocks:
        yy = y[off:off + size]
        if kind == 'box_lin':
            lin += float(np.dot(par[1], yy))
        elif kind == 'lin':
            lin += float(np.dot(par[0], yy))
        off += size
    box_u = float(np.maximum(-KTy[:P.M], 0.0).sum())
    infeas_w = float(np.abs(KTy[P.M:]).sum()) if len(P.primal) == 2 else 0.0
    return {'lin': lin, 'box_u': box_u, 'infeas_w': infeas_w}


def dual_energy(P, y, w_bound=0.0):
    """D(y) with w confined to [-w_bound, w_bound] (G*_w(z) = w_bound * |z|_1).  w_bound = 0 is
    the relaxed value that ignores the constraint K_w'y = 0; any w_bound >= max|w*| gives a valid
    lower bound of the optimal primal energy."""
    t = dual_terms(P, y)
    return -t['lin'] - t['box_u'] - w_bound * t['infeas_w']


def primal_dual_gap(P, x, y, w_bound=0.0):
    """primal_energy(x) - dual_energy(y, w_bound) for x inside the box and y inside its sets."""
    return primal_energy(P, x) - dual_energy(P, y, w_bound)


def optimality_residuals(P, x, y):
    """Distance of (x, y) from the optimality conditions, independent of step sizes:
      primal: min-norm element of  dG(x) + K'y   (u: projected onto the normal cone of the box)
      dual  : min-norm element of  dF*(y) - Kx   (per block, projected onto the normal cone of the
              constraint set of y; the 'lin' blocks have dF* = {x0})
    Returns (r_primal, r_dual, |K'y|, |Kx|) as 2-norms; a solver in the spirit of prost's
    abs+rel test (MultiframeMotionCoupling.m:357-363) stops when
      r_primal <= tol*sqrt(len x) + tol*|K'y|  and  r_dual <= tol*sqrt(len y) + tol*|Kx|."""
    KTy = P.KT @ y
    Kx = P.K @ x
    M = P.M
    gu = KTy[:M].copy()            # need -gu in N_[0,1](u)
    u = x[:M]
    at0, at1 = u <= 0.0, u >= 1.0
    gu[at0] = np.minimum(gu[at0], 0.0)      # at u = 0 the cone is (-inf, 0]: -gu <= 0 is fine, gu < 0 violates
    gu[at1] = np.maximum(gu[at1], 0.0)
    rp2 = float((gu * gu).sum() + (KTy[M:] ** 2).sum())
    rd2 = 0.0
    off = 0
    for name, size, kind, par in P.dual_blocks:
        yy = y[off:off + size]
        z = Kx[off:off + size]
        if kind == 'box_lin' or kind == 'box':
            bound = par[0]
            d = z - (par[1] if kind == 'box_lin' else 0.0)          # need d in N_[-b,b](y)
            r = d.copy()
            up, lo = yy >= bound, yy <= -bound
            r[up] = np.minimum(d[up], 0.0)
            r[lo] = np.maximum(d[lo], 0.0)
            rd2 += float((r * r).sum())
        elif kind == 'lin':
            d = z - par[0]
            rd2 += float((d * d).sum())
        elif kind == 'ball':
            rad, gdim, glen = par
            g = yy.reshape(gdim, glen)
            d = z.reshape(gdim, glen)
            nrm = np.sqrt((g * g).sum(axis=0))
            on = nrm * nrm >= rad * rad * (1.0 - 1e-5)        # same threshold as the float32 implementations
            # on the sphere the cone is {lambda g, lambda >= 0}: remove the non-negative radial part
            lam = np.where(on, np.maximum((d * g).sum(axis=0), 0.0) / np.where(nrm > 0, nrm * nrm, 1.0), 0.0)
            r = d - lam[None, :] * g
            rd2 += float((r * r).sum())
        off += size
    return np.sqrt(rp2), np.sqrt(rd2), float(np.linalg.norm(KTy)), float(np.linalg.norm(Kx))


def operator_norm(P, iters=200, seed=0):
    """|K|_2 by power iteration on K'K."""
    rng = np.random.default_rng(seed)
    x = rng.normal(size=P.K.shape[1])
    x /= np.linalg.norm(x)
    s = 0.0
    for _ in range(iters):
        z = P.KT @ (P.K @ x)
        s = np.linalg.norm(z)
        x = z / s
    return float(np.sqrt(s))


def pdhg_scalar(P, iters, x=None, y=None, ratio=1.0, norm=None, callback=None, every=0, restart_every=0, score=None):
    """Plain Chambolle-Pock PDHG with ONE scalar step pair, tau*sigma*|K|^2 = 0.98^2 and
    tau/sigma = ratio, dual step first:
        y+ = prox_{sigma F*}(y + sigma K xb);  x+ = prox_G(x - tau K'y+);  xb = 2x+ - x.
    It shares neither the diagonal preconditioning nor the update order of pdhg() and of the
    HIP sweeps; it exists to reach the optimum of the same model by another route.
    restart_every > 0: every that many iterations the run continues from the average of the
    iterates since the last restart when score(x, y) (smaller is better, e.g. the gap) says the
    average is the better point (restarted averaged PDHG)."""
    L = operator_norm(P) if norm is None else norm
    tau = 0.98 / L * np.sqrt(ratio)
    sigma = 0.98 / L / np.sqrt(ratio)
    x = (P.x0 if x is None else x).astype(np.float64)
    y = (P.y0 if y is None else y).astype(np.float64)
    xb = x.copy()
    S = np.full(P.K.shape[0], sigma)
    xs, ys, cnt = np.zeros_like(x), np.zeros_like(y), 0
    for it in range(iters):
        y = prox_Fstar(P, y + sigma * (P.K @ xb), S)
        xn = prox_G(P, x - tau * (P.KT @ y))
        xb = 2.0 * xn - x
        x = xn
        if restart_every:
            xs += x
            ys += y
            cnt += 1
            if (it + 1) % restart_every == 0:
                xa, ya = xs / cnt, ys / cnt
                if score(xa, ya) < score(x, y):
                    x, y, xb = xa, ya, xa.copy()
                xs[:] = 0.0
                ys[:] = 0.0
                cnt = 0
        if callback is not None and every and (it + 1) % every == 0:
            if callback(it + 1, x, y):
                break
    return x, y


def split_primal(P, x):
    Ny, Nx, nc = P.dims[:3]
    u = x[:P.M].reshape((Ny, Nx, nc), order='F')
    w = x[P.M:2 * P.M].reshape((Ny, Nx, nc), order='F') if len(P.primal) == 2 else None
    return u, w


def solve(f_lr, v, iters, **kw):
    """Convenience: assemble and run `iters` PDHG iterations; returns u, w, P."""
    dtype = kw.pop('dtype', np.float64)
    P = build_problem(f_lr, v, **kw)
    x, y, chg = pdhg(P, iters, dtype=dtype)
    u, w = split_primal(P, x)
    P.last_y = y
    P.last_change = chg
    return u, w, P


def psnr_central(result, truth, frame=None):
    """testfile.m:71-75: PSNR of the central frame, cropped 20:end-20 on both
    axes (MATLAB 1-based inclusive => python [19:-20]), peak 1."""
    nc = result.shape[-1]
    t = int(np.ceil(nc / 2.0)) - 1 if frame is None else frame
    a = result[19:-20, 19:-20, ..., t]
    b = truth[19:-20, 19:-20, ..., t]
    mse = np.mean((a - b) ** 2)
    return 10.0 * np.log10(1.0 / mse)
