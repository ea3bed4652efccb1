"""CPU restatement of the classical single-frame super-resolution solve -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the
product path (superresolution_amd/) never does.  PARITY UNPINNED: like mmc_oracle.py this follows
the text of the reference's .m files; the reference ships no golden vectors and cannot run here.

What it restates: the problem that classical methods/singleframeMotionSR_mitzel_prost.m:75-163
(and ..._mitzel_prost_fast.m, ..._unger_prost.m, which differ in how the flows are estimated and in
the Huber parameter of the data term) hand to prost.solve:

    min_u  sum_i  H_delta( D B W_i u - f_i )  +  alpha * sum_j H_delta( |grad u|_j )

one high-resolution frame u (the central one), one data term per low-resolution frame i with the
warp W_i from the central frame to frame i (flow given; identity for the central frame, :64-70),
D the 'custom' box sampler {abs(x)<=2)/4, 7} (:75-77, the same rows as interpMethod 'average'),
B the true convolution with replicate border (RepConvMtx, :78-79), H_delta the Huber function
(delta = 0: the absolute value; ..._unger_prost.m:148-161, :163 passes huber_eps to the data terms AND to the
regulariser, which makes it Huber-smoothed TV; the Mitzel files pass 0 to both), TV isotropic.

Saddle-point form as assembled at :97-163:
    primal  u in R^N                                    (:97, no constraint, start value 0)
    duals   p_i in R^n:  ind{|p|<=1} + <f_i,p> + delta/2 |p|^2   (sum_1d('ind_box01',0.5,-0.5,1,f_i,delta), :148-161)
            q in R^2N:   ind{||q_j||_2 <= alpha} + (delta/alpha)/2 ||q_j||^2   (sum_norm2(2,false,'ind_leq0',1/alpha,1,1,0,delta/alpha), :163)
    K = [D B W_1; ...; D B W_nf; grad]                            (:128-145)

DECISIONS shared with mmc_oracle.py (the prost backend is absent): warpingOperator = bilinear,
gradient = forward differences, PDHG with the primal step first, theta = 1, diagonal
preconditioning with tau0 = sigma0 = 1 (:84-85; 'boyd' balancing not reproduced), one step per
isotropic group.
"""
import numpy as np
import scipy.sparse as sp

import mmc_oracle as orc


class SFProblem(object):
    pass


def central_slice(num_frames):
    """0-based index of ceil(numFrames/2) (:21)."""
    return (num_frames + 1) // 2 - 1


def build_problem(f_lr, v, factor=4, alpha=0.1, huber_eps=0.0, k=None, sampling='average', occ=None):
    """f_lr: ny x nx x nf; v: Ny x Nx x nf x 2 (flow from the central frame to frame i; the central
    slice is ignored and taken as zero, :35-37); occ: optional Ny x Nx x nf mask of rows to drop."""
    f_lr = np.asarray(f_lr, dtype=np.float64)
    ny, nx, nf = f_lr.shape
    Ny, Nx = ny * factor, nx * factor
    N, n = Ny * Nx, ny * nx
    v = np.asarray(v, dtype=np.float64)
    assert v.shape == (Ny, Nx, nf, 2), v.shape
    cs = central_slice(nf)
    D = orc.samplingOperator((Ny, Nx), (ny, nx), sampling)                      # :75-77
    A = D if k is None else (D @ orc.RepConvMtx(k, (Ny, Nx))).tocsr()           # :78-79, :128
    rows = []
    for i in range(nf):
        vi = np.zeros((Ny, Nx, 2)) if i == cs else v[:, :, i, :]
        W = orc.warpingOperator((Ny, Nx), vi)                                    # :64-70
        if occ is not None and i != cs:
            keep = sp.diags((~np.asarray(occ[:, :, i], dtype=bool)).ravel(order='F').astype(np.float64))
            W = (keep @ W).tocsr()
        rows.append((A @ W).tocsr())                                             # :128-141
    G = orc.spmat_gradient2d(Nx, Ny, 1)                                          # :145
    P = SFProblem()
    P.K = sp.vstack(rows + [G], format='csr')
    P.KT = P.K.T.tocsr()
    P.N, P.n, P.nf, P.Ny, P.Nx, P.ny, P.nx = N, n, nf, Ny, Nx, ny, nx
    P.alpha, P.delta = float(alpha), float(huber_eps)
    P.f = np.concatenate([f_lr[:, :, i].ravel(order='F') for i in range(nf)])
    absK = abs(P.K)
    col = np.asarray(absK.sum(axis=0)).ravel()
    row = np.asarray(absK.sum(axis=1)).ravel()
    g = row[n * nf:].reshape(2, N)
    row[n * nf:] = np.tile(g.max(axis=0), 2)              # one step per isotropic group
    P.T = np.where(col > 0, 1.0 / np.where(col > 0, col, 1.0), 1.0)
    P.S = np.where(row > 0, 1.0 / np.where(row > 0, row, 1.0), 1.0)
    return P


def prox_Fstar(P, y, S):
    """p <- clamp((p~ - s f) / (1 + s delta), -1, 1);  q_j <- projection of q~_j / (1 + s delta/alpha) onto
    the alpha-ball: F*(q) = ind{||q_j|| <= alpha} + (delta/alpha)/2 ||q_j||^2
    (sum_norm2(2,false,'ind_leq0',1/alpha,1,1,0,huber_eps/alpha), singleframeMotionSR_unger_prost.m:163)."""
    m = P.n * P.nf
    out = np.empty_like(y)
    s = S[:m]
    out[:m] = np.clip((y[:m] - s * P.f.astype(y.dtype)) / (1.0 + s * P.delta), -1.0, 1.0)
    g = (y[m:] / (1.0 + S[m:] * (P.delta / P.alpha))).reshape(2, P.N)
    nrm = np.sqrt((g * g).sum(axis=0))
    scale = np.where(nrm > P.alpha, P.alpha / np.where(nrm > 0, nrm, 1.0), 1.0)
    out[m:] = (g * scale[None, :]).ravel()
    return out


def pdhg(P, iters, x=None, y=None, dtype=np.float64):
    """x+ = x - T K'y;  xb = 2x+ - x;  y+ = prox_F*(y + S K xb).  Returns (x, y, last change)."""
    K, KT = P.K.astype(dtype), P.KT.astype(dtype)
    T, S = P.T.astype(dtype), P.S.astype(dtype)
    x = (np.zeros(P.N) if x is None else x).astype(dtype)
    y = (np.zeros(P.K.shape[0]) if y is None else y).astype(dtype)
    chg = (0.0, 0.0, 0.0, 0.0)
    for it in range(iters):
        xn = x - T * (KT @ y)
        xb = 2 * xn - x
        yn = prox_Fstar(P, y + S * (K @ xb), S).astype(dtype)
        if it == iters - 1:
            chg = (float(np.linalg.norm(xn - x)), float(np.linalg.norm(xn)),
                   float(np.linalg.norm(yn - y)), float(np.linalg.norm(yn)))
        x, y = xn, yn
    return x, y, chg


def primal_energy(P, u):
    """sum_i H_delta(A W_i u - f_i) + alpha sum_j H_delta(|grad u|_j): Huber data terms and Huber-smoothed TV
    (the conjugates of the dual functions of singleframeMotionSR_unger_prost.m:148-163); delta = 0: L1 + TV."""
    z = P.K @ u
    m = P.n * P.nf
    r = np.abs(z[:m] - P.f)
    if P.delta > 0:
        data = np.where(r <= P.delta, r * r / (2 * P.delta), r - P.delta / 2).sum()
    else:
        data = r.sum()
    g = z[m:].reshape(2, P.N)
    gn = np.sqrt((g * g).sum(axis=0))
    if P.delta > 0:
        reg = np.where(gn <= P.delta, gn * gn / (2 * P.delta), gn - P.delta / 2).sum()
    else:
        reg = gn.sum()
    return float(data + P.alpha * reg)


def image(P, u):
    return np.asarray(u).reshape((P.Ny, P.Nx), order='F')
