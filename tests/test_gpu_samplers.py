"""GPU parity for the samplers of helpers/samplingOperator.m other than 'average'/'stride':
the library receives the rows of the two 1-D factors, applies the interior taps as a composite
kernel and the deviating border rows as an explicit sparse correction.  Checked against the
oracle's explicit sparse A = S*B (forward, adjoint, step sizes, full solve)."""
import numpy as np
import pytest

import mmc_oracle as orc
from helpers import make_case, rel_l2

pytestmark = pytest.mark.gpu

KBLUR = orc.fspecial_gaussian(7, np.sqrt(0.6))
TRI = (lambda x: np.maximum(1 - np.abs(x), 0), 2.0)


def _make(f, v, P, method, kernel, aa, k, factor, kappa=0.25):
    from superresolution_amd.sampling import dim1_taps
    from superresolution_amd.solver import FrameBlockSolver
    ny, nx, nc = f.shape
    s = FrameBlockSolver(ny, nx, factor, nc, alpha=P.alpha, kappa=kappa, h=P.h, sampling="taps")
    fy, wy = dim1_taps(factor * ny, ny, method, kernel, aa)
    fx, wx = dim1_taps(factor * nx, nx, method, kernel, aa)
    s.set_sampling_taps(fy, wy, fx, wx)
    s.set_lowres(f); s.set_blur(k); s.set_flow(v); s.set_state(P.u_up, P.u_up); s.init()
    return s


def _split_y(P, y):
    out, off = {}, 0
    for name, size, kind, par in P.dual_blocks:
        out[name] = y[off:off + size]
        off += size
    return out


CASES = [("bicubic", None, True, KBLUR, 4, 20, 12), ("bicubic", None, False, KBLUR, 4, 9, 7),
         ("bilinear", None, True, None, 4, 10, 8), ("bilinear", None, False, KBLUR, 4, 9, 9),
         ("nearest", None, False, None, 4, 8, 8), ("nearest", None, True, KBLUR, 2, 10, 14),
         ("lanczos2", None, False, KBLUR, 4, 12, 9), ("lanczos3", None, True, None, 2, 18, 12),
         ("custom", TRI, True, KBLUR, 4, 8, 11), ("average", None, False, KBLUR, 2, 12, 10),
         ("average", None, False, None, 3, 8, 8),
         # interp1-based samplers (samplingOperator.m:99-106); at zoom 2 the first and last rows use the end formulas
         ("pchip", None, False, KBLUR, 2, 10, 9), ("pchip", None, False, None, 4, 8, 9),
         ("bicubic-0", None, False, KBLUR, 4, 9, 8), ("bicubic-0", None, False, None, 2, 10, 12)]


@pytest.mark.parametrize("method,kernel,aa,k,factor,ny,nx", CASES)
def test_operator_parity(method, kernel, aa, k, factor, ny, nx):
    f, v, _ = make_case(71, ny, nx, 2, factor=factor)
    P = orc.build_problem(f, v, factor=factor, k=k, interpMethod=method, interpKernel=kernel, interpAA=aa)
    s = _make(f, v, P, method, kernel, aa, k, factor)
    rng = np.random.default_rng(1)
    Ny, Nx, nc = P.dims[:3]
    x = rng.normal(size=P.K.shape[1])
    u = x[:P.M].reshape((Ny, Nx, nc), order="F")
    w = x[P.M:].reshape((Ny, Nx, nc), order="F")
    p, g1, g2 = s.debug_apply_K(u, w)
    ref = _split_y(P, P.K @ x)
    assert rel_l2(p, ref["p"]) < 3e-6
    y = rng.normal(size=P.K.shape[0])
    y[np.asarray(abs(P.K).sum(axis=1)).ravel() == 0] = 0
    yb = _split_y(P, y)
    ku, kw = s.debug_apply_KT(yb["p"], yb["g1"], yb["g2"])
    refx = P.KT @ y
    assert rel_l2(ku, refx[:P.M]) < 3e-6
    tu, tw, sp, s1, s2 = s.debug_steps()
    assert rel_l2(tu, P.T[:P.M]) < 2e-6
    assert rel_l2(sp, _split_y(P, P.S)["p"]) < 2e-6
    s.close()


@pytest.mark.parametrize("method,aa,ny,nx", [("bicubic", True, 40, 24), ("bilinear", False, 20, 12)])
def test_solve_parity(method, aa, ny, nx):
    """Full solve; 160x96 high-res pixels for the first case, so interior tiles, border tiles and
    the sparse border rows all take part."""
    f, v, _ = make_case(72, ny, nx, 3, flow="smooth", flow_scale=1.5)
    P = orc.build_problem(f, v, k=KBLUR, interpMethod=method, interpAA=aa)
    x, y, chg = orc.pdhg(P, 60)
    u_ref, w_ref = orc.split_primal(P, x)
    s = _make(f, v, P, method, None, aa, KBLUR, 4)
    s.h = P.h
    st = s.solve(60, 0.0, 0)
    u, w = s.get_state()
    e = s.energy()
    s.close()
    assert rel_l2(u, u_ref) < 1e-4
    e_ref = orc.primal_energy(P, x)
    assert abs(e - e_ref) <= 1e-3 * abs(e_ref)


def _disc(n):
    """A blur that is not an outer product: a disc of diameter n."""
    y, x = np.mgrid[:n, :n] - (n - 1) / 2.0
    k = (x * x + y * y <= (n / 2.0) ** 2).astype(np.float64)
    return k / k.sum()


@pytest.mark.parametrize("name,k,method", [("gaussian13", orc.fspecial_gaussian(13, 2.0), "average"),
                                           ("disc11", _disc(11), "average"), ("disc11-bicubic", _disc(11), "bicubic"),
                                           ("gaussian9x15", np.outer(orc.fspecial_gaussian(9, 1.5)[:, 4], orc.fspecial_gaussian(15, 2.5)[7, :]), "average")])
def test_large_blur_kernels(name, k, method):
    """Composite kernels beyond 3 x 3 low-res taps at zoom 4: a separable one is applied as two 1-D passes
    (column sums of the low-res window, then the rows; row sums of the xbar window, then the columns), a
    non-separable one from the full table.  40 x 28 low-res pixels: interior tiles, border tiles, and the
    virtual rows and columns of the replicate border all take part."""
    from superresolution_amd.solver import FrameBlockSolver
    k = k / k.sum()
    ny, nx, nc = 40, 28, 3
    f, v, _ = make_case(74, ny, nx, nc, flow="smooth", flow_scale=1.5)
    aa = method == "bicubic"
    P = orc.build_problem(f, v, k=k, interpMethod=method, interpAA=aa)
    if method == "average":
        s = FrameBlockSolver(ny, nx, 4, nc, alpha=P.alpha, kappa=0.25, h=P.h, sampling="average")
        s.set_lowres(f); s.set_blur(k); s.set_flow(v); s.set_state(P.u_up, P.u_up); s.init()
    else:
        s = _make(f, v, P, method, None, aa, k, 4)
    assert s.separable == name.startswith("gaussian")
    rng = np.random.default_rng(5)
    Ny, Nx = P.dims[:2]
    x = rng.normal(size=P.K.shape[1])
    p, g1, g2 = s.debug_apply_K(x[:P.M].reshape((Ny, Nx, nc), order="F"), x[P.M:].reshape((Ny, Nx, nc), order="F"))
    assert rel_l2(p, _split_y(P, P.K @ x)["p"]) < 3e-6
    y = rng.normal(size=P.K.shape[0])
    y[np.asarray(abs(P.K).sum(axis=1)).ravel() == 0] = 0
    yb = _split_y(P, y)
    ku, kw = s.debug_apply_KT(yb["p"], yb["g1"], yb["g2"])
    assert rel_l2(ku, (P.KT @ y)[:P.M]) < 3e-6
    assert rel_l2(s.debug_steps()[0], P.T[:P.M]) < 2e-6
    s.solve(30, 0.0, 0)
    u, w = s.get_state()
    s.close()
    xr, _, _ = orc.pdhg(P, 30)
    assert rel_l2(u, orc.split_primal(P, xr)[0]) < 2e-5


def test_mirror_class_with_bicubic_sampler():
    """The mirror of the MATLAB class with interpMethod = 'bicubic', interpAA = true."""
    from superresolution_amd import MultiframeMotionCoupling
    f, v, _ = make_case(73, 12, 10, 3, flow_scale=1.0)
    M = MultiframeMotionCoupling(f, 4, flowField=v, verbose=0)
    M.k = KBLUR
    M.interpMethod = "bicubic"
    M.interpAA = True
    M.comp_mode = "fast"
    M.init()
    M._opts["max_iters"] = 80
    M.run()
    P = orc.build_problem(f, v, k=KBLUR, interpMethod="bicubic", interpAA=True)
    x, y, _ = orc.pdhg(P, 80)
    u_ref, _ = orc.split_primal(P, x)
    assert rel_l2(M.result1, u_ref) < 1e-4
    M.close()


def test_taps_are_validated():
    from superresolution_amd._capi import MMCError
    from superresolution_amd.solver import FrameBlockSolver
    s = FrameBlockSolver(8, 8, 4, 2, sampling="taps")
    f = np.zeros((8, 8, 2), np.float32)
    first = (4 * np.arange(8)).astype(np.int32)
    w = np.full((8, 4), 0.25, np.float32)
    bad = first.copy(); bad[7] = 30                      # weights beyond the last row
    with pytest.raises(MMCError):
        s.set_sampling_taps(bad, w, first, w)
    s.set_lowres(f); s.set_flow(np.zeros((32, 32, 1, 2), np.float32)); s.set_state(np.zeros((32, 32, 2)), np.zeros((32, 32, 2)))
    with pytest.raises(MMCError):                        # taps were never given
        s.init()
    s.close()
    s2 = FrameBlockSolver(8, 8, 4, 2, sampling="average")
    with pytest.raises(MMCError):
        s2.set_sampling_taps(first, w, first, w)
    s2.close()


@pytest.mark.parametrize("method,aa,k", [("bicubic", True, KBLUR), ("lanczos3", False, None)])
def test_full_hd_adjoint_and_run(method, aa, k):
    """1080 x 1920 frames with a sampler handed over as rows (interior taps as a composite kernel, the
    deviating border rows as a sparse correction with some ten thousand entries): <K x, y> = <x, K' y> for
    random x, y, which holds whatever the size; then a few iterations stay finite and move the iterate."""
    from superresolution_amd import synthetic
    from superresolution_amd.sampling import dim1_taps
    from superresolution_amd.solver import FrameBlockSolver
    ny, nx, nc = 270, 480, 2
    Ny, Nx = 4 * ny, 4 * nx
    blk = synthetic.make_block(ny, nx, nc, 0, nc, device="cpu")
    s = FrameBlockSolver(ny, nx, 4, nc, sampling="taps")
    fy, wy = dim1_taps(Ny, ny, method, None, aa)
    fx, wx = dim1_taps(Nx, nx, method, None, aa)
    s.set_sampling_taps(fy, wy, fx, wx)
    s.set_lowres(blk["f"]); s.set_blur(k); s.set_flow(blk["v"]); s.set_state(); s.init()
    rng = np.random.default_rng(2)
    u = rng.standard_normal((Ny, Nx, nc)).astype(np.float32)
    w = rng.standard_normal((Ny, Nx, nc)).astype(np.float32)
    Kp, Kg1, Kg2 = s.debug_apply_K(u, w)
    yp = rng.standard_normal(Kp.size).astype(np.float32)
    y1 = rng.standard_normal(Kg1.size).astype(np.float32)
    y2 = rng.standard_normal(Kg2.size).astype(np.float32)
    ku, kw = s.debug_apply_KT(yp, y1, y2)
    d = lambda a, b: float(np.dot(a.astype(np.float64).ravel(order="F"), b.astype(np.float64).ravel(order="F")))
    lhs = d(Kp, yp) + d(Kg1, y1) + d(Kg2, y2)
    rhs = d(u, ku) + d(w, kw)
    scale = sum(np.linalg.norm(a) * np.linalg.norm(b) for a, b in ((Kp, yp), (Kg1, y1), (Kg2, y2)))
    assert abs(lhs - rhs) < 1e-5 * scale
    u0, _ = s.get_state()
    st = s.solve(10, 0.0, 10)
    u1, w1 = s.get_state()
    assert np.isfinite(u1).all() and np.isfinite(w1).all() and 0 < st["res_primal"] < 1.0
    assert float(np.abs(u1 - u0).max()) > 1e-4
    s.close()
