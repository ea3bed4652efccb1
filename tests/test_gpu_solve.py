"""GPU parity of the full PDHG solve against the oracle at equal iteration
count.  Tolerance: relative L2 <= 1e-4 in fp32 (the bar the task states)."""
import numpy as np
import pytest

import mmc_oracle as orc
from helpers import make_case, rel_l2

pytestmark = pytest.mark.gpu

KBLUR = orc.fspecial_gaussian(7, np.sqrt(0.6))
TOL = 1e-4


def _run(f, v, occ, kappa, direction, iters, k=KBLUR, check_every=0):
    from superresolution_amd.solver import FrameBlockSolver
    P = orc.build_problem(f, v, kappa=kappa, flowDirection=direction, k=k, occ=occ)
    x, y, chg = orc.pdhg(P, iters)
    u_ref, w_ref = orc.split_primal(P, x)
    ny, nx, nc = f.shape
    s = FrameBlockSolver(ny, nx, 4, nc, alpha=P.alpha, kappa=kappa, h=0.0, flow_direction=direction)
    s.set_lowres(f); s.set_blur(k); s.set_flow(v, occ); s.set_state(P.u_up, P.u_up); s.init()
    st = s.solve(iters, 0.0, check_every)
    u, w = s.get_state()
    e = s.energy()
    s.close()
    return P, x, chg, u_ref, w_ref, u, w, st, e


@pytest.mark.parametrize("direction", ["backward", "forward", "forward-backward"])
@pytest.mark.parametrize("kappa", [0.25, 1.0, float("nan")])
def test_solve_parity(direction, kappa):
    f, v, occ = make_case(21, 10, 12, 4, flow="smooth", flow_scale=2.0, with_occ=True)
    P, x, chg, u_ref, w_ref, u, w, st, e = _run(f, v, occ, kappa, direction, 150)
    assert st["iterations"] == 150
    assert rel_l2(u, u_ref) < TOL
    if w_ref is not None:
        assert rel_l2(w, w_ref) < 10 * TOL * max(1.0, np.linalg.norm(u_ref) / max(np.linalg.norm(w_ref), 1e-12))
    # residuals of the last step and the primal energy
    assert abs(st["res_primal"] - chg[0] / chg[1]) <= 2e-2 * (chg[0] / chg[1]) + 1e-7
    assert abs(st["res_dual"] - chg[2] / chg[3]) <= 2e-2 * (chg[2] / chg[3]) + 1e-7
    e_ref = orc.primal_energy(P, x)
    assert abs(e - e_ref) <= 1e-3 * abs(e_ref)


def test_solve_parity_random_flow_and_ragged_tiles():
    """Non-smooth flow (large source boxes) and a frame that is not a multiple of the tile."""
    f, v, occ = make_case(8, 19, 5, 3, flow="random", flow_scale=3.0)
    P, x, chg, u_ref, w_ref, u, w, st, e = _run(f, v, occ, 0.25, "backward", 100)
    assert rel_l2(u, u_ref) < TOL


def test_single_frame_and_two_frames():
    for nc in (1, 2):
        f, v, occ = make_case(3, 8, 8, nc)
        P, x, chg, u_ref, w_ref, u, w, st, e = _run(f, v, occ, 0.25, "backward", 60)
        assert rel_l2(u, u_ref) < TOL


def test_energy_decreases_and_tolerance_stop():
    from superresolution_amd.solver import FrameBlockSolver
    f, v, _ = make_case(5, 12, 12, 3)
    P = orc.build_problem(f, v, k=KBLUR)
    s = FrameBlockSolver(12, 12, 4, 3)
    s.set_lowres(f); s.set_blur(KBLUR); s.set_flow(v); s.set_state(P.u_up, P.u_up); s.init()
    from superresolution_amd import _capi
    s.set_diagnostics(_capi.DIAG_OFF)        # stop on the relative change of the iterates (the residual test: test_gpu_diagnostics.py)
    e0 = s.energy()
    seen = []
    st = s.solve(4000, 1e-4, 50, callback=lambda it, d: seen.append((it, d["res_primal"])) and False)
    e1 = s.energy()
    assert e1 < e0
    assert st["converged"] and st["iterations"] < 4000 and st["iterations"] % 50 == 0
    assert max(st["res_primal"], st["res_dual"]) < 1e-4
    assert [it for it, _ in seen] == list(range(50, st["iterations"] + 1, 50))
    s.close()


@pytest.mark.parametrize("kappa", [0.25, float("nan")])
def test_solve_parity_interior_tiles(kappa):
    """A frame of 192x112 HR pixels has workgroup tiles (64x16) whose blur window, low-res ring and
    warp tap region lie wholly inside the image, i.e. the vectorised interior paths of both sweeps,
    next to border tiles and a ragged last tile column."""
    f, v, occ = make_case(31, 48, 28, 3, flow="smooth", flow_scale=1.5, with_occ=True)
    P, x, chg, u_ref, w_ref, u, w, st, e = _run(f, v, occ, kappa, "backward", 60)
    assert rel_l2(u, u_ref) < TOL
    e_ref = orc.primal_energy(P, x)
    assert abs(e - e_ref) <= 1e-3 * abs(e_ref)


def test_solve_parity_violent_flow():
    """Flow of +-12 pixels per frame, random per pixel: tap regions and source boxes outgrow what a
    tile stages on chip, so the global-memory gather and multi-pass scatter run."""
    f, v, occ = make_case(32, 40, 24, 3, flow="random", flow_scale=12.0)
    P, x, chg, u_ref, w_ref, u, w, st, e = _run(f, v, occ, 0.25, "forward-backward", 40)
    assert rel_l2(u, u_ref) < TOL


def _with_outliers(v, seed, fraction, amplitude):
    """Replaces a fraction of the vectors by uniform ones in [-amplitude, amplitude]^2."""
    rng = np.random.default_rng(seed)
    v = v.copy()
    Ny, Nx, npairs, _ = v.shape
    for i in range(npairs):
        idx = rng.choice(Ny * Nx, size=max(int(fraction * Ny * Nx), 1), replace=False)
        r, c = idx % Ny, idx // Ny
        v[r, c, i, :] = rng.uniform(-amplitude, amplitude, (idx.size, 2))
    return v.astype(np.float32).astype(np.float64)


@pytest.mark.parametrize("direction,kappa", [("backward", 0.25), ("forward-backward", 0.25), ("forward", float("nan"))])
def test_solve_parity_flow_with_outliers(direction, kappa):
    """A smooth flow in which 1 % of the vectors are strays of up to 60 pixels (what an estimator
    leaves at occlusions).  The source rectangles and tap regions stay tight around the regular
    part of the flow; the strays go through the per-tile overflow lists of the primal sweep and
    the per-pixel global gather of the dual sweep."""
    f, v, occ = make_case(35, 48, 28, 3, flow="smooth", flow_scale=1.5, with_occ=True)
    v = _with_outliers(v, 7, 0.01, 60.0)
    P, x, chg, u_ref, w_ref, u, w, st, e = _run(f, v, occ, kappa, direction, 60)
    assert rel_l2(u, u_ref) < TOL
    if w_ref is not None:
        assert np.abs(w - w_ref).max() < TOL
    e_ref = orc.primal_energy(P, x)
    assert abs(e - e_ref) <= 1e-3 * abs(e_ref)


def test_motion_boundary_inside_tiles():
    """Two motions 20 pixels apart split every tile: no pixel is close to the tile's mean
    displacement, so every source goes through the overflow lists and every tap through the
    global gather."""
    f, v, occ = make_case(36, 32, 24, 3, flow="zero")
    Ny, Nx = v.shape[:2]
    half = (np.arange(Ny)[:, None] // 8 + np.arange(Nx)[None, :] // 8) % 2 == 0
    for i in range(v.shape[2]):
        v[:, :, i, 0] = np.where(half, 10.3, -9.6)
        v[:, :, i, 1] = np.where(half, -7.2, 12.4)
    v = v.astype(np.float32).astype(np.float64)
    P, x, chg, u_ref, w_ref, u, w, st, e = _run(f, v, occ, 0.25, "backward", 40)
    assert rel_l2(u, u_ref) < TOL


@pytest.mark.parametrize("factor,ny,nx", [(2, 14, 18), (3, 7, 8)])
def test_solve_parity_other_zoom_factors(factor, ny, nx):
    """Update sweeps for zoom factors other than 4: zoom 2, and zoom 3, which does not divide the tile
    and here gives a row count that is not a multiple of 4 (short last row quads); 'stride' sampling
    with a 5x5 blur."""
    from superresolution_amd.solver import FrameBlockSolver
    f, v, _ = make_case(33, ny, nx, 3, factor=factor, flow="smooth", flow_scale=1.5)
    k = orc.fspecial_gaussian(5, 1.0)
    P = orc.build_problem(f, v, factor=factor, k=k, interpMethod="stride")
    x, y, chg = orc.pdhg(P, 80)
    u_ref, w_ref = orc.split_primal(P, x)
    s = FrameBlockSolver(ny, nx, factor, 3, alpha=P.alpha, kappa=0.25, h=0.0, sampling="stride")
    s.set_lowres(f); s.set_blur(k); s.set_flow(v); s.set_state(P.u_up, P.u_up); s.init()
    s.solve(80, 0.0, 0)
    u, w = s.get_state()
    s.close()
    assert rel_l2(u, u_ref) < TOL
