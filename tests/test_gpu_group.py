"""GPU tests of the single-process group driver (mmc_group_*): one sequence cut into frame blocks
that one host thread drives, each block on its own stream with peer-to-peer halo copies.  On the
one-GPU test box all blocks share device 0; the event ordering, the copies and the sums over blocks
are the same code that runs with one block per GPU."""
import numpy as np
import pytest

import mmc_oracle as orc
from helpers import make_case, rel_l2

pytestmark = pytest.mark.gpu
KBLUR = orc.fspecial_gaussian(7, np.sqrt(0.6))


def _single(f, v, occ, u0, kappa, direction, h, iters):
    from superresolution_amd.solver import FrameBlockSolver
    ny, nx, nc = f.shape
    with FrameBlockSolver(ny, nx, 4, nc, alpha=0.01, kappa=kappa, h=h, flow_direction=direction) as s:
        s.set_lowres(f); s.set_blur(KBLUR); s.set_flow(v, occ); s.set_state(u0, u0); s.init()
        s.solve(iters)
        return s.get_state() + (s.h,)


@pytest.mark.parametrize("direction", ["backward", "forward", "forward-backward"])
@pytest.mark.parametrize("kappa", [0.25, 1.0, float("nan")])
def test_group_equals_single_block_bit_for_bit_and_the_oracle(direction, kappa):
    from superresolution_amd.solver import SequenceGroupSolver
    nc, iters = 5, 40
    f, v, occ = make_case(23, 8, 10, nc, flow="smooth", flow_scale=2.0, with_occ=True)
    P = orc.build_problem(f, v, kappa=kappa, flowDirection=direction, k=KBLUR, occ=occ)
    x_ref, _, chg = orc.pdhg(P, iters)
    u_ref, w_ref = orc.split_primal(P, x_ref)
    u1, w1, _ = _single(f, v, occ, P.u_up, kappa, direction, P.h, iters)
    with SequenceGroupSolver(8, 10, 4, nc, nblocks=3, alpha=0.01, kappa=kappa, h=P.h, flow_direction=direction) as g:
        assert g.num_blocks == 3
        assert [g.block_range(i)[:2] for i in range(3)] == [(0, 2), (2, 4), (4, 5)]     # ends in a one-frame block
        g.set_lowres(f); g.set_blur(KBLUR); g.set_flow(v, occ); g.set_state(P.u_up, P.u_up); g.init()
        st = g.solve(iters)
        u, w = g.get_state()
    assert st["iterations"] == iters
    assert np.array_equal(u, u1)
    assert rel_l2(u, u_ref) < 1e-4
    if w_ref is not None:
        assert np.array_equal(w, w1)
        assert np.abs(w - w_ref).max() < 1e-4
    assert abs(st["res_primal"] - chg[0] / chg[1]) < 2e-2 * chg[0] / chg[1] + 1e-7


def test_group_computes_h_and_the_bicubic_start_point_over_all_blocks():
    """h <= 0 and no start point: every block makes its own bicubic frames, the first frame of each
    block travels to its predecessor's ghost slot, and h is the ratio of the sums over all blocks
    (MultiframeMotionCoupling.m:415-437)."""
    from superresolution_amd.solver import SequenceGroupSolver
    nc = 6
    f, v, _ = make_case(5, 12, 9, nc, flow="smooth", flow_scale=1.5)
    P = orc.build_problem(f, v, k=KBLUR)
    x_ref, _, _ = orc.pdhg(P, 30)
    u_ref, w_ref = orc.split_primal(P, x_ref)
    with SequenceGroupSolver(12, 9, 4, nc, nblocks=4) as g:
        g.set_lowres(f); g.set_blur(KBLUR); g.set_flow(v); g.set_state(); g.init()
        assert abs(g.h - P.h) < 1e-5 * P.h
        g.solve(30)
        u, w = g.get_state()
    assert rel_l2(u, u_ref) < 1e-4 and np.abs(w - w_ref).max() < 1e-4


def test_group_stops_on_tolerance_and_reports_progress():
    from superresolution_amd.solver import SequenceGroupSolver
    f, v, _ = make_case(7, 8, 8, 4)
    seen = []
    with SequenceGroupSolver(8, 8, 4, 4, nblocks=2) as g:
        g.set_lowres(f); g.set_blur(KBLUR); g.set_flow(v); g.set_state(); g.init()
        st = g.solve(4000, tol=1e-3, check_every=50, callback=lambda it, s: seen.append(it) and False)
    assert st["converged"] and st["iterations"] < 4000 and st["iterations"] % 50 == 0
    assert seen and seen[-1] == st["iterations"]
    # a solve with a tolerance and a callback stops on the absolute + relative residual test
    # (MultiframeMotionCoupling.m:357-363); IC model: 2 primal and 6 + 1/16 dual unknowns per pixel
    tol, M = 1e-3, 32 * 32 * 4
    assert st["opt_res_primal"] <= tol * np.sqrt(2 * M) + tol * st["norm_KTy"]
    assert st["opt_res_dual"] <= tol * np.sqrt(6 * M + M / 16) + tol * st["norm_Kx"]
    assert st["gap"] > -1e-3 * abs(st["primal_energy"])
    # the same sequence as one block stops at the same check
    from superresolution_amd.solver import FrameBlockSolver
    with FrameBlockSolver(8, 8, 4, 4) as s:
        s.set_lowres(f); s.set_blur(KBLUR); s.set_flow(v); s.set_state(); s.init()
        st1 = s.solve(4000, 1e-3, 50, callback=lambda it, d: False)
    assert st1["iterations"] == st["iterations"]
    assert abs(st1["primal_energy"] - st["primal_energy"]) < 1e-6 * abs(st["primal_energy"])


def test_group_with_fixed_first_frame_and_flow_swap():
    """The first-frame constraint reaches the block that holds frame 0; a flow swap on a live group
    keeps the iterate (variants/MultiframeMotionCouplingAlternating.m:108-159)."""
    from superresolution_amd.solver import FrameBlockSolver, SequenceGroupSolver
    nc = 4
    f, v, _ = make_case(31, 8, 8, nc, flow="smooth", flow_scale=1.0)
    _, v2, _ = make_case(32, 8, 8, nc, flow="smooth", flow_scale=1.0)
    P = orc.build_problem(f, v, k=KBLUR)
    rng = np.random.default_rng(0)
    u0 = rng.random((32, 32)).astype(np.float32)
    w0 = (0.1 * rng.random((32, 32))).astype(np.float32)

    def drive(s):
        s.set_lowres(f); s.set_blur(KBLUR); s.set_flow(v); s.set_state(P.u_up, P.u_up)
        s.set_fixed_frame(u0, w0); s.init(); s.solve(25)
        s.set_flow(v2); s.init(); s.solve(25)
        return s.get_state()

    with FrameBlockSolver(8, 8, 4, nc, h=P.h) as s:
        u1, w1 = drive(s)
    with SequenceGroupSolver(8, 8, 4, nc, nblocks=2, h=P.h) as g:
        u2, w2 = drive(g)
    assert np.array_equal(u1, u2) and np.array_equal(w1, w2)


def test_group_at_full_hd_two_blocks():
    """Two blocks of three 1080p frames each (the per-GPU shape of the 240-frame sequence cut over
    GPUs) against one block of six: identical iterates."""
    from superresolution_amd import synthetic
    from superresolution_amd.solver import FrameBlockSolver, SequenceGroupSolver
    ny, nx, nc, iters = 270, 480, 6, 12
    blk = synthetic.make_block(ny, nx, nc, 0, nc, flow_kind="outliers")
    f, v = blk["f"], blk["v"]
    with FrameBlockSolver(ny, nx, 4, nc, h=0.3) as s:
        s.set_lowres(f); s.set_blur(KBLUR); s.set_flow(v); s.set_state(); s.init(); s.solve(iters)
        u1, w1 = s.get_state()
    with SequenceGroupSolver(ny, nx, 4, nc, nblocks=2, h=0.3) as g:
        g.set_lowres(f); g.set_blur(KBLUR); g.set_flow(v); g.set_state(); g.init(); g.solve(iters)
        u2, w2 = g.get_state()
    assert np.array_equal(u1, u2) and np.array_equal(w1, w2)
    assert np.isfinite(u2).all() and float(np.abs(u2 - w2).max()) > 1e-3


def test_group_rejects_bad_arguments():
    from superresolution_amd import MMCError
    from superresolution_amd.solver import SequenceGroupSolver
    with pytest.raises(MMCError, match="between 1 and nc_global"):
        SequenceGroupSolver(8, 8, 4, 3, nblocks=4)
    with pytest.raises(MMCError, match="device ordinal"):
        SequenceGroupSolver(8, 8, 4, 3, nblocks=2, devices=[0, 99])
    with SequenceGroupSolver(8, 8, 4, 3, nblocks=2) as g:
        with pytest.raises(MMCError, match="No initialization"):
            g.solve(1)


def test_group_with_one_block_per_gpu():
    """One frame block per visible GPU (up to four), peer-to-peer halo copies between devices: the
    iterates equal the single-block solve bit for bit.  Runs wherever at least two GPUs are visible."""
    import torch
    from superresolution_amd.solver import SequenceGroupSolver
    ndev = min(torch.cuda.device_count(), 4)
    if ndev < 2:
        pytest.skip("needs 2 GPUs, %d visible" % torch.cuda.device_count())
    nc, iters = 2 * ndev + 1, 30
    f, v, occ = make_case(29, 16, 12, nc, flow="smooth", flow_scale=2.0, with_occ=True)
    P = orc.build_problem(f, v, k=KBLUR, occ=occ)
    u1, w1, _ = _single(f, v, occ, P.u_up, 0.25, "backward", P.h, iters)
    with SequenceGroupSolver(16, 12, 4, nc, nblocks=ndev, devices=list(range(ndev)), h=P.h) as g:
        assert [g.block_range(i)[2] for i in range(ndev)] == list(range(ndev))
        g.set_lowres(f); g.set_blur(KBLUR); g.set_flow(v, occ); g.set_state(P.u_up, P.u_up); g.init()
        g.solve(iters)
        u, w = g.get_state()
    assert np.array_equal(u, u1) and np.array_equal(w, w1)
