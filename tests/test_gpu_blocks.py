"""GPU tests of the frame-block (ghost frame) logic of the HIP kernels: a
sequence split into blocks on ONE GPU, boundary frames copied between the
blocks' halo buffers exactly as the RCCL exchange does, must reproduce the
single-block HIP solve and the oracle."""
import numpy as np
import pytest

import mmc_oracle as orc
from helpers import make_case, rel_l2

pytestmark = pytest.mark.gpu
KBLUR = orc.fspecial_gaussian(7, np.sqrt(0.6))


def _make_blocks(f, v, occ, P, kappa, direction, cuts):
    from superresolution_amd.solver import FrameBlockSolver
    from superresolution_amd import sharding
    ny, nx, nc = f.shape
    blocks = []
    for a, b in cuts:
        s = FrameBlockSolver(ny, nx, 4, nc, a, b, alpha=P.alpha, kappa=kappa, h=0.0, flow_direction=direction)
        lo, hi = sharding.pairs_needed(a, b, nc)
        s0, s1 = sharding.state_frames_needed(a, b, nc)
        s.set_lowres(f[:, :, a:b])
        s.set_blur(KBLUR)
        s.set_flow(v[:, :, lo:hi, :], None if occ is None else occ[:, :, lo:hi])
        s.set_state(P.u_up[:, :, s0:s1], P.u_up[:, :, s0:s1])
        s.init()
        blocks.append(s)
    return blocks


@pytest.mark.parametrize("direction", ["backward", "forward", "forward-backward"])
@pytest.mark.parametrize("kappa", [0.25, float("nan")])
def test_blocks_with_halo_copies_match_single_block(direction, kappa):
    import torch
    from superresolution_amd import sharding
    nc, iters = 5, 40
    f, v, occ = make_case(17, 8, 10, nc, flow="smooth", flow_scale=2.0, with_occ=True)
    P = orc.build_problem(f, v, kappa=kappa, flowDirection=direction, k=KBLUR, occ=occ)
    x_ref, _, chg = orc.pdhg(P, iters)
    u_ref, w_ref = orc.split_primal(P, x_ref)
    cuts = [(0, 2), (2, 3), (3, 5)]                 # includes a one-frame block
    blocks = _make_blocks(f, v, occ, P, kappa, direction, cuts)
    # h from the all-reduced partial sums (MultiframeMotionCoupling.m:431-437)
    parts = np.array([b.h_partials() for b in blocks]).sum(axis=0)
    h = parts[0] / parts[1]
    assert abs(h - P.h) < 1e-5 * P.h
    for b in blocks:
        b.h = h
    halos = [sharding.halo_tensors(b) for b in blocks]     # zero-copy torch views of the device frames

    def exchange(src_key, dst_key, src_blocks, dst_blocks):
        for b in blocks:
            b.sync()
        for i_src, i_dst in zip(src_blocks, dst_blocks):
            for k in range(2):
                if halos[i_src][src_key][k] is not None:
                    halos[i_dst][dst_key][k].copy_(halos[i_src][src_key][k])
        torch.cuda.synchronize()

    nb = len(blocks)
    for it in range(iters):
        last = it == iters - 1
        for b in blocks:
            if last:
                b.sums_reset()
            b.sweep_primal(0, b.L, last)
        exchange("send_prev", "recv_next", range(1, nb), range(0, nb - 1))
        for b in blocks:
            b.sweep_dual(0, b.L, last)
        exchange("send_next", "recv_prev", range(0, nb - 1), range(1, nb))
    sums = np.array([b.sums_read() for b in blocks]).sum(axis=0)
    assert abs(np.sqrt(sums[0] / sums[1]) - chg[0] / chg[1]) < 2e-2 * chg[0] / chg[1] + 1e-7
    u = np.concatenate([b.get_state()[0] for b in blocks], axis=2)
    assert rel_l2(u, u_ref) < 1e-4
    if w_ref is not None:
        w = np.concatenate([b.get_state()[1] for b in blocks], axis=2)
        assert np.abs(w - w_ref).max() < 1e-4
    for b in blocks:
        b.close()


def test_block_solve_entry_point_refuses_blocks_with_neighbours():
    from superresolution_amd import MMCError
    f, v, occ = make_case(2, 8, 8, 4)
    P = orc.build_problem(f, v, k=KBLUR)
    blocks = _make_blocks(f, v, None, P, 0.25, "backward", [(0, 2), (2, 4)])
    with pytest.raises(MMCError, match="neighbours"):
        blocks[0].solve(5)
    for b in blocks:
        b.close()


def test_block_with_neighbours_needs_h_before_sweeping():
    """A block of a sharded sequence only holds part of the sums behind h (a trailing one-frame
    block holds no pair at all): mmc_init leaves h unset there and the sweeps refuse to run until
    the all-reduced value has been set."""
    from superresolution_amd import MMCError
    f, v, occ = make_case(3, 8, 8, 3)
    P = orc.build_problem(f, v, k=KBLUR)
    blocks = _make_blocks(f, v, None, P, 0.25, "backward", [(0, 2), (2, 3)])
    last = blocks[1]
    assert last.h_partials()[0] == 0.0                      # no pair owned
    with pytest.raises(MMCError, match="h is not set"):
        last.sweep_primal(0, 1)
    parts = np.array([b.h_partials() for b in blocks]).sum(axis=0)
    for b in blocks:
        b.h = parts[0] / parts[1]
    last.sweep_primal(0, 1)
    last.sync()
    for b in blocks:
        b.close()


def test_sharded_loop_refuses_a_private_library_stream():
    """solve_sharded orders the exchange against torch's current stream; a solver on the library's
    own stream would race with it."""
    import torch
    from superresolution_amd import sharding
    f, v, occ = make_case(3, 8, 8, 4)
    P = orc.build_problem(f, v, k=KBLUR)
    blocks = _make_blocks(f, v, None, P, 0.25, "backward", [(0, 2), (2, 4)])      # stream=None

    class FakeExchanger(object):
        on_device, prev, next = True, None, 1

    with pytest.raises(RuntimeError, match="current_stream"):
        sharding.solve_sharded(blocks[0], FakeExchanger(), 1)
    for b in blocks:
        b.close()


def test_blocks_with_a_generic_sampler():
    """Frame blocks with a sampler given by taps (bicubic with antialiasing): the sparse border rows
    and their per-frame correction planes are indexed by local frames in every block."""
    import torch
    from superresolution_amd import sharding
    from superresolution_amd.sampling import dim1_taps
    from superresolution_amd.solver import FrameBlockSolver
    nc, iters = 4, 30
    ny, nx = 10, 9
    f, v, _ = make_case(19, ny, nx, nc, flow="smooth", flow_scale=1.5)
    P = orc.build_problem(f, v, k=KBLUR, interpMethod="bicubic", interpAA=True)
    x_ref, _, _ = orc.pdhg(P, iters)
    u_ref, _ = orc.split_primal(P, x_ref)
    fy, wy = dim1_taps(4 * ny, ny, "bicubic", None, True)
    fx, wx = dim1_taps(4 * nx, nx, "bicubic", None, True)
    blocks = []
    for a, b in [(0, 1), (1, 4)]:
        s = FrameBlockSolver(ny, nx, 4, nc, a, b, alpha=P.alpha, kappa=0.25, h=P.h, sampling="taps")
        lo, hi = sharding.pairs_needed(a, b, nc)
        s0, s1 = sharding.state_frames_needed(a, b, nc)
        s.set_sampling_taps(fy, wy, fx, wx)
        s.set_lowres(f[:, :, a:b]); s.set_blur(KBLUR); s.set_flow(v[:, :, lo:hi, :])
        s.set_state(P.u_up[:, :, s0:s1], P.u_up[:, :, s0:s1]); s.init()
        blocks.append(s)
    halos = [sharding.halo_tensors(b) for b in blocks]

    def exchange(src_key, dst_key, i_src, i_dst):
        for b in blocks:
            b.sync()
        for k in range(2):
            if halos[i_src][src_key][k] is not None:
                halos[i_dst][dst_key][k].copy_(halos[i_src][src_key][k])
        torch.cuda.synchronize()

    for it in range(iters):
        for b in blocks:
            b.sweep_primal(0, b.L, False)
        exchange("send_prev", "recv_next", 1, 0)
        for b in blocks:
            b.sweep_dual(0, b.L, False)
        exchange("send_next", "recv_prev", 0, 1)
    u = np.concatenate([b.get_state()[0] for b in blocks], axis=2)
    assert rel_l2(u, u_ref) < 1e-4
    for b in blocks:
        b.close()
