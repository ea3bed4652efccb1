"""Two real processes, one frame block each, sharing the one visible GPU: the
product's sharded solve (FrameBlockSolver + sharding.solve_sharded) with the
halo frames exchanged over torch.distributed.  The backend is gloo with
host-staged frames (RCCL refuses two ranks on one device); everything else --
block set-up, all-reduced h, stream handling, sweep/exchange order -- is the
path bench.py runs with --gpus N."""
import os
import socket

import numpy as np
import pytest

import mmc_oracle as orc
from helpers import make_case, rel_l2, spawn_ranks

pytestmark = pytest.mark.gpu
KBLUR = orc.fspecial_gaussian(7, np.sqrt(0.6))


def _worker(rank, world, port, nc, kappa, direction, iters, out_dir, backend="gloo"):
    import torch
    import torch.distributed as dist
    from superresolution_amd import sharding
    from superresolution_amd.solver import FrameBlockSolver
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    devno = rank if backend == "nccl" else 0
    dev = "cuda:%d" % devno
    if backend == "nccl":
        torch.cuda.set_device(devno)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(devno)
        torch.cuda.set_stream(torch.cuda.Stream())
        f, v, occ = make_case(41, 8, 10, nc, flow="smooth", flow_scale=2.0, with_occ=True)
        P = orc.build_problem(f, v, kappa=kappa, flowDirection=direction, k=KBLUR, occ=occ)
        a, b = sharding.partition_frames(nc, world)[rank]
        lo, hi = sharding.pairs_needed(a, b, nc)
        s0, s1 = sharding.state_frames_needed(a, b, nc)
        s = FrameBlockSolver(8, 10, 4, nc, a, b, alpha=P.alpha, kappa=kappa, h=0.0, flow_direction=direction,
                             device=devno, stream=torch.cuda.current_stream().cuda_stream)
        s.set_lowres(f[:, :, a:b]); s.set_blur(KBLUR); s.set_flow(v[:, :, lo:hi, :], occ[:, :, lo:hi])
        s.set_state(P.u_up[:, :, s0:s1], P.u_up[:, :, s0:s1])
        s.init()
        red = dev if backend == "nccl" else None
        s.h = sharding.sharded_h(s, red)
        ex = sharding.HaloExchanger(sharding.halo_tensors(s), rank, world)
        assert ex.staged == (backend != "nccl")
        st = sharding.solve_sharded(s, ex, iters, 0.0, 10, red)
        u, w = s.get_state()
        np.savez(os.path.join(out_dir, "r%d.npz" % rank), a=a, b=b, u=u, h=s.h,
                 res=np.array([st["res_primal"], st["res_dual"]]), **({} if w is None else {"w": w}))
        s.close()
    finally:
        dist.destroy_process_group()


# (3, 3, ...): one frame per rank; the last rank owns no flow pair, so its share of sum|W u| is zero
@pytest.mark.parametrize("world,nc,kappa,direction", [(2, 5, 0.25, "backward"), (3, 6, float("nan"), "forward-backward"),
                                                      (3, 3, 0.25, "backward")])
def test_two_processes_one_gpu(tmp_path, world, nc, kappa, direction):
    _run_and_check(tmp_path, world, nc, kappa, direction, "gloo")


@pytest.mark.parametrize("world,nc,kappa,direction", [(2, 5, 0.25, "backward"), (2, 4, float("nan"), "forward-backward")])
def test_ranks_on_their_own_gpus_over_rccl(tmp_path, world, nc, kappa, direction):
    """The device-to-device branch of the halo exchange (backend nccl = RCCL, frames sent from the
    library's own buffers, stream-ordered waits).  Needs one GPU per rank."""
    import torch
    if torch.cuda.device_count() < world:
        pytest.skip("needs %d GPUs, %d visible" % (world, torch.cuda.device_count()))
    _run_and_check(tmp_path, world, nc, kappa, direction, "nccl")


def test_one_rank_over_rccl(tmp_path):
    """The RCCL set-up of the sharded driver on the one GPU every box has: communicator bound to the device
    at init, sums all-reduced on the device, the library's stream as torch's current stream.  (The
    device-to-device frames need a second GPU: the test above.)"""
    _run_and_check(tmp_path, 1, 4, 0.25, "backward", "nccl")


def _self_exchange_worker(rank, world, port, out_dir):
    """One rank, RCCL, a middle block of a longer sequence: the frames it would send to its neighbours go to
    its own ghost slots instead (RCCL accepts a send and a receive of one rank to itself in one batch)."""
    import torch
    import torch.distributed as dist
    from superresolution_amd import sharding
    from superresolution_amd.solver import FrameBlockSolver
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        torch.cuda.set_stream(torch.cuda.Stream())
        nc, a, b = 6, 2, 4
        f, v, occ = make_case(43, 8, 10, nc, flow="smooth", flow_scale=2.0, with_occ=True)
        P = orc.build_problem(f, v, k=KBLUR, occ=occ)
        lo, hi = sharding.pairs_needed(a, b, nc)
        s0, s1 = sharding.state_frames_needed(a, b, nc)
        s = FrameBlockSolver(8, 10, 4, nc, a, b, alpha=P.alpha, kappa=0.25, h=P.h, device=0,
                             stream=torch.cuda.current_stream().cuda_stream)
        s.set_lowres(f[:, :, a:b]); s.set_blur(KBLUR); s.set_flow(v[:, :, lo:hi, :], occ[:, :, lo:hi])
        s.set_state(P.u_up[:, :, s0:s1], P.u_up[:, :, s0:s1]); s.init()
        t = sharding.halo_tensors(s)
        ok = True
        for it in range(3):
            # the sweeps are queued on torch's current stream; the exchange is posted right behind them without a
            # host synchronisation, as solve_sharded does
            s.sweep_primal(0, b - a)
            ops = [dist.P2POp(dist.isend, x, 0) for x in t["send_prev"]] + [dist.P2POp(dist.irecv, x, 0) for x in t["recv_next"]]
            for r in dist.batch_isend_irecv(ops):
                r.wait()
            s.sweep_dual(0, b - a)
            ops = [dist.P2POp(dist.isend, x, 0) for x in t["send_next"]] + [dist.P2POp(dist.irecv, x, 0) for x in t["recv_prev"]]
            for r in dist.batch_isend_irecv(ops):
                r.wait()
            torch.cuda.synchronize()
            for sname, rname in (("send_prev", "recv_next"), ("send_next", "recv_prev")):
                for x, y in zip(t[sname], t[rname]):
                    ok = ok and bool(torch.equal(x, y)) and bool(torch.isfinite(x).all())
            ok = ok and float(t["send_prev"][0].abs().max()) > 0
        np.savez(os.path.join(out_dir, "self.npz"), ok=ok)
        s.close()
    finally:
        dist.destroy_process_group()


def _two_blocks_worker(rank, world, port, kappa, direction, out_dir):
    """One rank, RCCL, two frame blocks of one sequence on the same GPU: every halo frame travels as an RCCL
    send matched by a receive of the same rank, in the order solve_sharded posts them for two ranks; nothing
    but the stream orders the transfers against the sweeps."""
    import torch
    import torch.distributed as dist
    from superresolution_amd import sharding
    from superresolution_amd.solver import FrameBlockSolver
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        torch.cuda.set_stream(torch.cuda.Stream())
        nc, iters = 6, 30
        f, v, occ = make_case(44, 8, 10, nc, flow="smooth", flow_scale=2.0, with_occ=True)
        P = orc.build_problem(f, v, kappa=kappa, flowDirection=direction, k=KBLUR, occ=occ)
        blocks = []
        for a, b in sharding.partition_frames(nc, 2):
            lo, hi = sharding.pairs_needed(a, b, nc)
            s0, s1 = sharding.state_frames_needed(a, b, nc)
            s = FrameBlockSolver(8, 10, 4, nc, a, b, alpha=P.alpha, kappa=kappa, h=P.h, flow_direction=direction, device=0,
                                 stream=torch.cuda.current_stream().cuda_stream)
            s.set_lowres(f[:, :, a:b]); s.set_blur(KBLUR); s.set_flow(v[:, :, lo:hi, :], occ[:, :, lo:hi])
            s.set_state(P.u_up[:, :, s0:s1], P.u_up[:, :, s0:s1]); s.init()
            blocks.append((s, sharding.halo_tensors(s), b - a))
        (A, tA, LA), (B, tB, LB) = blocks

        def move(src, dst):
            ops = [dist.P2POp(dist.isend, x, 0) for x in src if x is not None] + [dist.P2POp(dist.irecv, x, 0) for x in dst if x is not None]
            for r in dist.batch_isend_irecv(ops):
                r.wait()

        for it in range(iters):
            A.sweep_primal(0, LA)
            B.sweep_primal(0, LB)
            move(tB["send_prev"], tA["recv_next"])          # xbar of B's first frame -> A's ghost
            A.sweep_dual(0, LA)
            B.sweep_dual(0, LB)
            move(tA["send_next"], tB["recv_prev"])          # temporal duals of A's last frame -> B's ghost
        torch.cuda.synchronize()
        uA, wA = A.get_state()
        uB, wB = B.get_state()
        u = np.concatenate([uA, uB], axis=2)
        out = {"u": u}
        if wA is not None:
            out["w"] = np.concatenate([wA, wB], axis=2)
        np.savez(os.path.join(out_dir, "two.npz"), **out)
        A.close(); B.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kappa,direction", [(0.25, "backward"), (float("nan"), "forward-backward"), (0.25, "forward")])
def test_two_blocks_exchange_over_rccl_on_one_gpu(tmp_path, kappa, direction):
    """A sequence in two frame blocks whose halo frames travel over RCCL (one rank, sends matched by its own
    receives), against the single-block oracle: the device-to-device branch of the sharded solve on one GPU."""
    iters = 30
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    spawn_ranks("test_gpu_multiprocess", "_two_blocks_worker", (1, port, kappa, direction, str(tmp_path)), 1)
    f, v, occ = make_case(44, 8, 10, 6, flow="smooth", flow_scale=2.0, with_occ=True)
    P = orc.build_problem(f, v, kappa=kappa, flowDirection=direction, k=KBLUR, occ=occ)
    x, _, _ = orc.pdhg(P, iters)
    u_ref, w_ref = orc.split_primal(P, x)
    d = np.load(os.path.join(str(tmp_path), "two.npz"))
    assert rel_l2(d["u"], u_ref) < 1e-4
    if w_ref is not None:
        assert np.abs(d["w"] - w_ref).max() < 1e-4


def test_rccl_moves_library_frames_on_one_gpu(tmp_path):
    """The device-to-device transfer of the halo frames over RCCL, on the one GPU every box has: the buffers are
    the library's own allocations (not torch's), the transfers are ordered against the sweeps by the stream alone."""
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    spawn_ranks("test_gpu_multiprocess", "_self_exchange_worker", (1, port, str(tmp_path)), 1)
    assert bool(np.load(os.path.join(str(tmp_path), "self.npz"))["ok"])


def _run_and_check(tmp_path, world, nc, kappa, direction, backend):
    iters = 30
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    spawn_ranks("test_gpu_multiprocess", "_worker", (world, port, nc, kappa, direction, iters, str(tmp_path), backend), world)
    f, v, occ = make_case(41, 8, 10, nc, flow="smooth", flow_scale=2.0, with_occ=True)
    P = orc.build_problem(f, v, kappa=kappa, flowDirection=direction, k=KBLUR, occ=occ)
    x, _, chg = orc.pdhg(P, iters)
    u_ref, w_ref = orc.split_primal(P, x)
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), "r%d.npz" % r))
        a, b = int(d["a"]), int(d["b"])
        assert abs(float(d["h"]) - P.h) < 1e-5 * P.h
        assert rel_l2(d["u"], u_ref[:, :, a:b]) < 1e-4
        if w_ref is not None:
            assert np.abs(d["w"] - w_ref[:, :, a:b]).max() < 1e-4
        assert abs(d["res"][0] - chg[0] / chg[1]) < 2e-2 * chg[0] / chg[1] + 1e-7
