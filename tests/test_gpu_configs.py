"""Every benchmark configuration at its real frame size, HIP against the C restatement of the
solve (oracle/mmc_oracle_c.c, float32, explicit sparse operators) at equal iteration counts.

  config 1  tube3, 600x600x3, zoom 4 (testfile.m:22-31): the bundled scene with Lucas-Kanade
            flows (tests/golden/tube3_full.npz), once with the infimal-convolution model at the
            flexBox 'fast' cap of 800 iterations (MultiframeMotionCoupling.m:798-802) and once
            with the additive model of comp_mode 'fastest' (:283-290, :515-533); PSNR of the
            central frame on the 20:end-20 crop within 0.02 dB (testfile.m:71-75)
  config 2  calendar size, 576x720x13 (testfile_full_set.m:19-23), 'accurate' parameters;
            the Vid4 data is not bundled, the scene is the synthetic one
  config 3  1080x1920: tests/test_gpu_end_to_end.py::test_full_size_properties, and here the
            step sizes of the update sweep at that size
  config 4  1080x1920 frames sharded over two processes (three frames each) that share the GPU
  config 5  2160x3840x3, which walks the tiles in sub-ranges

Bars: relative L2 of u <= 1e-4, max-abs of u and w <= 1e-4 (w is compared absolutely: it is
an order of magnitude smaller than u).  Parity is unpinned against real prost / flexBox output.
"""
import os
import socket

import numpy as np
import pytest

import mmc_oracle as orc
import mmc_oracle_c as occ
from helpers import rel_l2

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KBLUR = orc.fspecial_gaussian(7, np.sqrt(0.6))


def _need_c_oracle():
    if not occ.available():
        pytest.skip("C oracle not built")


def _hip_solve(f, v, u0, iters, kappa=0.25, direction="backward"):
    from superresolution_amd.solver import FrameBlockSolver
    ny, nx, nc = f.shape
    with FrameBlockSolver(ny, nx, 4, nc, kappa=kappa, flow_direction=direction) as s:
        s.set_lowres(f); s.set_blur(KBLUR); s.set_flow(v); s.set_state(u0, u0); s.init()
        st = s.solve(iters, 0.0, 0)
        u, w = s.get_state()
        return u, w, s.h, st


def _compare(u, w, u_ref, w_ref):
    assert rel_l2(u, u_ref) <= 1e-4
    assert np.abs(u - u_ref).max() <= 1e-4
    if w_ref is not None:
        assert np.abs(w - w_ref).max() <= 1e-4


@pytest.mark.parametrize("kappa,iters", [(0.25, 800), (float("nan"), 2500)])
def test_config1_tube3_full_frames(kappa, iters):
    _need_c_oracle()
    g = np.load(os.path.join(GOLD, "tube3_full.npz"))
    truth = g["truth_u8"].astype(np.float64) / 255.0
    lr = g["lr"].astype(np.float64)
    assert truth.shape == (600, 600, 3) and lr.shape == (150, 150, 3)
    # flows estimated on the low-res frames, upscaled and multiplied by the factor (:955-959)
    v = np.stack([np.stack([orc.imresize(g["flow_lr"][:, :, i, c].astype(np.float64), 4) * 4
                            for c in range(2)], axis=2) for i in range(2)], axis=2).astype(np.float32)
    u_up = np.stack([orc.imresize(lr[:, :, i], 4) for i in range(3)], axis=2).astype(np.float32)
    u, w, h, st = _hip_solve(lr, v, u_up, iters, kappa=kappa)
    u_ref, w_ref, info = occ.solve(lr, v, u_up, u_up, iters, k=KBLUR, kappa=kappa, h=h)
    assert st["iterations"] == iters
    _compare(u, w, u_ref, w_ref)
    p_hip, p_ref = orc.psnr_central(u, truth), orc.psnr_central(u_ref, truth)
    assert abs(p_hip - p_ref) < 0.02
    assert p_hip > orc.psnr_central(u_up, truth) + 0.5            # and it super-resolves
    # h from the device sums against the C restatement's own (float32 data, double sums)
    _, _, info0 = occ.solve(lr, v, u_up, u_up, 0, k=KBLUR, kappa=kappa, h=0.0)
    assert abs(h - info0["h"]) <= 1e-5 * info0["h"]


def _synthetic(ny, nx, nc):
    from superresolution_amd import synthetic
    from superresolution_amd.imresize import upsample_sequence
    blk = synthetic.make_block(ny, nx, nc, 0, nc, device="cuda")
    u_up = upsample_sequence(blk["f_state"], 4).astype(np.float32)
    return blk["f"], blk["v"], u_up


def test_config2_calendar_size_accurate():
    _need_c_oracle()
    f, v, u_up = _synthetic(144, 180, 13)
    iters = 200
    u, w, h, st = _hip_solve(f, v, u_up, iters)
    u_ref, w_ref, _ = occ.solve(f, v, u_up, u_up, iters, k=KBLUR, h=h)
    _compare(u, w, u_ref, w_ref)


def test_config5_4k_frames():
    _need_c_oracle()
    f, v, u_up = _synthetic(540, 960, 3)
    iters = 10
    u, w, h, st = _hip_solve(f, v, u_up, iters)
    u_ref, w_ref, _ = occ.solve(f, v, u_up, u_up, iters, k=KBLUR, h=h)
    _compare(u, w, u_ref, w_ref)


@pytest.mark.parametrize("direction", ["backward", "forward-backward"])
def test_config3_update_sweep_step_sizes_at_1080p(direction):
    """The update sweep rebuilds the column sums of W from its own scatter; the step-size hook
    and mmc_init use a plane.  On a flow with sub-pixel parts, masked border rows and a fast disc
    both must give the same tau, bit for bit."""
    from superresolution_amd import synthetic
    from superresolution_amd.solver import FrameBlockSolver
    ny, nx, nc = 270, 480, 3
    blk = synthetic.make_block(ny, nx, nc, 0, nc, device="cuda")
    v = blk["v"].copy()
    rng = np.random.default_rng(5)
    v += rng.normal(0.0, 0.6, v.shape).astype(np.float32)        # irregular sub-pixel parts
    u_up = np.zeros((4 * ny, 4 * nx, nc), np.float32) + 0.5
    with FrameBlockSolver(ny, nx, 4, nc, flow_direction=direction, h=0.7) as s:
        s.set_lowres(blk["f"]); s.set_blur(KBLUR); s.set_flow(v); s.set_state(u_up, u_up); s.init()
        tu, tw, _, _, _ = s.debug_steps()
        tu2, tw2 = s.debug_steps_update()
    assert np.array_equal(tu, tu2) and np.array_equal(tw, tw2)
    assert np.isfinite(tu).all() and tu.min() > 0 and np.isfinite(tw).all() and tw.min() > 0


def _rank(rank, world, port, ny, nx, L, iters, out_dir):
    import torch
    import torch.distributed as dist
    from superresolution_amd import sharding, synthetic
    from superresolution_amd.imresize import upsample_sequence
    from superresolution_amd.solver import FrameBlockSolver
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        torch.cuda.set_stream(torch.cuda.Stream())
        nc = L * world
        a, b = rank * L, (rank + 1) * L
        blk = synthetic.make_block(ny, nx, nc, a, b, device="cuda")
        u_up = upsample_sequence(blk["f_state"], 4).astype(np.float32)
        s = FrameBlockSolver(ny, nx, 4, nc, a, b, h=0.0, device=0, stream=torch.cuda.current_stream().cuda_stream)
        s.set_lowres(blk["f"]); s.set_blur(KBLUR); s.set_flow(blk["v"]); s.set_state(u_up, u_up)
        s.init()
        s.h = sharding.sharded_h(s, "cuda:0")
        ex = sharding.HaloExchanger(sharding.halo_tensors(s), rank, world)
        st = sharding.solve_sharded(s, ex, iters, 0.0, 0, "cuda:0")
        u, w = s.get_state()
        np.savez(os.path.join(out_dir, "r%d.npz" % rank), u=u, w=w, h=s.h, it=st["iterations"])
        s.close()
    finally:
        dist.destroy_process_group()


def test_config4_two_ranks_at_1080p(tmp_path):
    _need_c_oracle()
    from helpers import spawn_ranks
    ny, nx, L, world, iters = 270, 480, 3, 2, 10
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    spawn_ranks("test_gpu_configs", "_rank", (world, port, ny, nx, L, iters, str(tmp_path)), world)
    f, v, u_up = _synthetic(ny, nx, L * world)
    parts = [np.load(os.path.join(str(tmp_path), "r%d.npz" % r)) for r in range(world)]
    h = float(parts[0]["h"])
    assert float(parts[1]["h"]) == h and int(parts[0]["it"]) == iters
    u_ref, w_ref, info = occ.solve(f, v, u_up, u_up, iters, k=KBLUR, h=h)
    _, _, info0 = occ.solve(f, v, u_up, u_up, 0, k=KBLUR, h=0.0)
    assert abs(h - info0["h"]) <= 1e-5 * info0["h"]
    u = np.concatenate([p["u"] for p in parts], axis=2)
    w = np.concatenate([p["w"] for p in parts], axis=2)
    _compare(u, w, u_ref, w_ref)
