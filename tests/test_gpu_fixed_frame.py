"""GPU parity of the fixed-first-frame model (MultiframeMotionCoupling.m:535-572)
and of the batch chaining built on it (variants/VideoProcessor_supplementary_MMC.m)."""
import numpy as np
import pytest

import mmc_oracle as orc
from helpers import make_case, rel_l2

pytestmark = pytest.mark.gpu
KBLUR = orc.fspecial_gaussian(7, np.sqrt(0.6))


def _setup(f, v, P, u0, w0, direction="backward"):
    from superresolution_amd.solver import FrameBlockSolver
    ny, nx, nc = f.shape
    s = FrameBlockSolver(ny, nx, 4, nc, alpha=P.alpha, kappa=0.25, h=0.0, flow_direction=direction)
    s.set_lowres(f); s.set_blur(KBLUR); s.set_flow(v); s.set_state(P.u_up, P.u_up)
    s.set_fixed_frame(u0, w0)
    s.init()
    return s


@pytest.mark.parametrize("direction", ["backward", "forward"])
def test_fixed_frame_parity(direction):
    f, v, _ = make_case(61, 9, 8, 3)
    rng = np.random.default_rng(0)
    u0 = np.clip(0.5 + 0.2 * rng.normal(size=(36, 32)), 0, 1)
    w0 = 0.05 * rng.normal(size=(36, 32))
    P = orc.build_problem(f, v, k=KBLUR, flowDirection=direction, u0_frame=u0, w0_frame=w0)
    s = _setup(f, v, P, u0, w0, direction)
    tu, tw, sp, s1, s2 = s.debug_steps()
    assert rel_l2(tu, P.T[:P.M]) < 1e-6 and rel_l2(tw, P.T[P.M:2 * P.M]) < 1e-6
    s.set_state(P.u_up, P.u_up)
    x, y, chg = orc.pdhg(P, 120)
    st = s.solve(120)
    u, w = s.get_state()
    u_ref, w_ref = orc.split_primal(P, x)
    assert rel_l2(u, u_ref) < 1e-4 and np.abs(w - w_ref).max() < 1e-4
    assert abs(st["res_dual"] - chg[2] / chg[3]) < 2e-2 * chg[2] / chg[3] + 1e-7
    # run on: the first frame converges to the pinned values
    s.solve(4000)
    u, w = s.get_state()
    assert np.abs(u[:, :, 0] - u0).max() < 5e-3 and np.abs(w[:, :, 0] - w0).max() < 5e-3
    s.close()


def test_fixed_frame_argument_errors():
    from superresolution_amd import MMCError
    from superresolution_amd.solver import FrameBlockSolver
    s = FrameBlockSolver(8, 8, 4, 3)
    with pytest.raises(MMCError, match="Give both fix frames u0 and w0"):          # :190-192
        s.set_fixed_frame(np.zeros((32, 32)), None)
    s.close()
    s = FrameBlockSolver(8, 8, 4, 3, kappa=1.0)
    with pytest.raises(MMCError, match="infimal-convolution"):
        s.set_fixed_frame(np.zeros((32, 32)), np.zeros((32, 32)))
    s.close()


def test_video_batches_chain_on_the_shared_frame():
    from superresolution_amd.video import process_video, batch_ranges
    f, v, _ = make_case(62, 8, 8, 7, flow_scale=1.0)

    def cfg(m):
        m.comp_mode = "fast"
        m.check_every = 200

    u, w = process_video(f, v, 4, k=KBLUR, configure=cfg)
    assert u.shape == (32, 32, 7) and np.isfinite(u).all() and np.isfinite(w).all()
    assert batch_ranges(7, 4) == [(0, 4), (3, 7), (5, 7)]
    # batch 1 alone gives the values frame 3 was pinned to in batch 2; the final u keeps them within the solve tolerance
    from superresolution_amd.mmc import MultiframeMotionCoupling
    m = MultiframeMotionCoupling(f[:, :, 0:4], flowField=v[:, :, 0:3, :], verbose=0)
    m.k = KBLUR
    cfg(m)
    m.init(); m.run()
    u_first = m.result1
    m.close()
    m2 = MultiframeMotionCoupling(f[:, :, 3:7], flowField=v[:, :, 3:6, :], verbose=0,
                                  u0_frame=u_first[:, :, 3], w0_frame=m.result2[:, :, 3])
    m2.k = KBLUR
    cfg(m2)
    m2.init(); m2.run()
    assert np.abs(m2.result1[:, :, 0] - u_first[:, :, 3]).max() < 2e-2          # pinned frame follows batch 1
    assert np.allclose(u[:, :, 0:3], u_first[:, :, 0:3])
    m2.close()
