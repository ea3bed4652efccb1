"""Frames above about 2 Mpx switch the block order to sub-ranges of tiles that are each walked
through all frames (288x580 low-res -> 1152x2320 high-res: 2610 tiles, 327 per XCD class, two
sub-ranges of 164 and 163 / 157 tiles).  Every (tile, frame) must still be processed exactly once:
checked against the C restatement of the solve (oracle/mmc_oracle_c.c), which handles this size in
seconds, through the product's synthetic scene."""
import numpy as np
import pytest

import mmc_oracle_c as occ
from helpers import rel_l2

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("direction", ["backward", "forward-backward"])
def test_large_frame_block_order(direction):
    if not occ.available():
        pytest.skip("C oracle not built")
    from superresolution_amd import synthetic
    from superresolution_amd.imresize import upsample_sequence
    from superresolution_amd.solver import FrameBlockSolver
    ny, nx, nc, iters = 288, 580, 3, 12
    blk = synthetic.make_block(ny, nx, nc, 0, nc, device="cpu")
    k = synthetic.gaussian_kernel()
    u_up = upsample_sequence(blk["f_state"], 4).astype(np.float32)
    s = FrameBlockSolver(ny, nx, 4, nc, flow_direction=direction)
    s.set_lowres(blk["f"]); s.set_blur(k); s.set_flow(blk["v"]); s.set_state(u_up, u_up); s.init()
    st = s.solve(iters, 0.0, 0)
    u, w = s.get_state()
    h = s.h
    s.close()
    u_ref, w_ref, info = occ.solve(blk["f"], blk["v"], u_up, u_up, iters, k=k, flowDirection=direction, h=h)
    assert st["iterations"] == iters
    assert rel_l2(u, u_ref) < 1e-5 and rel_l2(w, w_ref) < 1e-3
    # a tile skipped or done twice would leave a block of pixels one update behind / ahead
    assert np.abs(u - u_ref).max() < 1e-4


@pytest.mark.parametrize("factor,ny,nx,kappa", [(3, 361, 640, 0.25), (2, 540, 960, float("nan")), (3, 360, 640, 1.0)])
def test_full_hd_at_other_zoom_factors(factor, ny, nx, kappa):
    """1080p-sized frames at zoom 3 and 2 ('stride' sampling, 7x7 blur): 1083 rows are not a multiple of 4,
    which takes the sweeps that do not work in row quads; 1080 rows at zoom 2 or 3 take the row-quad
    sweeps in their form for any zoom factor.  Against the C restatement at equal iteration count."""
    if not occ.available():
        pytest.skip("C oracle not built")
    from superresolution_amd import synthetic
    from superresolution_amd.imresize import upsample_sequence
    from superresolution_amd.solver import FrameBlockSolver
    nc, iters = 3, 8
    blk = synthetic.make_block(ny, nx, nc, 0, nc, factor=factor, device="cpu")
    k = synthetic.gaussian_kernel()
    u_up = upsample_sequence(blk["f_state"], factor).astype(np.float32)
    s = FrameBlockSolver(ny, nx, factor, nc, kappa=kappa, sampling="stride")
    s.set_lowres(blk["f"]); s.set_blur(k); s.set_flow(blk["v"]); s.set_state(u_up, u_up); s.init()
    st = s.solve(iters, 0.0, 0)
    u, w = s.get_state()
    h = s.h
    s.close()
    u_ref, w_ref, info = occ.solve(blk["f"], blk["v"], u_up, u_up, iters, k=k, factor=factor, kappa=kappa, sampling="stride", h=h)
    assert st["iterations"] == iters
    assert rel_l2(u, u_ref) < 1e-5 and np.abs(u - u_ref).max() < 1e-4
    if w is not None:
        assert rel_l2(w, w_ref) < 1e-3
