"""mmc_h_partials_static -- the entry point that allocates nothing -- against mmc_h_partials
of a handle fed the same frames and flows, through the C ABI.  Two calls in a row on the same
GPU, the second with a smaller sequence: the static workspace keeps the first call's partial
sums, and the second result must not see them.  The two entry points add the same float32
workgroup sums, one with double atomics in arrival order and one in a fixed order, so they agree
to the rounding of double additions (1e-12 relative is the bar written here); the static one is
bitwise reproducible, which is checked as well.  Both are also held against the sums of the explicit
sparse restatement (oracle/mmc_oracle.py), and the contract itself is pinned: free device memory
is the same before and after a call, the call can be recorded into a HIP graph (stream capture
rejects allocations) and replayed, and the caller's current device is left alone."""
import ctypes as C

import numpy as np
import pytest

from helpers import make_case

pytestmark = pytest.mark.gpu


def _handle_sums(f, v, occ, direction):
    import mmc_oracle as orc
    from superresolution_amd.solver import FrameBlockSolver
    ny, nx, nc = f.shape
    u_up = np.stack([orc.imresize(f[:, :, i], 4) for i in range(nc)], axis=2).astype(np.float32)
    with FrameBlockSolver(ny, nx, 4, nc, flow_direction=direction) as s:
        s.set_flow(v, occ)
        s.set_state(u_up, u_up)
        return u_up, s.h_partials()


def _static_sums(u_up, v, occ, direction, stream=None):
    import torch
    from superresolution_amd import _capi
    lib = _capi.load_library()
    Ny, Nx, nc = u_up.shape
    vv = np.asfortranarray(v, dtype=np.float32).copy(order="F")
    if occ is not None:                       # what mmc_set_flow does with an occlusion map
        vv[:, :, :, 0][np.asarray(occ) != 0] = np.nan
    d_u = torch.from_numpy(np.ravel(np.asfortranarray(u_up, dtype=np.float32), order="F").copy()).cuda()
    d_v = torch.from_numpy(np.ravel(vv, order="F").copy()).cuda()
    d_out = torch.full((2,), -1.0, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    rc = lib.mmc_h_partials_static(d_u.data_ptr(), d_v.data_ptr(), Ny, Nx, nc,
                                   {"forward": 0, "backward": 1, "forward-backward": 2}[direction],
                                   0, stream, d_out.data_ptr())
    assert rc == 0, lib.mmc_last_error(None).decode()
    torch.cuda.synchronize()
    return tuple(d_out.cpu().numpy().tolist())


@pytest.mark.parametrize("direction", ["backward", "forward-backward"])
def test_static_entry_point_matches_handle_twice(direction):
    big = make_case(5, 24, 40, 4, flow="smooth", flow_scale=3.0, with_occ=True)
    small = make_case(6, 8, 12, 2, flow="smooth", flow_scale=1.5, with_occ=False)
    for f, v, occ in (big, small):            # second call: smaller input, same GPU, same workspace
        u_up, ref = _handle_sums(f, v, occ, direction)
        got = _static_sums(u_up, v, occ, direction)
        again = _static_sums(u_up, v, occ, direction)
        assert ref[0] > 0 and ref[1] > 0
        for a, b in zip(got, ref):
            assert abs(a - b) <= 1e-12 * abs(b), (got, ref)
        assert got == again                   # fixed summation order
        # the same two sums from the explicit sparse matrices (MultiframeMotionCoupling.m:431-437), float64
        import mmc_oracle as orc
        P = orc.build_problem(f, v, occ=occ, flowDirection=direction)
        uu = P.u_up.ravel(order="F")
        for a, b in zip(got, (np.abs(P.W @ uu).sum(), np.abs(P.G @ uu).sum())):
            assert abs(a - b) <= 2e-5 * abs(b), (got, b)


def test_static_entry_point_allocates_nothing_and_can_be_captured():
    """Free device memory is unchanged by a call, the call records into a HIP graph (capture fails on
    any allocation or synchronisation inside it) and the replayed graph gives the same two sums."""
    import torch
    from superresolution_amd import _capi
    lib = _capi.load_library()
    f, v, occ = make_case(9, 16, 20, 3, flow="smooth", flow_scale=2.0, with_occ=False)
    u_up, ref = _handle_sums(f, v, occ, "backward")
    first = _static_sums(u_up, v, occ, "backward")         # pays the one-off symbol lookup on this device
    Ny, Nx, nc = u_up.shape
    d_u = torch.from_numpy(np.ravel(np.asfortranarray(u_up, dtype=np.float32), order="F").copy()).cuda()
    d_v = torch.from_numpy(np.ravel(np.asfortranarray(v, dtype=np.float32), order="F").copy()).cuda()
    d_out = torch.full((2,), -1.0, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    assert lib.mmc_h_partials_static(d_u.data_ptr(), d_v.data_ptr(), Ny, Nx, nc, 1, 0, None, d_out.data_ptr()) == 0
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0
    assert tuple(d_out.cpu().numpy().tolist()) == first
    # under stream capture
    d_out.fill_(-1.0)
    s = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=s, capture_error_mode="global"):
        rc = lib.mmc_h_partials_static(d_u.data_ptr(), d_v.data_ptr(), Ny, Nx, nc, 1, 0,
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream), d_out.data_ptr())
    assert rc == 0, lib.mmc_last_error(None).decode()
    torch.cuda.synchronize()
    assert tuple(d_out.cpu().numpy().tolist()) == (-1.0, -1.0)      # recorded, not run
    graph.replay()
    torch.cuda.synchronize()
    assert tuple(d_out.cpu().numpy().tolist()) == first
    for a, b in zip(first, ref):
        assert abs(a - b) <= 1e-12 * abs(b)


def test_static_entry_point_rejects_what_does_not_fit():
    from superresolution_amd import _capi
    lib = _capi.load_library()
    # 2160x3840 frames need 32400 workgroups each: 17 of them exceed the workspace
    rc = lib.mmc_h_partials_static(C.c_void_p(256), C.c_void_p(256), 2160, 3840, 17, 1, 0, None, C.c_void_p(256))
    assert rc == -1 and "static workspace" in lib.mmc_last_error(None).decode()
