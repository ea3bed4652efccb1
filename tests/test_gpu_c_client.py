"""The C ABI driven from a plain C program (tests/capi_client/capi_client.c, built with gcc against
include/mmc_hip.h): the same problem as the ctypes path, checked against the oracle."""
import os
import spawn

import numpy as np
import pytest

import mmc_oracle as orc
from helpers import make_case, rel_l2

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
CLIENT_DIR = os.path.join(HERE, "capi_client")
CLIENT = os.path.join(CLIENT_DIR, "capi_client")


def _client():
    if not os.path.exists(CLIENT):
        assert spawn.run(["make", "-C", CLIENT_DIR]).returncode == 0
    return CLIENT


@pytest.mark.parametrize("kappa,direction", [(0.25, 1), (float("nan"), 2)])
def test_c_program_reproduces_the_oracle(tmp_path, kappa, direction):
    ny, nx, nc, iters = 10, 12, 3, 60
    f, v, occ = make_case(91, ny, nx, nc, flow="smooth", flow_scale=2.0, with_occ=True)
    k = orc.fspecial_gaussian(7, np.sqrt(0.6))
    name = ["forward", "backward", "forward-backward"][direction]
    P = orc.build_problem(f, v, kappa=kappa, flowDirection=name, k=k, occ=occ)
    x, y, chg = orc.pdhg(P, iters)
    u_ref, w_ref = orc.split_primal(P, x)
    Ny, Nx = 4 * ny, 4 * nx

    def f32(a):
        return np.asarray(a, dtype=np.float32).ravel(order="F").tobytes()

    inp, out = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(inp, "wb") as fh:
        fh.write(np.array([ny, nx, nc, 4, 7, 7, iters, direction, 1], dtype=np.int32).tobytes())
        fh.write(np.array([P.alpha, kappa], dtype=np.float32).tobytes())
        fh.write(f32(f)); fh.write(f32(k)); fh.write(f32(v)); fh.write(f32(P.u_up)); fh.write(f32(P.u_up))
        fh.write(np.asarray(occ, dtype=np.uint8).ravel(order="F").tobytes())
    res = spawn.run([_client(), str(inp), str(out)], timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    raw = open(out, "rb").read()
    M = Ny * Nx * nc
    u = np.frombuffer(raw, dtype=np.float32, count=M).reshape((Ny, Nx, nc), order="F")
    tail = np.frombuffer(raw, dtype=np.float64, count=4, offset=8 * M)
    assert rel_l2(u, u_ref) < 1e-4
    assert abs(tail[0] - P.h) <= 1e-5 * P.h
    assert abs(tail[3] - orc.primal_energy(P, x)) <= 1e-3 * abs(orc.primal_energy(P, x))


def test_c_program_reports_library_errors(tmp_path):
    inp, out = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(inp, "wb") as fh:       # even blur kernel: helpers/RepConvMtx.m:6-8
        fh.write(np.array([4, 4, 2, 4, 6, 6, 5, 1, 0], dtype=np.int32).tobytes())
        fh.write(np.array([0.01, 0.25], dtype=np.float32).tobytes())
        fh.write(np.zeros(4 * 4 * 2 + 36 + 16 * 16 * 2 + 2 * 16 * 16 * 2, dtype=np.float32).tobytes())
    res = spawn.run([_client(), str(inp), str(out)], timeout=120)
    assert res.returncode == 2 and "even dimensions" in res.stderr
