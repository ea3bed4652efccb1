"""World-size-2 (and 3) CPU test of the frame-block sharding: the product's
halo-exchange loop (superresolution_amd.sharding.solve_sharded / HaloExchanger)
drives an oracle-backed block engine over gloo and must reproduce the
single-process oracle solve exactly.  The engine poisons every frame it does not
own, so a missing or mis-ordered exchange cannot go unnoticed."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import mmc_oracle as orc
from helpers import make_case

KBLUR = orc.fspecial_gaussian(7, np.sqrt(0.6))
POISON = 1.0e3


class OracleBlockEngine(object):
    """Frame block [a,b) of the oracle's PDHG, exposing the sweep interface of FrameBlockSolver."""

    def __init__(self, P, a, b):
        self.P, self.a, self.b, self.L = P, a, b, b - a
        Ny, Nx, nc, ny, nx = P.dims
        self.N, self.M, self.n, self.nc = Ny * Nx, P.M, ny * nx, nc
        self.ic = len(P.primal) == 2
        self.x = np.full(P.K.shape[1], POISON)
        self.xb = np.full(P.K.shape[1], POISON)
        self.y = np.full(P.K.shape[0], POISON)
        own = self._primal_idx(a, b)
        self.x[own] = P.x0[own]
        self.xb[own] = P.x0[own]
        self.y[self._dual_idx(a, b)] = 0.0
        if a > 0:                                   # ghost duals start at zero like all duals
            self.y[self._dual_idx(a - 1, a)] = 0.0
        if b < nc:                                  # ghost of the start point (needed for h only)
            gi = self._primal_idx(b, b + 1)
            self.xb[gi] = P.x0[gi]
        self.sums = np.zeros(4)

    def _primal_idx(self, t0, t1):
        r = np.arange(t0 * self.N, t1 * self.N)
        return np.concatenate([r, self.M + r]) if self.ic else r

    def _dual_idx(self, t0, t1):
        parts = [np.arange(t0 * self.n, t1 * self.n)]
        off = self.n * self.nc
        r = np.arange(t0 * self.N, t1 * self.N)
        for name, size, kind, par in self.P.dual_blocks[1:]:
            for c in range(size // self.M):
                parts.append(off + c * self.M + r)
            off += size
        return np.concatenate(parts)

    def h_partials(self):
        P = self.P
        u = np.where(np.arange(self.M) // self.N < self.b + 1, self.xb[:self.M], 0.0)
        rows_w = np.arange(self.a * self.N, min(self.b, self.nc - 1) * self.N)
        wu = np.abs(P.W[rows_w] @ np.where(np.abs(self.xb[:self.M]) < POISON, self.xb[:self.M], 0.0)).sum() if len(rows_w) else 0.0
        r = np.arange(self.a * self.N, self.b * self.N)
        gu = np.abs(P.G[np.concatenate([r, self.M + r])] @ np.where(np.abs(self.xb[:self.M]) < POISON, self.xb[:self.M], 0.0)).sum()
        return wu, gu

    def sweep_primal(self, t0, t1, reduce=False):
        if t1 <= t0:
            return
        P = self.P
        idx = self._primal_idx(self.a + t0, self.a + t1)
        xo = self.x[idx]
        xn = xo - P.T[idx] * (P.KT[idx] @ self.y)
        nu = (t1 - t0) * self.N
        xn[:nu] = np.clip(xn[:nu], 0.0, 1.0)
        self.x[idx] = xn
        self.xb[idx] = 2 * xn - xo
        if reduce:
            self.sums[0] += ((xn - xo) ** 2).sum()
            self.sums[1] += (xn ** 2).sum()

    def sweep_dual(self, t0, t1, reduce=False):
        if t1 <= t0:
            return
        P = self.P
        fa, fb = self.a + t0, self.a + t1
        idx = self._dual_idx(fa, fb)
        yo = self.y[idx]
        yt = yo + P.S[idx] * (P.K[idx] @ self.xb)
        # prox per block on the frame slice (same maps as orc.prox_Fstar)
        np_, nN = (fb - fa) * self.n, (fb - fa) * self.N
        out = np.empty_like(yt)
        bound, lin = P.dual_blocks[0][3]
        out[:np_] = np.clip(yt[:np_] - P.S[idx][:np_] * lin[fa * self.n:fb * self.n], -bound, bound)
        off = np_
        for name, size, kind, par in P.dual_blocks[1:]:
            comps = size // self.M
            seg = yt[off:off + comps * nN].reshape(comps, nN)
            if kind == "ball":
                r = par[0]
                nrm = np.sqrt((seg ** 2).sum(axis=0))
                seg = seg * np.where(nrm > r, r / np.where(nrm > 0, nrm, 1.0), 1.0)[None, :]
            elif kind == "box":
                seg = np.clip(seg, -par[0], par[0])
            out[off:off + comps * nN] = seg.ravel()
            off += comps * nN
        self.y[idx] = out
        if reduce:
            self.sums[2] += ((out - yo) ** 2).sum()
            self.sums[3] += (out ** 2).sum()

    def sums_reset(self):
        self.sums[:] = 0

    def sums_read(self):
        return self.sums.copy()

    def sync(self):
        pass

    def halo_tensors(self):
        """Views of the frames exchanged, same roles as mmc_halo."""
        N, M = self.N, self.M
        tv = lambda arr, s: torch.from_numpy(arr[s:s + N])
        off1 = self.n * self.nc
        sizes = [sz for _, sz, _, _ in self.P.dual_blocks]
        off2 = off1 + sizes[1]
        t = {"send_prev": [None, None], "recv_next": [None, None], "send_next": [None, None], "recv_prev": [None, None]}
        if self.a > 0:
            t["send_prev"] = [tv(self.xb, self.a * N), tv(self.xb, M + self.a * N) if self.ic else None]
            t["recv_prev"] = [tv(self.y, off1 + (self.a - 1) * N), tv(self.y, off2 + (self.a - 1) * N) if self.ic else None]
        if self.b < self.nc:
            t["recv_next"] = [tv(self.xb, self.b * N), tv(self.xb, M + self.b * N) if self.ic else None]
            t["send_next"] = [tv(self.y, off1 + (self.b - 1) * N), tv(self.y, off2 + (self.b - 1) * N) if self.ic else None]
        return t


def _worker(rank, world, port, nc, kappa, direction, iters, out_dir):
    import torch.distributed as dist
    from superresolution_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        f, v, occ = make_case(31, 6, 7, nc, flow="smooth", flow_scale=1.5, with_occ=True)
        P = orc.build_problem(f, v, kappa=kappa, flowDirection=direction, k=KBLUR, occ=occ)
        a, b = sharding.partition_frames(nc, world)[rank]
        eng = OracleBlockEngine(P, a, b)
        h = sharding.sharded_h(eng)
        ex = sharding.HaloExchanger(eng.halo_tensors(), rank, world)
        seen = []
        st = sharding.solve_sharded(eng, ex, iters, 0.0, 10, callback=lambda it, s: seen.append((it, s["res_primal"])) and False)
        np.savez(os.path.join(out_dir, "r%d.npz" % rank), a=a, b=b, x=eng.x, y=eng.y, h=h,
                 res=np.array([st["res_primal"], st["res_dual"]]), checks=np.array([it for it, _ in seen]))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,nc,kappa,direction", [
    (2, 4, 0.25, "backward"), (2, 5, 0.25, "forward-backward"), (2, 4, float("nan"), "forward"),
    (3, 5, 1.0, "backward"), (2, 2, 0.25, "backward"),
])
def test_sharded_solve_equals_single_process(tmp_path, world, nc, kappa, direction):
    iters = 30
    mp.spawn(_worker, args=(world, _free_port(), nc, kappa, direction, iters, str(tmp_path)), nprocs=world, join=True)
    f, v, occ = make_case(31, 6, 7, nc, flow="smooth", flow_scale=1.5, with_occ=True)
    P = orc.build_problem(f, v, kappa=kappa, flowDirection=direction, k=KBLUR, occ=occ)
    x_ref, y_ref, chg = orc.pdhg(P, iters)
    N, M = P.N, P.M
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), "r%d.npz" % r))
        a, b = int(d["a"]), int(d["b"])
        assert abs(float(d["h"]) - P.h) < 1e-12 * P.h
        assert np.abs(d["x"][a * N:b * N] - x_ref[a * N:b * N]).max() < 1e-12
        if len(P.primal) == 2:
            assert np.abs(d["x"][M + a * N:M + b * N] - x_ref[M + a * N:M + b * N]).max() < 1e-12
        assert abs(d["res"][0] - chg[0] / chg[1]) < 1e-9 and abs(d["res"][1] - chg[2] / chg[3]) < 1e-9
        assert list(d["checks"]) == [10, 20, 30]
