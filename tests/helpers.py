"""Shared synthetic inputs for the tests (seeded, small)."""
import numpy as np


def smooth_field(rng, shape, scale, cells=4):
    """Low-frequency random field: bilinear blow-up of a coarse random grid."""
    Ny, Nx = shape
    coarse = rng.normal(0.0, scale, (cells + 1, cells + 1))
    yy = np.linspace(0, cells, Ny)
    xx = np.linspace(0, cells, Nx)
    y0 = np.clip(np.floor(yy).astype(int), 0, cells - 1)
    x0 = np.clip(np.floor(xx).astype(int), 0, cells - 1)
    a = (yy - y0)[:, None]
    b = (xx - x0)[None, :]
    return ((1 - a) * (1 - b) * coarse[np.ix_(y0, x0)] + a * (1 - b) * coarse[np.ix_(y0 + 1, x0)]
            + (1 - a) * b * coarse[np.ix_(y0, x0 + 1)] + a * b * coarse[np.ix_(y0 + 1, x0 + 1)])


def make_case(seed, ny, nx, nc, factor=4, flow="smooth", flow_scale=2.0, with_occ=False):
    """Low-res frames in [0,1], flows (Ny,Nx,nc-1,2) and optional occlusion maps."""
    rng = np.random.default_rng(seed)
    Ny, Nx = factor * ny, factor * nx
    f = np.clip(0.5 + 0.25 * np.stack([smooth_field(rng, (ny, nx), 1.0, 3) for _ in range(nc)], axis=2)
                + 0.05 * rng.normal(size=(ny, nx, nc)), 0.0, 1.0)
    v = np.zeros((Ny, Nx, max(nc - 1, 0), 2))
    for i in range(nc - 1):
        for comp in range(2):
            if flow == "smooth":
                v[:, :, i, comp] = smooth_field(rng, (Ny, Nx), flow_scale)
            elif flow == "random":
                v[:, :, i, comp] = rng.normal(0.0, flow_scale, (Ny, Nx))
            elif flow == "integer":
                v[:, :, i, comp] = np.round(smooth_field(rng, (Ny, Nx), flow_scale))
            elif flow == "zero":
                pass
            else:
                raise ValueError(flow)
    occ = None
    if with_occ and nc > 1:
        occ = (rng.random((Ny, Nx, nc - 1)) < 0.1).astype(np.uint8)
    return f, v.astype(np.float32).astype(np.float64), occ


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def make_sf_case(seed, ny, nx, nf, factor=4, flow="smooth", flow_scale=1.5, with_occ=False):
    """Single-frame problem: low-res frames, flows from the central frame to every frame
    (Ny,Nx,nf,2; the central slice is filled with garbage on purpose, it must be ignored) and
    optional occlusion maps."""
    rng = np.random.default_rng(seed)
    Ny, Nx = factor * ny, factor * nx
    f = np.clip(0.5 + 0.25 * np.stack([smooth_field(rng, (ny, nx), 1.0, 3) for _ in range(nf)], axis=2)
                + 0.05 * rng.normal(size=(ny, nx, nf)), 0.0, 1.0)
    v = np.zeros((Ny, Nx, nf, 2))
    for i in range(nf):
        for comp in range(2):
            v[:, :, i, comp] = smooth_field(rng, (Ny, Nx), flow_scale) if flow == "smooth" else rng.normal(0.0, flow_scale, (Ny, Nx))
    occ = (rng.random((Ny, Nx, nf)) < 0.1).astype(np.uint8) if with_occ else None
    return f, v.astype(np.float32).astype(np.float64), occ


def spawn_ranks(module, function, args, world, timeout=280):
    """torch.multiprocessing.spawn(module.function, args, nprocs=world) from a fresh process (tests/mp_launch.py,
    started through tests/spawn.py): the ranks are not forked from the test process."""
    import json
    import os
    import sys
    import spawn
    here = os.path.dirname(os.path.abspath(__file__))
    r = spawn.run([sys.executable, os.path.join(here, "mp_launch.py"), module, function, str(world), json.dumps(list(args))],
                  timeout=timeout)
    assert r.returncode == 0, (r.stdout[-1500:] + r.stderr[-3000:])
