#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/.

The reference (MATLAB) cannot run in this environment and ships no expected
outputs, so these vectors are produced by the CPU oracle (oracle/mmc_oracle.py)
and pin it -- and through it the HIP path -- against regressions.  PARITY
UNPINNED with respect to the real prost/flexBox binaries.

  synthetic_small.npz   seeded 10x12 LR x 4 frames, all three models, 60 PDHG
                        iterations of the oracle (u, w, h, energies)
  tube3_crop.npz        luma of frames 1-3 of the reference's only bundled scene
                        (data/video_scenes/tube3, rows/cols 172:428 of the
                        600x600 frames): 8-bit ground truth crop, the 64x64 LR
                        frames obtained as helpers/LoadImSequence.m does (bicubic,
                        anti-aliased, clipped), and Lucas-Kanade flows estimated
                        on the LR frames and upscaled x4 like
                        MultiframeMotionCoupling.m:958-959.
  tube3_full.npz        the same for the whole 600x600 frames (150x150 LR): the size
                        testfile.m:22-31 runs at.
Run from the repository root:  python tests/golden/make_golden.py [/root/reference]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mmc_oracle as orc  # noqa: E402
from helpers import make_case  # noqa: E402


def synthetic_small():
    f, v, occ = make_case(1234, 10, 12, 4, flow="smooth", flow_scale=2.0, with_occ=True)
    f = f.astype(np.float32).astype(np.float64)      # stored as float32; solve exactly what is stored
    k = orc.fspecial_gaussian(7, np.sqrt(0.6))
    out = {"f": f.astype(np.float32), "v": v.astype(np.float32), "occ": occ, "k": k}
    for name, kappa in (("ic", 0.25), ("k1", 1.0), ("add", float("nan"))):
        for direction in ("backward", "forward-backward"):
            P = orc.build_problem(f, v, kappa=kappa, flowDirection=direction, k=k, occ=occ)
            x, y, chg = orc.pdhg(P, 60)
            u, w = orc.split_primal(P, x)
            tag = "%s_%s" % (name, "b" if direction == "backward" else "fb")
            out["u_" + tag] = u.astype(np.float32)
            if w is not None:
                out["w_" + tag] = w.astype(np.float32)
            out["h_" + tag] = np.float64(P.h)
            out["e0_" + tag] = np.float64(orc.primal_energy(P, P.x0))
            out["e_" + tag] = np.float64(orc.primal_energy(P, x))
    np.savez_compressed(os.path.join(HERE, "synthetic_small.npz"), **out)


def lucas_kanade(a, b, win=7, iters=3):
    """Dense LK flow d with b(x + d) ~ a(x); returns (dy, dx).  Fixture tooling only."""
    from scipy.ndimage import uniform_filter, map_coordinates, gaussian_filter
    a = gaussian_filter(a, 1.0)
    b = gaussian_filter(b, 1.0)
    ny, nx = a.shape
    yy, xx = np.mgrid[0:ny, 0:nx].astype(np.float64)
    dy = np.zeros_like(a)
    dx = np.zeros_like(a)
    for _ in range(iters):
        bw = map_coordinates(b, [yy + dy, xx + dx], order=1, mode="nearest")
        gy, gx = np.gradient(bw)
        it = bw - a
        Sxx = uniform_filter(gx * gx, win) + 1e-4
        Syy = uniform_filter(gy * gy, win) + 1e-4
        Sxy = uniform_filter(gx * gy, win)
        Sxt = uniform_filter(gx * it, win)
        Syt = uniform_filter(gy * it, win)
        det = Sxx * Syy - Sxy * Sxy
        dx -= (Syy * Sxt - Sxy * Syt) / det
        dy -= (Sxx * Syt - Sxy * Sxt) / det
        dx = np.clip(gaussian_filter(dx, 1.5), -3, 3)
        dy = np.clip(gaussian_filter(dy, 1.5), -3, 3)
    return dy, dx


def tube3_crop(reference_root, sl=slice(172, 428), out_name="tube3_crop.npz"):
    from PIL import Image
    folder = os.path.join(reference_root, "data", "video_scenes", "tube3")
    names = sorted(n for n in os.listdir(folder) if n.lower().endswith(".png"))[:3]
    truth = []
    for n in names:
        rgb = np.asarray(Image.open(os.path.join(folder, n)).convert("RGB"), dtype=np.float64) / 255.0
        y = (16.0 + 65.481 * rgb[:, :, 0] + 128.553 * rgb[:, :, 1] + 24.966 * rgb[:, :, 2]) / 255.0   # rgb2ycbcr luma
        truth.append(y[sl, sl])
    truth = np.stack(truth, axis=2)
    truth8 = np.round(truth * 255.0).astype(np.uint8)
    truth = truth8.astype(np.float64) / 255.0
    # helpers/LoadImSequence.m:45-46, :61-62
    lr = np.clip(np.stack([orc.imresize(truth[:, :, i], 0.25, "bicubic", True) for i in range(3)], axis=2), 0, 1)
    # backward convention: u_t(x) ~ u_{t+1}(x + v_t(x))
    flows = np.zeros(lr.shape[:2] + (2, 2))
    for i in range(2):
        dy, dx = lucas_kanade(lr[:, :, i], lr[:, :, i + 1])
        flows[:, :, i, 0] = dx          # component 1: along columns
        flows[:, :, i, 1] = dy          # component 2: along rows
    np.savez_compressed(os.path.join(HERE, out_name), truth_u8=truth8,
                        lr=lr.astype(np.float32), flow_lr=flows.astype(np.float16), frames=np.array(names))


if __name__ == "__main__":
    synthetic_small()
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    if os.path.isdir(ref):
        tube3_crop(ref)
        tube3_crop(ref, slice(0, 600), "tube3_full.npz")      # testfile.m:22-31 at its real size
    else:
        print("reference checkout not found; tube3_crop.npz not regenerated")
    for n in sorted(os.listdir(HERE)):
        print(n, os.path.getsize(os.path.join(HERE, n)))
