"""Random sequences cut into random numbers of frame blocks (all on this GPU) through the single-process group
driver: the result must equal the single-block solve bit for bit, whatever the cut (one-frame blocks included),
the model, the flow direction, the masks and the outliers.  This exercises everything of the multi-GPU path but
the transport.  usage: python tools/fuzz_group.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np

import mmc_oracle as orc
from helpers import make_case
from superresolution_amd.imresize import upsample_sequence
from superresolution_amd.solver import FrameBlockSolver, SequenceGroupSolver

import ctypes
import faulthandler

if os.environ.get("FUZZ_SYSTEM_HIP", "0") != "0":
    # bring in the system's HIP runtime before anything imports torch, which otherwise supplies its own copy
    # under the same SONAME to every library of the process (torch then sees no GPU; this tool does not need it)
    ctypes.CDLL("/opt/rocm/lib/libamdhip64.so.7", mode=ctypes.RTLD_GLOBAL)

faulthandler.enable()
verbose = os.environ.get("FUZZ_VERBOSE", "0") != "0"      # print every case before it runs
trace = os.path.join(ROOT, "tests", "capi_client", "libcrashtrace.so")
if os.path.exists(trace):                                 # native call stack of a crash inside a shared library
    ctypes.CDLL(trace).crashtrace_install()
last = os.path.join(ROOT, "gpurun_out", "fuzz_group_last_case.txt")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for n in range(cases):
    factor = int(rng.choice([4, 4, 2, 3]))
    ny, nx = int(rng.integers(6, 60)), int(rng.integers(6, 80))
    nc = int(rng.integers(2, 10))
    nblocks = int(rng.integers(2, nc + 1))
    if os.environ.get("FUZZ_ONE_BLOCK", "0") != "0":     # the group calls without neighbours (no events, no frame copies)
        nblocks = 1
    sampling = "average" if factor == 4 and rng.random() < 0.6 else "stride"
    k = orc.fspecial_gaussian(int(rng.choice([3, 5, 7])), 0.8) if rng.random() < 0.8 else None
    kappa = float(rng.choice([0.25, 1.0, float("nan"), 0.6]))
    direction = str(rng.choice(["backward", "forward", "forward-backward"]))
    flow = str(rng.choice(["smooth", "random", "integer"]))
    scale = float(rng.choice([0.7, 2.5, 8.0]))
    with_occ = bool(rng.random() < 0.4)
    f, v, oc = make_case(int(rng.integers(1 << 30)), ny, nx, nc, factor=factor, flow=flow, flow_scale=scale, with_occ=with_occ)
    if rng.random() < 0.4:
        m = rng.random(v.shape[:3]) < 2e-3
        v[m] += rng.choice([-150.0, 90.0, 200.0])
    v = v.astype(np.float32).astype(np.float64)
    device_start = bool(rng.random() < 0.5)          # start point from the device-side bicubic estimate, h from the library
    u_up = upsample_sequence(f, factor).astype(np.float32)
    iters = int(rng.integers(5, 30))
    check = int(rng.choice([0, 7]))
    desc = "kappa=%g %s ny=%d nx=%d nc=%d blocks=%d f=%d %s k=%s occ=%d flow=%s/%g device_start=%d iters=%d" % (
        kappa, direction, ny, nx, nc, nblocks, factor, sampling, None if k is None else k.shape[0], with_occ, flow, scale,
        device_start, iters)
    if verbose:
        print("case %d: %s" % (n, desc))
        sys.stdout.flush()
    if os.path.isdir(os.path.dirname(last)):
        with open(last, "w") as fh:                  # names the case if the process dies
            fh.write("case %d: %s\n" % (n, desc))
    with FrameBlockSolver(ny, nx, factor, nc, kappa=kappa, h=0.0, flow_direction=direction, sampling=sampling) as s:
        s.set_lowres(f); s.set_blur(k); s.set_flow(v, oc)
        if device_start:
            s.set_state()
        else:
            s.set_state(u_up, u_up)
        s.init()
        h1 = s.h
        st1 = s.solve(iters, 0.0, check)
        u1, w1 = s.get_state()
    with SequenceGroupSolver(ny, nx, factor, nc, nblocks=nblocks, kappa=kappa, h=0.0, flow_direction=direction, sampling=sampling) as g:
        g.set_lowres(f); g.set_blur(k); g.set_flow(v, oc)
        if device_start:
            g.set_state()
        else:
            g.set_state(u_up, u_up)
        g.init()
        hg = g.h                                     # summed over the blocks in another order: equal to rounding
        g.h = h1
        stg = g.solve(iters, 0.0, check)
        ug, wg = g.get_state()
    ok = np.array_equal(u1, ug) and (w1 is None or np.array_equal(w1, wg)) and abs(h1 - hg) <= 1e-6 * abs(h1)
    ok = ok and abs(st1["res_primal"] - stg["res_primal"]) <= 1e-5 * abs(st1["res_primal"]) + 1e-12
    if not ok:
        bad += 1
        print("MISS max|du| %.3e  h %.7g / %.7g  res %.6e / %.6e  %s" % (float(np.abs(u1 - ug).max()), h1, hg,
                                                                       st1["res_primal"], stg["res_primal"], desc))
        sys.stdout.flush()
print("cases %d, misses %d" % (cases, bad))
sys.exit(1 if bad else 0)
