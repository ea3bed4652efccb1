#!/usr/bin/env python3
"""Throughput of the MMC super-resolution PDHG solve on MI355X.

  python bench.py --gpus N --steps K --warmup W

A step is one PDHG iteration (primal sweep + dual sweep) over this rank's frame
block.  Workload: synthetic 1080p sequence, 30 HR frames of 1080x1920 per GPU
(LR 270x480, zoom 4), infimal-convolution model (kappa 0.25), alpha 0.01,
backward flow, 7x7 Gaussian blur + 4x4 block mean; with N GPUs the sequence has
30*N frames, block-sharded, boundary frames exchanged every sweep (weak
scaling).  Inputs are resident in HBM before the timed region.  Rank 0 prints
one JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Algorithmic HBM bytes per HR pixel and PDHG iteration, fp32, counting every
# array once per sweep (stencil / warp neighbours assumed on-chip):
#   primal sweep : read u,w (2) g1,g2 (6) flow (2) p (1/16); write u,w,ub,wb (4)   = 14.0625 floats
#   dual sweep   : read ub,wb (2) flow (2) g1,g2 (6) p,f (2/16); write g1,g2 (6) p (1/16) = 16.1875 floats
BYTES_PRIMAL = {"ic": 14.0625 * 4, "single": 8.0625 * 4}
BYTES_DUAL = {"ic": 16.1875 * 4, "single": 9.1875 * 4}
BYTES_ITER = {"ic": 121.0, "single": 69.0}
HBM_PEAK = 8.0e12


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--frames-per-gpu", type=int, default=30)
    ap.add_argument("--lr", type=int, nargs=2, default=[270, 480], metavar=("ny", "nx"))
    ap.add_argument("--kappa", type=float, default=0.25)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="functional rehearsal: ranks share the visible GPUs (use with --backend gloo)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    return ap.parse_args()


def cpu_baseline(args, budget_s):
    """The oracle's PDHG iteration timed on the host cores: a bounded sample of the same
    workload (first 6 frames, 544x960 HR) with the C/OpenMP restatement, or, if that cannot
    be built, a smaller crop with the scipy.sparse restatement on one core."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from superresolution_amd import synthetic
    try:
        import mmc_oracle_c as oc
        have_c = oc.available()
    except Exception:
        have_c = False
    ny, nx, nc = (136, 240, 6) if have_c else (64, 96, 3)
    blk = synthetic.make_block(ny, nx, nc, 0, nc, device="cpu")
    k = synthetic.gaussian_kernel()
    if have_c:
        res = oc.time_solve(blk["f"], blk["v"], k, kappa=args.kappa, budget_s=budget_s)
        return {"value": res["mpx_iter_per_s"], "unit": "HR Mpixel*PDHG-iters/s", "cores": res["threads"],
                "kind": "port",
                "sample": "%dx%dx%d HR crop of the workload, %d iterations, C/OpenMP matrix-free restatement (oracle/mmc_oracle_c.c)"
                          % (4 * ny, 4 * nx, nc, res["iterations"])}
    import mmc_oracle as orc
    P = orc.build_problem(blk["f"], blk["v"], kappa=args.kappa, k=k)
    x, y, _ = orc.pdhg(P, 2)
    t0 = time.perf_counter()
    iters = 0
    while time.perf_counter() - t0 < budget_s and iters < 400:
        x, y, _ = orc.pdhg(P, 10, x=x, y=y)
        iters += 10
    dt = time.perf_counter() - t0
    return {"value": P.M * iters / dt / 1e6, "unit": "HR Mpixel*PDHG-iters/s", "cores": 1, "kind": "port",
            "sample": "%dx%dx%d HR crop of the workload, %d iterations, scipy.sparse restatement (oracle/mmc_oracle.py)"
                      % (4 * ny, 4 * nx, nc, iters)}


def measured_traffic(kernel, model, L, Ny, Nx):
    """HBM bytes per launch of `kernel` from the committed hardware-counter summary
    (separate rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE passes over this same workload;
    FETCH_SIZE doubled as gfx950 needs).  None when the summary is for another workload."""
    path = os.path.join(ROOT, "profiles", "r1_pmc_hbm_bytes.json")
    try:
        with open(path) as fh:
            d = json.load(fh)
    except (OSError, ValueError):
        return None
    wl = d.get("workload", {})
    if wl.get("frames") != L or wl.get("hr") != [Ny, Nx] or wl.get("model") != model:
        return None
    for name, v in d.get("kernels", {}).items():
        if name.startswith(kernel + "<") and ", false," in name:
            return v["hbm_bytes_per_launch"]
    return None


def main():
    args = parse()
    import torch
    from superresolution_amd import synthetic
    from superresolution_amd.imresize import upsample_sequence
    from superresolution_amd.solver import FrameBlockSolver
    from superresolution_amd import sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..."
                             % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; no HIP device is visible")
    if args.share_gpu:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device(dev))
        else:
            dist.init_process_group(backend=args.backend)
        # the library queues on torch's current stream so that torch orders kernels and transfers;
        # a stream of our own, because the default stream's handle is 0 (= "library, make your own")
        torch.cuda.set_stream(torch.cuda.Stream())

    ny, nx = args.lr
    L = args.frames_per_gpu
    nc = L * world
    begin, end = rank * L, (rank + 1) * L
    factor = 4
    Ny, Nx = ny * factor, nx * factor
    model = "ic" if not (np.isnan(args.kappa) or args.kappa == 1) else "single"

    blk = synthetic.make_block(ny, nx, nc, begin, end, factor=factor, device=dev)
    u_up = upsample_sequence(blk["f_state"], factor).astype(np.float32)
    stream = torch.cuda.current_stream().cuda_stream if world > 1 else None
    s = FrameBlockSolver(ny, nx, factor, nc, begin, end, alpha=0.01, kappa=args.kappa, h=0.0,
                         flow_direction="backward", sampling="average", device=local_rank, stream=stream)
    s.set_lowres(blk["f"])
    s.set_blur(synthetic.gaussian_kernel())
    s.set_flow(blk["v"])
    s.set_state(u_up, u_up)
    del blk, u_up
    s.init()

    M_local = Ny * Nx * L
    prof = None
    if world == 1:
        s.solve(args.warmup, 0.0, 0)
        s.set_profiling(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st = s.solve(args.steps, 0.0, 0)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        prof = s.get_profile()
    else:
        s.h = sharding.sharded_h(s, dev)
        ex = sharding.HaloExchanger(sharding.halo_tensors(s), rank, world)
        sharding.solve_sharded(s, ex, args.warmup, 0.0, 0, dev)
        s.set_profiling(True)
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st = sharding.solve_sharded(s, ex, args.steps, 0.0, 0, dev)
        torch.cuda.synchronize()
        dist.barrier()
        wall = time.perf_counter() - t0
        t = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
        # this rank's sweeps, HIP events on the library's stream; a sweep is two launches when the
        # block waits for a ghost frame in between
        sp = s.get_sweep_profile()
        cp = 2 if (s.has_prev and L > 1) else 1
        cd = 2 if (s.has_next and L > 1) else 1
        it_p, it_d = sp["primal_calls"] / cp, sp["dual_calls"] / cd
        if it_p >= 1 and it_d >= 1:
            prof = {"primal_ms": sp["primal_ms"] / it_p, "dual_reg_ms": sp["dual_ms"] / it_d, "dual_data_ms": 0.0,
                    "iterations": int(min(it_p, it_d))}
    assert st["iterations"] == args.steps

    if rank == 0:
        M_total = M_local * world
        value = M_total * args.steps / wall / 1e6
        out = {
            "metric": "HR Mpixel*PDHG-iters/s per node",
            "value": value,
            "unit": "HR Mpixel*PDHG-iters/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "synthetic 1080p-style sequence, %d HR frames of %dx%d per GPU (LR %dx%d, zoom 4), "
                                   "%s model kappa=%g, alpha=0.01, backward flow, gaussian(7,sqrt(0.6)) blur + 4x4 mean"
                                   % (L, Ny, Nx, ny, nx, "infimal-convolution" if model == "ic" else "single-primal", args.kappa),
                       "frames_total": nc, "hr_pixels_total": M_total, "parallelism": "frame-block x%d" % world},
            "wall_s_per_sr_frame_at_7500_iters": wall / args.steps * 7500 / nc,
            "iteration_roofline": {"bytes_per_px_iter": BYTES_ITER[model],
                                   "achieved_GBps_per_gpu": BYTES_ITER[model] * M_total * args.steps / wall / world / 1e9,
                                   "frac_of_8TBps": BYTES_ITER[model] * M_total * args.steps / wall / world / HBM_PEAK},
        }
        if prof is not None and prof["iterations"] > 0:
            # the dual sweep is one fused kernel (k_dual4); its second slot only times the event gap
            kernels = {"k_primal4": (prof["primal_ms"], BYTES_PRIMAL[model]),
                       "k_dual4": (prof["dual_reg_ms"] + prof["dual_data_ms"], BYTES_DUAL[model])}
            name = max(kernels, key=lambda n: kernels[n][0])
            ms, bpp = kernels[name]
            achieved = bpp * M_local / (ms * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK / 1e9,
                               "unit": "GB/s", "frac": achieved / (HBM_PEAK / 1e9),
                               "traffic": measured_traffic(name, model, L, Ny, Nx),
                               "traffic_source": "profiles/r1_pmc_hbm_bytes.json (rocprofv3 --pmc, bytes per launch)",
                               "algorithmic_bytes_per_launch": bpp * M_local, "avg_launch_ms": ms,
                               "launches_timed": prof["iterations"],
                               "all_kernels_ms": {k: v[0] for k, v in kernels.items()}}
        if not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
            except Exception as e:      # the GPU number stands on its own
                out["cpu_baseline"] = {"value": None, "error": repr(e)}
        print(json.dumps(out))
    s.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
