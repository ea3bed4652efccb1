#!/usr/bin/env python3
"""Throughput of the MMC super-resolution PDHG solve on MI355X.

  python bench.py --gpus N --steps K --warmup W [--total-frames T] [--lr ny nx]

A step is one PDHG iteration (primal sweep + dual sweep) over this rank's frame
block.  Workload: synthetic 1080p sequence, HR frames of 1080x1920 (LR 270x480,
zoom 4), infimal-convolution model (kappa 0.25), alpha 0.01, backward flow, 7x7
Gaussian blur + 4x4 block mean.  Default (weak scaling): 30 frames per GPU, the
sequence has 30*N frames.  --total-frames T fixes the sequence (strong scaling):
240 for the long 1080p sequence, `--lr 540 960 --total-frames 120` for the 4K
one; frames are block-sharded, boundary frames exchanged every sweep.  Inputs
are resident in HBM before the timed region.  Rank 0 prints one JSON line.

With --gpus N > 1 and no launcher environment, this process starts the N ranks
itself (python -m torch.distributed.run, one process per GPU, 127.0.0.1) before
it touches any GPU, relays their output and exits with their status.
"""
import argparse
import json
import os
import socket
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
PMC_SUMMARIES = ("r3_pmc_hbm_bytes.json", "r2_pmc_hbm_bytes.json", "r1_pmc_hbm_bytes.json")


def log(msg):
    """Progress on stderr (unbuffered): a stall under a profiler names its phase."""
    sys.stderr.write("[bench %.1fs] %s\n" % (time.perf_counter() - T_START, msg))
    sys.stderr.flush()


T_START = time.perf_counter()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--frames-per-gpu", type=int, default=30)
    ap.add_argument("--total-frames", type=int, default=0,
                    help="fixed sequence length, block-sharded over the GPUs (strong scaling); 0: frames-per-gpu * gpus")
    ap.add_argument("--lr", type=int, nargs=2, default=[270, 480], metavar=("ny", "nx"))
    ap.add_argument("--kappa", type=float, default=0.25)
    ap.add_argument("--flow", default="rigid", choices=["rigid", "smooth", "outliers"],
                    help="flow field of the headline run (see synthetic.perturb_flows)")
    ap.add_argument("--direction", default="backward", choices=["forward", "backward", "forward-backward"],
                    help="flowDirection of the coupling operator (the same synthetic flow field is used for all three: "
                         "other directions time the other pair layouts, their images mean nothing)")
    ap.add_argument("--extra-flows", default="smooth,outliers",
                    help="comma list of further flow kinds timed after the headline run at N=1 (extra keys only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-inputs", action="store_true",
                    help="render the synthetic inputs with torch on the CPU, so that this process launches no "
                         "kernels but the library's (for hardware-counter runs)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="functional rehearsal: ranks share the visible GPUs (use with --backend gloo)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-group-leg", action="store_true",
                    help="N > 1: skip the extra leg that times the same sequence through mmc_group_* (one process, one "
                         "host thread driving all N devices: what the MEX gateway calls)")
    ap.add_argument("--group-blocks", type=int, default=0,
                    help="N = 1: also time the sequence cut into this many frame blocks on the one GPU through mmc_group_*")
    return ap.parse_args()


def self_launch(args):
    """Start the ranks of an N-GPU run from a plain `python bench.py --gpus N`.  Nothing in this
    process has touched a GPU; the children are fresh processes."""
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    import subprocess       # the only child process this program starts, and it starts it before any GPU call
    return subprocess.call(cmd, env=env)


def host_topology():
    """Sockets and cores per socket from /proc/cpuinfo (read in-process: no child program is started
    once the GPU runtime is live)."""
    try:
        phys, cores = set(), {}
        cur = None
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.startswith("physical id"):
                    cur = int(ln.split(":")[1])
                    phys.add(cur)
                elif ln.startswith("cpu cores") and cur is not None:
                    cores[cur] = int(ln.split(":")[1])
        if phys and cores:
            return {"cores_per_socket": max(cores.values()), "sockets": len(phys)}
    except (OSError, ValueError):
        pass
    return None


def c_oracle_ready():
    """Locate (or build, with make) the C restatement.  Called before the first GPU call: building
    starts a child process."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    try:
        import mmc_oracle_c as oc
        return oc.available()
    except Exception:
        return False


def cpu_baseline(args, budget_s, have_c, keep=None):
    """The oracle's PDHG iteration timed on the host cores with the C/OpenMP restatement: the whole
    single-GPU workload for 50 iterations with stored sparse rows for W' and D*B (the flexBox-style
    path), a 4-frame sample in the mode that stores no operator, and the reference's own
    quick-start size (600x600x3) for its 800 iterations; with a budget under 10 s only the 4-frame
    samples; if the C library is not there, a small crop with the scipy.sparse restatement.
    keep: a dict that receives the inputs, the start point and the iterate of the whole-workload run
    (for the parity check of the HIP path against this very run).  Starts no child process."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from superresolution_amd import synthetic
    k = synthetic.gaussian_kernel()
    host_cpus = os.cpu_count()
    if have_c:
        import mmc_oracle_c as oc
        ny, nx = args.lr
        lscpu = host_topology()
        stored = "C/OpenMP restatement with the warp adjoint and D*B stored as sparse rows, the way init_u_flexBox hands " \
                 "them to flexBox (oracle/mmc_oracle_c.c); loop time only, operator set-up excluded"
        full = budget_s >= 10.0 and args.frames_per_gpu >= 8
        if full:
            # the whole single-GPU workload for 50 iterations (the loop takes about 10 s on 16 threads)
            nc, iters = args.frames_per_gpu, 50
            blk = synthetic.make_block(ny, nx, nc, 0, nc, device="cpu")
            res = oc.time_solve(blk["f"], blk["v"], k, kappa=args.kappa, iters=iters, return_state=keep is not None)
            if keep is not None:
                keep.update(f=blk["f"], v=blk["v"], iters=iters, u_up=res.pop("u_up"), u=res.pop("u"), w=res.pop("w"),
                            h=res.pop("h"), primal_energy=res.pop("primal_energy"), gap=res.pop("gap"))
            sample = "the whole workload (%dx%dx%d HR) for %d iterations; %s" % (4 * ny, 4 * nx, nc, iters, stored)
        else:
            nc = 4
            blk = synthetic.make_block(ny, nx, nc, 0, nc, device="cpu")
            res = oc.time_solve(blk["f"], blk["v"], k, kappa=args.kappa, budget_s=0.5 * budget_s)
            sample = "first %d frames of the workload at full size (%dx%dx%d HR), %d iterations; %s" % (
                nc, 4 * ny, 4 * nx, nc, res["iterations"], stored)
        out = {"value": res["mpx_iter_per_s"], "unit": "HR Mpixel*PDHG-iters/s", "cores": res["threads"],
               "kind": "port", "host_cpus": host_cpus, "omp_max_threads": res["threads"], "lscpu": lscpu, "sample": sample,
               "wall_s_per_sr_frame_at_7500_iters_extrapolated": 7500 * 16 * ny * nx / res["mpx_iter_per_s"] / 1e6}
        # the same loop without stored operators, on a 4-frame sample
        blk = synthetic.make_block(ny, nx, 4, 0, 4, device="cpu")
        r2 = oc.time_solve(blk["f"], blk["v"], k, kappa=args.kappa, budget_s=0.25 * budget_s, matrix_free=True)
        out["matrix_free"] = {"value": r2["mpx_iter_per_s"], "iterations": r2["iterations"],
                              "sample": "first 4 frames at full size; taps recomputed, warp adjoint scattered with atomic adds, "
                                        "D*B through the composite kernel"}
        # the reference's quick-start size in full: 3 frames of 600x600, 800 iterations (the flexBox cap of 'fast')
        blk = synthetic.make_block(150, 150, 3, 0, 3, device="cpu")
        r1 = oc.time_solve(blk["f"], blk["v"], k, kappa=args.kappa, iters=800 if full else None, budget_s=0.25 * budget_s, max_iters=800)
        out["quickstart_600x600x3"] = {"value": r1["mpx_iter_per_s"], "iterations": r1["iterations"],
                                       "seconds_per_800_iterations": 800 * 1.08 / r1["mpx_iter_per_s"]}
        return out
    import mmc_oracle as orc
    ny, nx, nc = 64, 96, 3
    blk = synthetic.make_block(ny, nx, nc, 0, nc, device="cpu")
    P = orc.build_problem(blk["f"], blk["v"], kappa=args.kappa, k=k)
    x, y, _ = orc.pdhg(P, 2)
    t0 = time.perf_counter()
    iters = 0
    while time.perf_counter() - t0 < budget_s and iters < 400:
        x, y, _ = orc.pdhg(P, 10, x=x, y=y)
        iters += 10
    dt = time.perf_counter() - t0
    return {"value": P.M * iters / dt / 1e6, "unit": "HR Mpixel*PDHG-iters/s", "cores": 1, "kind": "port",
            "host_cpus": host_cpus,
            "sample": "%dx%dx%d HR crop of the workload, %d iterations, scipy.sparse restatement (oracle/mmc_oracle.py)"
                      % (4 * ny, 4 * nx, nc, iters)}


def parity_check(s, keep, model):
    """Re-runs the solver `s` on the CPU leg's inputs and compares the iterates (float32 both)."""
    s.set_lowres(keep["f"])
    s.set_flow(keep["v"])
    s.set_state(keep["u_up"], keep["u_up"])
    s.init()
    st = s.solve(keep["iters"], 0.0, 0)
    u, w = s.get_state()
    e = s.diagnostics()
    du = u.astype(np.float64) - keep["u"]
    res = {"iters": int(st["iterations"]), "frames": int(u.shape[2]), "hr": [int(u.shape[0]), int(u.shape[1])],
           "rel_l2_u": float(np.linalg.norm(du) / np.linalg.norm(keep["u"])), "max_abs_u": float(np.abs(du).max()),
           "h_hip": s.h, "h_cpu": float(keep["h"]),
           "primal_energy_hip": e["primal_energy"], "primal_energy_cpu": float(keep["primal_energy"]),
           "gap_hip": e["gap"], "gap_cpu": float(keep["gap"]), "tolerance": "rel_l2_u <= 1e-4 (float32, same iteration count)"}
    if w is not None and keep["w"] is not None:
        res["max_abs_w"] = float(np.abs(w.astype(np.float64) - keep["w"]).max())
    res["ok"] = bool(res["rel_l2_u"] <= 1e-4)
    return res


def group_leg(args, ny, nx, nc, devices, in_dev):
    """The same sequence through mmc_group_* (one process, one host thread, one frame block per entry
    of `devices`; MultiframeMotionCoupling.m:977-986 is one call from one MATLAB session): warm-up,
    then args.steps iterations timed between two device synchronisations."""
    from superresolution_amd import synthetic
    from superresolution_amd.solver import SequenceGroupSolver
    blk = synthetic.make_block(ny, nx, nc, 0, nc, factor=4, device=in_dev, flow_kind=args.flow)
    with SequenceGroupSolver(ny, nx, 4, nc, nblocks=len(devices), devices=devices, alpha=0.01, kappa=args.kappa, h=0.0,
                             flow_direction=args.direction, sampling="average", device=devices[0]) as g:
        g.set_lowres(blk["f"])
        g.set_blur(synthetic.gaussian_kernel())
        g.set_flow(blk["v"])
        del blk
        g.set_state()               # bicubic start point, computed on the devices
        g.init()
        g.solve(args.warmup, 0.0, 0)
        t0 = time.perf_counter()
        st = g.solve(args.steps, 0.0, 0)      # returns device-synchronised
        wall = time.perf_counter() - t0
        assert st["iterations"] == args.steps
    M = 16 * ny * nx * nc
    return {"driver": "mmc_group_* (one process, one host thread)", "devices": list(devices), "frames_total": nc,
            "ms_per_step": wall / args.steps * 1e3, "value": M * args.steps / wall / 1e6}


def measured_traffic(kernel, model, L, Ny, Nx):
    """HBM bytes per launch of `kernel` from the committed hardware-counter summary
    (separate rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE passes over this same workload;
    FETCH_SIZE doubled as gfx950 needs).  None when no summary is for this workload."""
    for fname in PMC_SUMMARIES:
        path = os.path.join(ROOT, "profiles", fname)
        try:
            with open(path) as fh:
                d = json.load(fh)
        except (OSError, ValueError):
            continue
        wl = d.get("workload", {})
        if wl.get("frames") != L or wl.get("hr") != [Ny, Nx] or wl.get("model") != model:
            continue
        for name, v in d.get("kernels", {}).items():
            if name.startswith(kernel + "<") and ", false," in name:
                return v["hbm_bytes_per_launch"], "profiles/" + fname
    return None, None


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))

    # ranks that share device buffers (RCCL) need the dmabuf IPC mode of this driver stack; set before the runtime loads
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # whatever may start a child process (make for the C restatement) happens before the GPU runtime is loaded
    have_c = c_oracle_ready() if (world == 1 and not args.no_cpu_baseline) else False

    import torch
    from superresolution_amd import synthetic
    from superresolution_amd.imresize import upsample_sequence
    from superresolution_amd.solver import FrameBlockSolver
    from superresolution_amd import sharding

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; no HIP device is visible")
    if args.share_gpu:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    dist = None
    ranks_seen = 1
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device(dev))
        else:
            dist.init_process_group(backend=args.backend)
        # the library queues on torch's current stream so that torch orders kernels and transfers;
        # a stream of our own, because the default stream's handle is 0 (= "library, make your own")
        torch.cuda.set_stream(torch.cuda.Stream())
        one = torch.ones(1, dtype=torch.int64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(one)
        ranks_seen = int(one.item())

    ny, nx = args.lr
    factor = 4
    Ny, Nx = ny * factor, nx * factor
    if args.total_frames > 0:
        nc = args.total_frames
        scaling = "strong"
    else:
        nc = args.frames_per_gpu * world
        scaling = "weak"
    begin, end = sharding.partition_frames(nc, world)[rank]
    L = end - begin
    model = "ic" if not (np.isnan(args.kappa) or args.kappa == 1) else "single"
    # Hardware-counter collection (rocprofv3 --pmc) over torch's own rendering kernels, which include
    # MIOpen's convolution search, stalls before the first library call; under a profiler the
    # inputs are therefore rendered on the host.
    profiled = [k for k in os.environ if k.startswith(("ROCPROF", "ROCP_"))]
    if profiled and not args.host_inputs:
        log("profiler environment (%s): rendering inputs on the host" % profiled[0])
        args.host_inputs = True
    in_dev = "cpu" if args.host_inputs else dev

    log("rank %d: rendering frames [%d, %d) of %d on %s" % (rank, begin, end, nc, in_dev))
    blk = synthetic.make_block(ny, nx, nc, begin, end, factor=factor, device=in_dev, flow_kind=args.flow)
    u_up = upsample_sequence(blk["f_state"], factor).astype(np.float32)
    stream = torch.cuda.current_stream().cuda_stream if world > 1 else None
    s = FrameBlockSolver(ny, nx, factor, nc, begin, end, alpha=0.01, kappa=args.kappa, h=0.0,
                         flow_direction=args.direction, sampling="average", device=local_rank, stream=stream)
    s.set_lowres(blk["f"])
    s.set_blur(synthetic.gaussian_kernel())
    s.set_flow(blk["v"])
    s.set_state(u_up, u_up)
    del blk
    log("rank %d: mmc_init" % rank)
    s.init()

    from superresolution_amd import _capi
    hip_built, hip_runtime = _capi.hip_versions()
    M_local = Ny * Nx * L
    M_total = Ny * Nx * nc
    prof = None
    extra = {}
    if world == 1:
        log("warm-up, %d iterations" % args.warmup)
        s.solve(args.warmup, 0.0, 0)
        s.set_profiling(True)
        torch.cuda.synchronize()
        log("timed region, %d iterations" % args.steps)
        t0 = time.perf_counter()
        st = s.solve(args.steps, 0.0, 0)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        prof = s.get_profile()
        log("timed region done: %.3f ms per iteration" % (wall / args.steps * 1e3))
        # After an idle or lightly loaded stretch the GPU runs the first 10 to 15 iterations (about 20 ms)
        # several per cent slower, whatever the iterate (tools/transient_check.py): a short timed region
        # after a short warm-up sits partly inside that stretch.  The same sweeps continued for at
        # least 200 more iterations give the sustained rate (extra key; the headline is the region above).
        n_sus = max(args.steps, 200)
        s.set_profiling(False)
        s.set_profiling(True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        s.solve(n_sus, 0.0, 0)
        torch.cuda.synchronize()
        wall_sus = time.perf_counter() - t1
        prof_sus = s.get_profile()
        s.set_profiling(False)
        # the same sweeps on irregular flow fields (extra keys; the headline stays the line above)
        for kind in [k for k in args.extra_flows.split(",") if k and k != args.flow]:
            log("flow kind '%s'" % kind)
            blk2 = synthetic.make_block(ny, nx, nc, begin, end, factor=factor, device=in_dev, flow_kind=kind)
            s.set_flow(blk2["v"])
            del blk2
            s.set_state(u_up, u_up)
            s.init()
            s.solve(max(args.warmup // 2, 2), 0.0, 0)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            s.solve(args.steps, 0.0, 0)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t1) / args.steps * 1e3
            extra[kind] = {"ms_per_step": ms, "value": M_total / ms / 1e3, "vs_headline": (wall / args.steps * 1e3) / ms}
    else:
        s.h = sharding.sharded_h(s, dev if args.backend == "nccl" else None)
        ex = sharding.HaloExchanger(sharding.halo_tensors(s), rank, world)
        red_dev = dev if args.backend == "nccl" else None
        sharding.solve_sharded(s, ex, args.warmup, 0.0, 0, red_dev)
        s.set_profiling(True)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        st = sharding.solve_sharded(s, ex, args.steps, 0.0, 0, red_dev)
        torch.cuda.synchronize()
        dist.barrier()
        wall = time.perf_counter() - t0
        t = torch.tensor([wall], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
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

    # Extra leg (never the headline): the one-process driver on the same devices, from rank 0 while
    # the other ranks wait; at N = 1 only on request, as several blocks on the one GPU.
    group = None
    group_devices = None
    if world > 1 and not args.no_group_leg:
        group_devices = [r % torch.cuda.device_count() for r in range(world)] if args.share_gpu else list(range(world))
    elif world == 1 and args.group_blocks > 1:
        group_devices = [local_rank] * args.group_blocks
    if group_devices is not None:
        if dist is not None:
            dist.barrier()
        if rank == 0:
            log("group leg on devices %s" % group_devices)
            try:
                group = group_leg(args, ny, nx, nc, group_devices, in_dev)
                group["vs_headline"] = group["value"] / (M_total * args.steps / wall / 1e6)
            except Exception as e:      # the headline stands on its own
                group = {"error": repr(e)}
            torch.cuda.set_device(local_rank)
        if dist is not None:
            dist.barrier()

    if rank == 0:
        value = M_total * args.steps / wall / 1e6
        out = {
            "metric": "HR Mpixel*PDHG-iters/s per node",
            "value": value,
            "unit": "HR Mpixel*PDHG-iters/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "synthetic sequence, %d HR frames of %dx%d (LR %dx%d, zoom 4), %d on rank 0, "
                                   "%s model kappa=%g, alpha=0.01, %s flow (%s), gaussian(7,sqrt(0.6)) blur + 4x4 mean"
                                   % (nc, Ny, Nx, ny, nx, L, "infimal-convolution" if model == "ic" else "single-primal",
                                      args.kappa, args.direction, args.flow),
                       "frames_total": nc, "hr_pixels_total": M_total, "parallelism": "frame-block x%d" % world,
                       "ranks_seen": ranks_seen, "backend": args.backend if world > 1 else None,
                       "plane_skew_bytes": s.plane_skew,
                       "hip_built_against": hip_built, "hip_runtime": hip_runtime},
            "wall_s_per_sr_frame_at_7500_iters": wall / args.steps * 7500 / nc,
            "iteration_roofline": {"bytes_per_px_iter": BYTES_ITER[model],
                                   "achieved_GBps_per_gpu": BYTES_ITER[model] * M_total * args.steps / wall / world / 1e9,
                                   "frac_of_8TBps": BYTES_ITER[model] * M_total * args.steps / wall / world / HBM_PEAK},
        }
        if extra:
            out["other_flows"] = extra
        if group is not None:
            out["group_driver"] = group
        if prof is not None and prof["iterations"] > 0:
            # the dual sweep is one fused kernel (k_dual4); its second slot only times the event gap
            kernels = {"k_primal4": (prof["primal_ms"], BYTES_PRIMAL[model]),
                       "k_dual4": (prof["dual_reg_ms"] + prof["dual_data_ms"], BYTES_DUAL[model])}
            name = max(kernels, key=lambda n: kernels[n][0])
            ms, bpp = kernels[name]
            achieved = bpp * M_local / (ms * 1e-3) / 1e9
            traffic, source = measured_traffic(name, model, L, Ny, Nx)
            out["roofline"] = {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK / 1e9,
                               "unit": "GB/s", "frac": achieved / (HBM_PEAK / 1e9),
                               "traffic": traffic,
                               "traffic_source": None if source is None else source + " (rocprofv3 --pmc, bytes per launch)",
                               "algorithmic_bytes_per_launch": bpp * M_local, "avg_launch_ms": ms,
                               "launches_timed": prof["iterations"],
                               "all_kernels_ms": {k: v[0] for k, v in kernels.items()}}
        if world == 1 and prof is not None:
            ms_d = prof_sus["dual_reg_ms"] + prof_sus["dual_data_ms"]
            out["sustained"] = {"steps": n_sus, "after_steps": args.warmup + args.steps, "ms_per_step": wall_sus / n_sus * 1e3,
                                "value": M_total * n_sus / wall_sus / 1e6,
                                "all_kernels_ms": {"k_primal4": prof_sus["primal_ms"], "k_dual4": ms_d},
                                "k_dual4_roofline_frac": BYTES_DUAL[model] * M_local / (ms_d * 1e-3) / HBM_PEAK}
        if not args.no_cpu_baseline and world == 1:
            log("cpu baseline")
            keep = {}
            try:
                out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds, have_c, keep)
            except Exception as e:      # the GPU number stands on its own
                out["cpu_baseline"] = {"value": None, "error": repr(e)}
            if "u" in keep and args.direction == "backward":
                # The HIP path on the inputs and from the start point of the CPU run above, for the same
                # number of iterations: the iterate at the size that was timed, against the oracle's.
                log("parity check against the cpu run (%d iterations)" % keep["iters"])
                try:
                    out["parity_check"] = parity_check(s, keep, model)
                except Exception as e:
                    out["parity_check"] = {"error": repr(e)}
        print(json.dumps(out))
        sys.stdout.flush()
    s.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
