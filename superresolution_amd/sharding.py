"""Frame-block sharding of one sequence over several GPUs (one process each).

The motion-coupling operator only couples frame t with frame t+1
(helpers/constructWarpFF.m:40-58), blur/sampling and the spatial gradient act
inside a frame.  Rank r therefore owns a contiguous block of frames with all
their primal and dual variables, and per PDHG iteration exchanges exactly two
frame-sized messages with each chain neighbour:

  after the primal sweep   xbar of its FIRST frame      -> previous rank
  after the dual sweep     temporal duals of its LAST   -> next rank
                           frame (pair last|next)

Point-to-point over torch.distributed (backend "nccl" = RCCL over xGMI; "gloo"
in the CPU tests).  Each exchange is posted right after the frames it sends are
final and is only waited for before the one frame that needs the ghost, so it
overlaps with the sweep over the interior frames.
"""
import numpy as np


def partition_frames(nc, world):
    """Contiguous blocks [begin, end) for each rank; remainders go to the first ranks."""
    if world <= 0 or nc < world:
        raise ValueError("need at least one frame per rank (nc=%d, world=%d)" % (nc, world))
    base, rem = divmod(nc, world)
    out, b = [], 0
    for r in range(world):
        e = b + base + (1 if r < rem else 0)
        out.append((b, e))
        b = e
    return out


def pairs_needed(begin, end, nc):
    """Global flow pairs [lo, hi) a block touches: its own and the previous block's last."""
    lo = begin - 1 if begin > 0 else 0
    hi = min(end, nc - 1)
    return lo, max(hi, lo)


def state_frames_needed(begin, end, nc):
    """Frames whose start point a block needs: its own plus the next block's first."""
    return begin, min(end + 1, nc)


class DeviceBuffer(object):
    """Zero-copy view of `count` float32 at device address `ptr` for torch.as_tensor."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f4",
                                         "data": (int(ptr), False), "version": 2, "strides": None}


def halo_tensors(solver):
    """Torch views of the frames a FrameBlockSolver exchanges (see include/mmc_hip.h: mmc_halo)."""
    import torch
    hl = solver.halo_info()
    count = hl.frame_bytes // 4
    dev = "cuda:%d" % solver.params.device

    def view(p):
        return None if not p else torch.as_tensor(DeviceBuffer(p, count), device=dev)

    out = {name: [view(getattr(hl, name)[i]) for i in range(2)]
           for name in ("send_prev", "recv_next", "send_next", "recv_prev")}
    # first frame of the iterate (u, w) -> ghost slot of the previous block: needed by the diagnostics only
    sh = solver.state_halo_info()
    out["state_send_prev"] = [view(sh.send_prev[i]) for i in range(2)]
    out["state_recv_next"] = [view(sh.recv_next[i]) for i in range(2)]
    return out


def require_torch_stream(solver):
    """The exchange is ordered against torch's current stream: the RCCL send/recv kernels (or the
    staged copies) wait for work queued there and `wait()` makes that stream wait for them.  The
    sweeps must therefore be queued on the same stream.  A FrameBlockSolver created with
    stream=None, or with the default stream (handle 0), runs on a private stream of the library,
    and the frames would be sent and overwritten while the sweeps are still running."""
    import torch
    lib_stream = solver.params.stream or 0
    cur = torch.cuda.current_stream(solver.params.device).cuda_stream
    if lib_stream == 0 or lib_stream != cur:
        raise RuntimeError("the FrameBlockSolver must be created with stream=torch.cuda.current_stream().cuda_stream of a "
                           "non-default stream, and that stream must be current during solve_sharded "
                           "(library stream %#x, torch current stream %#x)" % (lib_stream, cur))


class HaloExchanger(object):
    """Posts the two per-iteration exchanges of one block as batched isend/irecv.

    With the "nccl" backend (RCCL) the device frames are sent as they are.  A
    backend without device point-to-point (gloo) gets them staged through pinned
    host memory; that mode exists for functional tests of the multi-process path
    on a machine with fewer GPUs than ranks, not for speed.
    """

    def __init__(self, tensors, rank, world, group=None):
        import torch
        import torch.distributed as dist
        self.torch = torch
        self.dist = dist
        self.t = tensors
        self.prev = rank - 1 if rank > 0 else None
        self.next = rank + 1 if rank < world - 1 else None
        self.group = group
        on_device = any(t is not None and t.is_cuda for ts in tensors.values() for t in ts)
        self.on_device = on_device
        self.staged = on_device and dist.get_backend(group) != "nccl"
        self.host = {}
        if self.staged:
            for name, ts in tensors.items():
                self.host[name] = [None if t is None else torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                                   for t in ts]

    def _start(self, send_name, send_to, recv_name, recv_from):
        P2POp, d = self.dist.P2POp, self.dist
        ops, after = [], []
        if send_to is not None:
            for k, t in enumerate(self.t[send_name]):
                if t is None:
                    continue
                if self.staged:
                    self.host[send_name][k].copy_(t, non_blocking=True)
                    t = self.host[send_name][k]
                ops.append(P2POp(d.isend, t, send_to, self.group))
        if recv_from is not None:
            for k, t in enumerate(self.t[recv_name]):
                if t is None:
                    continue
                if self.staged:
                    after.append((t, self.host[recv_name][k]))
                    t = self.host[recv_name][k]
                ops.append(P2POp(d.irecv, t, recv_from, self.group))
        if self.staged and ops:
            self.torch.cuda.current_stream().synchronize()      # host copies of the outgoing frames are complete
        reqs = self.dist.batch_isend_irecv(ops) if ops else []
        return (reqs, after)

    def start_xbar(self):
        """xbar of the first local frame -> previous rank; ghost of the next rank's first frame <- next."""
        return self._start("send_prev", self.prev, "recv_next", self.next)

    def start_state(self):
        """u, w of the first local frame -> previous rank's ghost slot (before the diagnostics passes)."""
        if "state_send_prev" not in self.t:
            return None
        return self._start("state_send_prev", self.prev, "state_recv_next", self.next)

    def start_dual(self):
        """temporal duals of the last local frame -> next rank; ghost of the previous rank's <- previous."""
        return self._start("send_next", self.next, "recv_prev", self.prev)

    @staticmethod
    def wait(pending):
        if not pending:
            return
        reqs, after = pending
        for r in reqs:
            r.wait()
        for dev, host in after:
            dev.copy_(host, non_blocking=True)


def allreduce_sums(values, device=None):
    """Sum a short float64 vector over all ranks; returns a numpy array."""
    import torch
    import torch.distributed as dist
    t = torch.as_tensor(np.asarray(values, dtype=np.float64))
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t)
    return t.cpu().numpy()


def stats_from_diag(d, n_primal, n_dual, tol=0.0):
    """The diagnostics of mmc_stats from the MMC_NUM_DIAG sums of the whole sequence (the formula of
    mmc_stats_from_diag in the C ABI) and whether they pass the absolute + relative residual test."""
    out = {"primal_energy": float(d[0] + d[1]), "dual_energy": float(-d[8] - d[4]), "dual_infeasibility": float(d[5]),
           "norm_Kx": float(np.sqrt(max(d[2], 0.0))), "opt_res_dual": float(np.sqrt(max(d[3], 0.0))),
           "norm_KTy": float(np.sqrt(max(d[6], 0.0))), "opt_res_primal": float(np.sqrt(max(d[7], 0.0)))}
    out["gap"] = out["primal_energy"] - out["dual_energy"]
    passes = tol > 0 and (out["opt_res_primal"] <= tol * np.sqrt(max(n_primal, 0.0)) + tol * out["norm_KTy"] and
                          out["opt_res_dual"] <= tol * np.sqrt(max(n_dual, 0.0)) + tol * out["norm_Kx"])
    return out, bool(passes)


def sharded_h(engine, device=None):
    """h = sum|W u0| / sum|grad u0| over the whole sequence (MultiframeMotionCoupling.m:431-437)."""
    s = allreduce_sums(engine.h_partials(), device)
    if not (s[0] > 0 and s[1] > 0):
        raise ValueError("warp or gradient energy of the start point is zero; pass h explicitly")
    return float(s[0] / s[1])


ADAPT_EVERY = 50      # iterations between two looks at the step balance (mmc_solve's cadence)


def solve_sharded(engine, exchanger, max_iters, tol=0.0, check_every=0, device=None, callback=None, diagnostics=None,
                  adapt_steps=False):
    """The loop of mmc_solve for a block with neighbours.

    `engine` provides L, sweep_primal(t0,t1,reduce), sweep_dual(t0,t1,reduce),
    sums_reset(), sums_read() and sync(); FrameBlockSolver does.  Ghost duals
    start at zero on both sides, matching the zero start of the duals.
    diagnostics: evaluate energies, gap and optimality residuals at every check and stop on the
    residual test (see mmc_stats in include/mmc_hip.h); None = when there is a tolerance or a
    callback, as mmc_solve does.  Needs engine.diag_partials() and engine.problem_size().
    adapt_steps: choose the step balance by residual balancing as mmc_solve does with
    mmc_set_step_adaptation (the step-weighted change sums added up over the ranks, the library's
    rule applied on every rank).  Needs a FrameBlockSolver.
    """
    L = engine.L
    if getattr(exchanger, "on_device", False) and hasattr(engine, "params"):
        require_torch_stream(engine)
    has_prev = exchanger.prev is not None
    has_next = exchanger.next is not None
    if check_every <= 0:
        check_every = max(max_iters, 1)
    stats = {"iterations": 0, "converged": False, "res_primal": 0.0, "res_dual": 0.0}
    if diagnostics is None:
        diagnostics = (tol > 0 or callback is not None) and hasattr(engine, "diag_partials")
    reqs_dual = None
    balance, alpha = None, None
    if adapt_steps:
        import ctypes as C
        balance, alpha = C.c_double(engine.step_balance), C.c_double(0.5)
    for it in range(max_iters):
        want_check = ((it + 1) % check_every == 0) or it == max_iters - 1
        adapt_now = adapt_steps and (it + 1) % ADAPT_EVERY == 0
        check = want_check or adapt_now            # sweeps with reduced sums
        if check:
            engine.sums_reset()
        # primal sweep: the first frame needs the previous block's temporal duals
        if has_prev:
            engine.sweep_primal(1, L, check)
            exchanger.wait(reqs_dual)
            engine.sweep_primal(0, 1, check)
        else:
            exchanger.wait(reqs_dual)
            engine.sweep_primal(0, L, check)
        reqs_x = exchanger.start_xbar()
        # dual sweep: the last frame needs the next block's xbar
        if has_next:
            engine.sweep_dual(0, L - 1, check)
            exchanger.wait(reqs_x)
            engine.sweep_dual(L - 1, L, check)
        else:
            engine.sweep_dual(0, L, check)
            exchanger.wait(reqs_x)
        reqs_dual = exchanger.start_dual()
        stats["iterations"] = it + 1
        if adapt_now:
            ws = allreduce_sums(engine.step_sums_read(), device)
            arr = (C.c_double * 2)(float(ws[0]), float(ws[1]))
            if engine._lib.mmc_step_balance_rule(arr, C.byref(balance), C.byref(alpha)):
                engine.step_balance = balance.value
            stats["step_balance"] = balance.value
        if want_check:
            s = allreduce_sums(engine.sums_read(), device)
            tiny = 1e-30
            stats["res_primal"] = float(np.sqrt(s[0]) / max(np.sqrt(s[1]), tiny))
            stats["res_dual"] = float(np.sqrt(s[2]) / max(np.sqrt(s[3]), tiny))
            done = tol > 0 and stats["res_primal"] < tol and stats["res_dual"] < tol
            if diagnostics:
                # the read-only passes need the ghost duals of this iteration and the neighbour's iterate
                exchanger.wait(reqs_dual)
                reqs_dual = None
                exchanger.wait(exchanger.start_state())
                d = allreduce_sums(np.concatenate([engine.diag_partials(), engine.problem_size()]), device)
                more, done = stats_from_diag(d[:-2], d[-2], d[-1], tol)
                stats.update(more)
            stop = callback(it + 1, stats) if callback is not None else False
            if stop:
                break
            if done:
                stats["converged"] = True
                break
    exchanger.wait(reqs_dual)
    engine.sync()
    return stats
