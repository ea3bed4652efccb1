"""ctypes wrapper of oracle/mmc_oracle_c.c (C/OpenMP restatement; test
infrastructure and CPU baseline only -- see the header of the C file)."""
import ctypes as C
import os
import subprocess
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libmmc_oracle.so")
_DIRS = {"forward": 0, "backward": 1, "forward-backward": 2}


class _Params(C.Structure):
    _fields_ = [("ny", C.c_int), ("nx", C.c_int), ("factor", C.c_int), ("nc", C.c_int),
                ("alpha", C.c_float), ("kappa", C.c_float), ("h", C.c_float),
                ("flow_direction", C.c_int), ("sampling", C.c_int), ("kh", C.c_int), ("kw", C.c_int),
                ("matrix_free", C.c_int)]


def build():
    subprocess.run(["make", "-C", HERE], check=True, capture_output=True)


def available():
    if not os.path.exists(LIB):
        try:
            build()
        except Exception:
            return False
    return os.path.exists(LIB)


def _lib():
    if not available():
        raise RuntimeError("oracle/_build/libmmc_oracle.so is not built (make -C oracle)")
    lib = C.CDLL(LIB)
    fp, dp, up = C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_uint8)
    lib.mmco_solve.restype = C.c_int
    lib.mmco_solve.argtypes = [C.POINTER(_Params), fp, fp, up, fp, fp, fp, C.c_int, C.c_int, dp]
    return lib


def solve(f_lr, v, u0, w0, iters, k=None, factor=4, alpha=0.01, kappa=0.25, flowDirection="backward",
          sampling="average", occ=None, h=0.0, threads=0, matrix_free=False):
    """Runs `iters` PDHG iterations from (u0, w0); returns u, w, info.  matrix_free: no stored
    operators (taps recomputed, W' scattered with atomic adds, D*B as composite-kernel loops)."""
    lib = _lib()
    f32 = lambda a: np.asfortranarray(np.asarray(a), dtype=np.float32)
    f_lr = f32(f_lr)
    ny, nx, nc = f_lr.shape
    Ny, Nx = ny * factor, nx * factor
    p = _Params(ny, nx, factor, nc, alpha, kappa, h, _DIRS[flowDirection], 0 if sampling == "average" else 1, 0, 0,
                1 if matrix_free else 0)
    kk = None
    if k is not None and np.size(k) > 0:
        kk = f32(np.atleast_2d(k))
        p.kh, p.kw = kk.shape
    vv = f32(np.asarray(v).reshape((Ny, Nx, max(nc - 1, 0), 2), order="F")) if nc > 1 else np.zeros(1, np.float32)
    oo = None if occ is None else np.asfortranarray(np.asarray(occ) != 0, dtype=np.uint8)
    u = f32(u0).copy(order="F")
    has_w = not (np.isnan(kappa) or kappa == 1)
    w = f32(w0).copy(order="F") if has_w else None
    out = (C.c_double * 24)()
    fp, up = C.POINTER(C.c_float), C.POINTER(C.c_uint8)
    ptr = lambda a: None if a is None else a.ctypes.data_as(fp)
    rc = lib.mmco_solve(C.byref(p), ptr(f_lr), ptr(vv), None if oo is None else oo.ctypes.data_as(up), ptr(kk),
                        ptr(u), ptr(w), int(iters), int(threads), out)
    if rc != 0:
        raise RuntimeError("mmco_solve: unsupported parameters")
    d = np.array(out[8:17])
    info = {"sums": np.array(out[0:4]), "h": out[4], "loop_seconds": out[5], "threads": int(out[6]),
            # optimality measures at the returned point (layout of MMC_NUM_DIAG in include/mmc_hip.h)
            "diag": d, "primal_energy": d[0] + d[1], "dual_energy": -d[8] - d[4], "gap": d[0] + d[1] + d[8] + d[4],
            "dual_infeasibility": d[5], "opt_res_primal": float(np.sqrt(d[7])), "opt_res_dual": float(np.sqrt(d[3])),
            "norm_KTy": float(np.sqrt(d[6])), "norm_Kx": float(np.sqrt(d[2]))}
    return u, w, info


def time_solve(f_lr, v, k, kappa=0.25, budget_s=15.0, threads=0, max_iters=2000, matrix_free=False, iters=None,
               return_state=False):
    """Throughput of the C restatement: iterations sized to roughly fill `budget_s`.  return_state: also
    hand back the start point and the iterate reached ("u_up", "u", "w"), for checking another
    implementation against the very run that was timed."""
    import sys
    sys.path.insert(0, HERE)
    import mmc_oracle as orc
    f_lr = np.asarray(f_lr, dtype=np.float64)
    ny, nx, nc = f_lr.shape
    u_up = np.stack([orc.imresize(f_lr[:, :, i], 4) for i in range(nc)], axis=2)
    M = 16 * ny * nx * nc
    if iters is None:
        _, _, probe = solve(f_lr, v, u_up, u_up, 3, k=k, kappa=kappa, threads=threads, matrix_free=matrix_free)
        per_it = max(probe["loop_seconds"] / 3, 1e-6)
        iters = int(min(max(budget_s / per_it, 5), max_iters))
    t0 = time.perf_counter()
    u, w, info = solve(f_lr, v, u_up, u_up, iters, k=k, kappa=kappa, threads=threads, matrix_free=matrix_free)
    out = {"mpx_iter_per_s": M * iters / info["loop_seconds"] / 1e6, "iterations": iters,
           "threads": info["threads"], "wall_s": time.perf_counter() - t0}
    if return_state:
        out.update(u_up=u_up, u=u, w=w, h=info["h"], primal_energy=info["primal_energy"], gap=info["gap"])
    return out
