"""CPU checks of the drop-in boundary: libmmc_hip.so loads, exports every symbol
that include/mmc_hip.h declares, and rejects bad parameters the way
MultiframeMotionCoupling.init does -- without any compute call (no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from superresolution_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mmc_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mmc_[a-z_A-Z0-9]+)\s*\(", text)) - {"mmc_progress_cb"})


def test_library_exports_every_declared_symbol():
    lib = _capi.load_library()
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libmmc_hip.so does not export %s" % n
    assert sorted(_capi.SIGNATURES) == names, "ctypes table and header disagree"
    assert lib.mmc_abi_version() == _capi.MMC_ABI_VERSION


def test_header_is_plain_c(tmp_path):
    """The boundary must be consumable from C (MEX gateways are C or C++)."""
    src = tmp_path / "t.c"
    src.write_text('#include "mmc_hip.h"\nint main(void){ mmc_params p; p.size = sizeof(p); return (int)p.size == 0; }\n')
    import subprocess
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src),
                    "-o", str(tmp_path / "t.o")], check=True)


def _params(**kw):
    p = _capi.mmc_params()
    p.size = C.sizeof(_capi.mmc_params)
    p.version = _capi.MMC_ABI_VERSION
    p.ny, p.nx, p.factor, p.nc_global, p.frame_begin, p.frame_end = 8, 8, 4, 3, 0, 3
    p.alpha, p.kappa, p.h = 0.01, 0.25, 0.0
    p.flow_direction, p.sampling, p.device = 1, 0, 0
    for k, v in kw.items():
        setattr(p, k, v)
    return p


@pytest.mark.parametrize("kw,status,text", [
    ({"factor": 0}, -1, "Input must be > 0"),                 # MultiframeMotionCoupling.m:228, :1057-1058
    ({"alpha": -1.0}, -1, "Input must be > 0"),
    ({"kappa": 0.0}, -1, "Input must be > 0"),
    ({"flow_direction": 7}, -1, "Invalid flow direction"),
    ({"frame_end": 5}, -1, "frame block"),
    ({"sampling": 9}, -4, "interpMethod"),
    ({"factor": 2}, -4, "factor 4"),                         # 'average' is a block mean only at factor 4
    ({"size": 4}, -1, "size/version"),
])
def test_create_rejects_bad_parameters(kw, status, text):
    lib = _capi.load_library()
    h = C.c_void_p()
    p = _params(**kw)
    rc = lib.mmc_create(C.byref(p), C.byref(h))
    assert rc == status and not h.value
    assert text in lib.mmc_last_error(None).decode()


def test_create_without_gpu_reports_no_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = _capi.load_library()
    h = C.c_void_p()
    p = _params()
    rc = lib.mmc_create(C.byref(p), C.byref(h))
    assert rc == -5 and "no HIP device" in lib.mmc_last_error(None).decode()
    with pytest.raises(_capi.MMCError):
        from superresolution_amd.solver import FrameBlockSolver
        FrameBlockSolver(8, 8, 4, 3)          # the product path fails loudly, no CPU fallback


def test_null_handle_is_rejected():
    lib = _capi.load_library()
    assert lib.mmc_init(None) == -1
    assert lib.mmc_solve(None, 1, 0.0, 0, None, _capi.PROGRESS_CB(), None) == -1
    assert lib.mmc_destroy(None) == 0


def test_static_entry_point_validates_before_touching_the_gpu():
    """mmc_h_partials_static checks pointers and the size of its fixed workspace first, so a
    rejection needs no device."""
    lib = _capi.load_library()
    bogus = C.c_void_p(256)
    assert lib.mmc_h_partials_static(None, bogus, 8, 8, 2, 1, 0, None, bogus) == -1
    assert "null" in lib.mmc_last_error(None).decode()
    assert lib.mmc_h_partials_static(bogus, None, 8, 8, 2, 1, 0, None, bogus) == -1
    assert lib.mmc_h_partials_static(bogus, bogus, 8, 8, 2, 5, 0, None, bogus) == -1
    assert "Invalid flow direction" in lib.mmc_last_error(None).decode()
    per_frame = (1080 * 1920 + 255) // 256
    too_many = _capi.MMC_STATIC_MAX_BLOCKS // per_frame + 1
    assert lib.mmc_h_partials_static(bogus, bogus, 1080, 1920, too_many, 1, 0, None, bogus) == -1
    assert "static workspace" in lib.mmc_last_error(None).decode()
    # a later mmc_create failure replaces the message
    h = C.c_void_p()
    p = _params(factor=0)
    assert lib.mmc_create(C.byref(p), C.byref(h)) == -1
    assert "Input must be > 0" in lib.mmc_last_error(None).decode()


def test_group_create_validates_before_touching_the_gpu():
    lib = _capi.load_library()
    g = C.c_void_p()
    p = _params()
    assert lib.mmc_group_create(C.byref(p), 4, None, C.byref(g)) == -1 and not g.value
    assert "between 1 and nc_global" in lib.mmc_group_last_error(None).decode()
    p = _params(frame_begin=1)
    assert lib.mmc_group_create(C.byref(p), 1, None, C.byref(g)) == -1
    assert "whole sequence" in lib.mmc_group_last_error(None).decode()
    p = _params(stream=C.c_void_p(64))
    assert lib.mmc_group_create(C.byref(p), 1, None, C.byref(g)) == -1
    assert "params.stream must be NULL" in lib.mmc_group_last_error(None).decode()
    assert lib.mmc_group_destroy(None) == 0
    assert lib.mmc_group_init(None) == -1
    assert not lib.mmc_group_block_handle(None, 0)


def test_group_create_without_gpu_reports_no_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = _capi.load_library()
    g = C.c_void_p()
    p = _params()
    assert lib.mmc_group_create(C.byref(p), 2, None, C.byref(g)) == -5 and not g.value
    assert "no HIP device" in lib.mmc_group_last_error(None).decode()


def test_mex_gateway_type_checks_against_the_header():
    """The gateways under mex/ cannot be built without MATLAB; compiled against a declaration-only mex.h
    (tests/mex_stub) every mmc_* call in them is at least checked against include/mmc_hip.h."""
    import subprocess
    for src in ("mmc_hip_mex.cpp", "mmc_sf_mex.cpp"):
        subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only",
                        "-I", os.path.join(ROOT, "tests", "mex_stub"), "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "mex", src)], check=True)


def _sf_params(**kw):
    p = _capi.mmc_sf_params()
    p.size = C.sizeof(_capi.mmc_sf_params)
    p.version = _capi.MMC_ABI_VERSION
    p.ny, p.nx, p.factor, p.num_frames = 8, 8, 4, 5
    p.alpha, p.huber_eps, p.sampling, p.device = 0.1, 0.0, 0, 0
    for k, v in kw.items():
        setattr(p, k, v)
    return p


@pytest.mark.parametrize("kw,status,text", [
    ({"size": 8}, -1, "size/version"),
    ({"num_frames": 0}, -1, "at least one low-resolution frame"),
    ({"num_frames": 70000}, -1, "65535"),
    ({"huber_eps": -0.5}, -1, "huber_eps"),
    ({"sampling": 2}, -4, "built-in samplers"),
    ({"factor": 0}, -1, "Input must be > 0"),                 # through the geometry handle, as mmc_create
    ({"factor": 3}, -4, "factor 4"),
])
def test_single_frame_create_rejects_bad_parameters(kw, status, text):
    """mmc_sf_create validates before it touches a GPU (classical methods/singleframeMotionSR_mitzel_prost.m
    takes any frame count on paper but only builds 5 or 13 data terms, :101-123; here any count from 1 up)."""
    lib = _capi.load_library()
    h = C.c_void_p()
    p = _sf_params(**kw)
    rc = lib.mmc_sf_create(C.byref(p), C.byref(h))
    assert rc == status and not h.value
    assert text in lib.mmc_sf_last_error(None).decode()
    assert lib.mmc_sf_destroy(None) == 0 and lib.mmc_sf_init(None) == -1
