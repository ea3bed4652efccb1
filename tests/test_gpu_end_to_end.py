"""GPU end-to-end tests: the reference's quick-start scene in miniature (tube3
fixture, PSNR bar of 0.02 dB against the oracle), the MATLAB-class mirror, flow
swap with warm start (Alternating variant), reproducibility, and size-independent
properties at the benchmark's full size."""
import os

import numpy as np
import pytest

import mmc_oracle as orc
from helpers import make_case, rel_l2

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KBLUR = orc.fspecial_gaussian(7, np.sqrt(0.6))


def _tube3():
    g = np.load(os.path.join(GOLD, "tube3_crop.npz"))
    truth = g["truth_u8"].astype(np.float64) / 255.0
    lr = g["lr"].astype(np.float64)
    v = np.stack([np.stack([orc.imresize(g["flow_lr"][:, :, i, c].astype(np.float64), 4) * 4
                            for c in range(2)], axis=2) for i in range(2)], axis=2)
    return truth, lr, v


def test_tube3_parity_and_psnr():
    """testfile.m in miniature: 3 frames, zoom 4, IC model kappa 0.25, default blur.
    Bars: rel-L2 <= 1e-4 against the oracle at equal iteration count, PSNR of the
    central frame on the 20:end-20 crop within 0.02 dB (testfile.m:71-75)."""
    from superresolution_amd.mmc import MultiframeMotionCoupling
    truth, lr, v = _tube3()
    iters = 300
    P = orc.build_problem(lr, v, k=KBLUR)
    x, _, _ = orc.pdhg(P, iters)
    u_ref, w_ref = orc.split_primal(P, x)
    m = MultiframeMotionCoupling(lr, zoom_factor=4, flowField=v, verbose=0)
    m.k = KBLUR
    m.comp_mode = "fast"
    m.init()
    assert abs(m.h - P.h) < 1e-5 * P.h
    m._opts["max_iters"], m._opts["tol"] = iters, 0.0           # equal iteration count, no early stop
    m.run()
    assert m.result1.shape == (256, 256, 3) and m.result2.shape == (256, 256, 3)
    assert rel_l2(m.result1, u_ref) < 1e-4
    p_hip, p_ref = orc.psnr_central(m.result1, truth), orc.psnr_central(u_ref, truth)
    assert abs(p_hip - p_ref) < 0.02
    assert p_hip > orc.psnr_central(P.u_up, truth) + 0.5         # and it actually super-resolves
    m.close()


def test_h_comes_from_the_bicubic_estimate_on_reinit():
    """MultiframeMotionCoupling.m:432-437 computes h from u_up whatever obj.u holds (:446-450):
    init() after run() must give the same h, and start from the previous result."""
    from superresolution_amd.mmc import MultiframeMotionCoupling
    f, v, _ = make_case(21, 8, 8, 3)
    P = orc.build_problem(f, v, k=KBLUR)
    m = MultiframeMotionCoupling(f, flowField=v, verbose=0)
    m.k = KBLUR
    m.comp_mode = "fast"
    m.init()
    h0 = m.h
    assert abs(h0 - P.h) < 1e-5 * P.h
    m._opts["max_iters"], m._opts["tol"] = 40, 0.0
    m.run()
    u1 = m.result1.copy()
    m.init()
    assert m.h == h0
    u_start, _ = m._solver.get_state()
    assert np.array_equal(u_start, u1.astype(np.float32))
    m.close()


@pytest.mark.parametrize("factor,kappa", [(4, 0.25), (2, float("nan")), (3, 1.0)])
def test_bicubic_start_point_on_the_device(factor, kappa):
    """mmc_set_state(h, NULL, NULL): u = w = imresize(f, factor) computed by the library
    (MultiframeMotionCoupling.m:415-419).  Against the oracle's float64 imresize to 1e-6, and h
    from it equal to the h of a host-computed start point to 1e-5."""
    from superresolution_amd.solver import FrameBlockSolver
    ny, nx, nc = 13, 9, 3
    f, v, _ = make_case(51, ny, nx, nc, factor=factor, flow="smooth", flow_scale=1.5)
    u_ref = np.stack([orc.imresize(f[:, :, i], factor) for i in range(nc)], axis=2)
    hs = []
    for start in (None, u_ref):
        with FrameBlockSolver(ny, nx, factor, nc, kappa=kappa, sampling="stride") as s:
            s.set_lowres(f); s.set_blur(KBLUR); s.set_flow(v)
            s.set_state(start)
            u, w = s.get_state()
            assert np.abs(u - u_ref).max() <= 1e-6
            if w is not None:
                assert np.array_equal(u, w)
            s.init()
            hs.append(s.h)
    assert abs(hs[0] - hs[1]) <= 1e-5 * hs[1]


def test_colour_input_and_presets():
    """4-D RGB input goes through Y of rgb2ycbcr (:110-119) and comes back as RGB (:1022-1046);
    'fastest' switches to the additive model (:283-290) and stops on the 1e-4 tolerance."""
    from superresolution_amd.mmc import MultiframeMotionCoupling, rgb2ycbcr
    rng = np.random.default_rng(3)
    f, v, _ = make_case(4, 8, 8, 3)
    rgb = np.clip(np.stack([f, f * 0.8 + 0.1, f * 0.6 + 0.2], axis=2) + 0.01 * rng.normal(size=(8, 8, 3, 3)), 0, 1)
    m = MultiframeMotionCoupling(rgb, flowField=v, verbose=0)
    m.k = KBLUR
    m.comp_mode = "fastest"
    with pytest.warns(UserWarning):
        m.init()
    m.run()
    assert np.isnan(m.kappa) and m.result1.shape == (32, 32, 3, 3) and m.result2 is None
    assert m.last_stats["converged"] and m.last_stats["iterations"] < 2500
    out = m.result1[:, :, :, 1]
    ingamut = np.all((out > 0) & (out < 1), axis=2)          # ycbcr2rgb clips to [0,1]
    y = rgb2ycbcr(out)[:, :, 0]
    assert ingamut.mean() > 0.5 and np.abs(y - m._u[:, :, 1])[ingamut].max() < 1e-6
    m.close()


def test_flow_swap_keeps_state_and_h():
    """variants/MultiframeMotionCouplingAlternating.m:108-159: new flows, same solver state, h fixed."""
    from superresolution_amd.solver import FrameBlockSolver
    f, v, _ = make_case(12, 8, 8, 3, flow_scale=1.0)
    _, v2, _ = make_case(13, 8, 8, 3, flow_scale=1.0)
    P = orc.build_problem(f, v, k=KBLUR)
    s = FrameBlockSolver(8, 8, 4, 3)
    s.set_lowres(f); s.set_blur(KBLUR); s.set_flow(v); s.set_state(P.u_up, P.u_up); s.init()
    h0 = s.h
    s.solve(50)
    u1, w1 = s.get_state()
    s.set_flow(v2)
    s.init()
    assert s.h == h0
    u1b, _ = s.get_state()
    assert np.array_equal(u1, u1b)
    s.solve(50)
    # oracle: 50 iterations on v, then continue from that primal/dual state on the problem built with v2 and the same h
    x, y, _ = orc.pdhg(P, 50)
    P2 = orc.build_problem(f, v2, k=KBLUR, h=P.h)
    rowzero = np.asarray(abs(P2.K).sum(axis=1)).ravel() == 0
    y = np.where(rowzero, 0.0, y)          # temporal duals of rows the new flow masks are reset
    x2, _, _ = orc.pdhg(P2, 50, x=x, y=y)
    u2, _ = s.get_state()
    assert rel_l2(u2, orc.split_primal(P2, x2)[0]) < 1e-4
    s.close()


def test_runs_are_bitwise_reproducible():
    from superresolution_amd.solver import FrameBlockSolver
    f, v, occ = make_case(14, 16, 20, 4, flow="random", flow_scale=2.0)
    P = orc.build_problem(f, v, k=KBLUR)
    outs = []
    for _ in range(2):
        s = FrameBlockSolver(16, 20, 4, 4, h=P.h)
        s.set_lowres(f); s.set_blur(KBLUR); s.set_flow(v); s.set_state(P.u_up, P.u_up); s.init()
        s.solve(60)
        outs.append(s.get_state())
        s.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


def test_results_do_not_depend_on_the_plane_placement(monkeypatch):
    """The per-pixel planes are shifted against each other inside their allocations (a tuning choice,
    MMC_PLANE_SKEW fixes it): every shift must give bit-for-bit the same iterates."""
    from superresolution_amd.solver import FrameBlockSolver
    f, v, occ = make_case(15, 16, 20, 4, flow="random", flow_scale=2.0)
    P = orc.build_problem(f, v, k=KBLUR)
    outs = []
    for skew in ("0", "256", "45056", "327680", "1048576"):
        monkeypatch.setenv("MMC_PLANE_SKEW", skew)
        with FrameBlockSolver(16, 20, 4, 4, h=P.h) as s:
            assert s.plane_skew == int(skew)
            s.set_lowres(f); s.set_blur(KBLUR); s.set_flow(v); s.set_state(P.u_up, P.u_up); s.init()
            s.solve(40)
            outs.append(s.get_state())
    for u, w in outs[1:]:
        assert np.array_equal(u, outs[0][0]) and np.array_equal(w, outs[0][1])
    assert rel_l2(outs[0][0], orc.split_primal(P, orc.pdhg(P, 40)[0])[0]) < 1e-4


def test_full_size_properties():
    """Benchmark size (1080x1920 HR, here 6 frames): properties that need no oracle run --
    adjointness <Kx,y> = <x,K'y> of the device operators, energy decrease, residual
    decrease, box constraint, and agreement of h with float64 sums."""
    from superresolution_amd import synthetic
    from superresolution_amd.imresize import upsample_sequence
    from superresolution_amd.solver import FrameBlockSolver
    ny, nx, nc = 270, 480, 6
    blk = synthetic.make_block(ny, nx, nc, 0, nc, device="cuda")
    u_up = upsample_sequence(blk["f_state"], 4).astype(np.float32)
    s = FrameBlockSolver(ny, nx, 4, nc)
    s.set_lowres(blk["f"]); s.set_blur(synthetic.gaussian_kernel()); s.set_flow(blk["v"]); s.set_state(u_up, u_up)
    s.init()
    N, M, n = 1080 * 1920, 1080 * 1920 * nc, ny * nx
    rng = np.random.default_rng(0)
    xu = rng.standard_normal((1080, 1920, nc), dtype=np.float32)
    xw = rng.standard_normal((1080, 1920, nc), dtype=np.float32)
    yp = rng.standard_normal(n * nc, dtype=np.float32)
    y1 = rng.standard_normal(3 * M, dtype=np.float32)
    y2 = rng.standard_normal(3 * M, dtype=np.float32)
    p, g1, g2 = s.debug_apply_K(xu, xw)
    # temporal rows that the operator masks (and the last frame's) carry no dual
    y1[:M][g1[:M] == 0] = 0
    y2[:M][g2[:M] == 0] = 0
    ku, kw = s.debug_apply_KT(yp, y1, y2)
    lhs = np.dot(p.astype(np.float64), yp) + np.dot(g1.astype(np.float64), y1) + np.dot(g2.astype(np.float64), y2)
    rhs = np.dot(ku.astype(np.float64), xu.ravel(order="F")) + np.dot(kw.astype(np.float64), xw.ravel(order="F"))
    assert abs(lhs - rhs) < 1e-4 * max(abs(lhs), abs(rhs))
    s.set_state(u_up, u_up)
    e0 = s.energy()
    res = []
    st = s.solve(200, 0.0, 50, callback=lambda it, d: res.append(d["res_primal"]) and False)
    e1 = s.energy()
    u, w = s.get_state()
    assert e1 < 0.9 * e0 and res[-1] < res[0] and st["iterations"] == 200
    assert u.min() >= 0.0 and u.max() <= 1.0 and np.isfinite(w).all()
    s.close()


def test_flexbox_framework_presets_start_point_and_block_mean():
    """framework = 'flexBox' (init_u_flexBox, MultiframeMotionCoupling.m:690-902) on the HIP library:
    iteration caps 2500 / 800 (:798-802), u starts at the bicubic estimate and w at zero (:893-896),
    'average' without anti-aliasing is the block mean of superpixelOperator at any zoom (:707-708).
    Against the oracle started at the same point, zoom 2, forward flow (the direction in which the
    reference's flexBox terms (:834) and its prost terms (:490-491) state the same model)."""
    from superresolution_amd.mmc import MultiframeMotionCoupling
    f, v, _ = make_case(8, 10, 8, 3, factor=2, flow="smooth", flow_scale=1.0)
    box2 = (lambda x: (np.abs(x) <= 0.5) / 2.0, 2.0)            # two taps of 1/2: the 2x2 block mean
    P = orc.build_problem(f, v, factor=2, k=KBLUR, flowDirection="forward", interpMethod="custom", interpKernel=box2)
    iters = 80
    x0 = np.concatenate([P.u_up.ravel(order="F"), np.zeros(P.M)])
    x, _, _ = orc.pdhg(P, iters, x=x0)
    u_ref, w_ref = orc.split_primal(P, x)
    m = MultiframeMotionCoupling(f, zoom_factor=2, flowField=v, verbose=0)
    m.framework, m.k, m.flowDirection = "flexBox", KBLUR, "forward"
    m.init()
    assert m._opts == {"max_iters": 2500, "tol": 1e-6}
    assert abs(m.h - P.h) < 1e-5 * P.h
    u0, w0 = m._solver.get_state()
    assert np.abs(u0 - P.u_up).max() < 1e-6 and not w0.any()
    m._opts["max_iters"], m._opts["tol"] = iters, 0.0
    m.run()
    assert rel_l2(m.result1, u_ref) < 1e-4 and np.abs(m.result2 - w_ref).max() < 1e-4
    m.comp_mode = "fast"
    m.init()
    assert m._opts == {"max_iters": 800, "tol": 1e-4}
    m.close()


def test_flexbox_backward_terms_as_written():
    """init_u_flexBox, flowDirection 'backward': the temporal rows {Id,-Wi,Id,-Wi} (:864) couple
    u + w.  The mirror solves that by default for framework 'flexBox'; the oracle states the same
    rows (flexbox_backward_terms).  Operators first (K x and K' y against the explicit sparse
    matrix), then the solve; with the switch off the result is the u - w model's."""
    from superresolution_amd.mmc import MultiframeMotionCoupling
    from superresolution_amd.solver import FrameBlockSolver
    f, v, occ = make_case(12, 9, 8, 3, flow="smooth", flow_scale=1.5, with_occ=True)
    P = orc.build_problem(f, v, k=KBLUR, occ=occ, flexbox_backward_terms=True)
    Pm = orc.build_problem(f, v, k=KBLUR, occ=occ)
    assert abs(P.K - Pm.K).sum() > 1.0 and np.allclose(P.T, Pm.T) and np.allclose(P.S, Pm.S)
    Ny, Nx, nc = P.dims[:3]
    M, n = P.M, P.n
    rng = np.random.default_rng(5)
    with FrameBlockSolver(9, 8, 4, 3, h=P.h) as s:
        s.set_lowres(f); s.set_blur(KBLUR); s.set_flow(v, occ); s.set_state(P.u_up, P.u_up)
        s.set_flexbox_backward_terms(True)
        s.init()
        x = rng.normal(size=2 * M)
        p, g1, g2 = s.debug_apply_K(x[:M].reshape((Ny, Nx, nc), order="F"), x[M:].reshape((Ny, Nx, nc), order="F"))
        ref = P.K @ x
        assert rel_l2(g1, ref[n * nc:n * nc + 3 * M]) < 3e-6 and rel_l2(g2, ref[n * nc + 3 * M:]) < 3e-6
        y = rng.normal(size=P.K.shape[0])
        y[np.asarray(abs(P.K).sum(axis=1)).ravel() == 0] = 0
        ku, kw = s.debug_apply_KT(y[:n * nc], y[n * nc:n * nc + 3 * M], y[n * nc + 3 * M:])
        ref = P.KT @ y
        assert rel_l2(ku, ref[:M]) < 3e-6 and rel_l2(kw, ref[M:]) < 3e-6
    iters = 80
    x0 = np.concatenate([P.u_up.ravel(order="F"), np.zeros(M)])
    results = {}
    for as_written, prob in ((True, P), (False, Pm)):
        x, _, _ = orc.pdhg(prob, iters, x=x0)
        u_ref, w_ref = orc.split_primal(prob, x)
        m = MultiframeMotionCoupling(f, flowField=v, occlusion=occ, verbose=0)
        m.framework, m.k, m.flexBoxBackwardAsWritten = "flexBox", KBLUR, as_written
        m.init()
        m._opts["max_iters"], m._opts["tol"] = iters, 0.0
        m.run()
        assert rel_l2(m.result1, u_ref) < 1e-4 and np.abs(m.result2 - w_ref).max() < 1e-4
        results[as_written] = m.result2.copy()
        m.close()
    assert np.abs(results[True] - results[False]).max() > 1e-3          # the two models do differ


def test_script_sequence_from_files(tmp_path):
    """The call sequence of variants/testfile_gt_sintel.m with every input read from files: frames
    through LoadImSequence (:28), ground-truth flows from .flo files (:33-39), occlusion maps from
    images (:42-46, handed over as maps where the script prebuilds a sparse warpOp), then init / run
    and the error margin of the central frame (:99-103)."""
    PIL = pytest.importorskip("PIL.Image")
    from scipy import ndimage
    from superresolution_amd import flowio, loaders, metrics
    from superresolution_amd.mmc import MultiframeMotionCoupling, rgb2ycbcr
    rng = np.random.default_rng(21)
    Ny, Nx, nf, d = 96, 128, 5, (0.75, -0.5)                       # d = (along columns, along rows) per frame
    base = ndimage.gaussian_filter(rng.random((Ny + 40, Nx + 40, 3)), (1.2, 1.2, 0))
    base = (base - base.min()) / (base.max() - base.min())
    frames_dir, flow_dir, occ_dir = tmp_path / "video", tmp_path / "flow", tmp_path / "occ"
    for p in (frames_dir, flow_dir, occ_dir):
        p.mkdir()
    for t in range(nf):
        img = ndimage.shift(base, (t * d[1], t * d[0], 0), order=3, mode="nearest")[20:20 + Ny, 20:20 + Nx]
        PIL.fromarray(np.clip(np.round(img * 255), 0, 255).astype(np.uint8)).save(str(frames_dir / ("frame_%04d.png" % (t + 1))))
    occ_true = np.zeros((Ny, Nx, nf - 1), dtype=bool)
    for t in range(nf - 1):
        flowio.writeFlowFile(np.stack([np.full((Ny, Nx), d[0]), np.full((Ny, Nx), d[1])], axis=2), str(flow_dir / ("frame_%04d.flo" % (t + 1))))
        occ_true[10 + 3 * t:20 + 3 * t, 30:50, t] = True
        PIL.fromarray((occ_true[:, :, t] * 255).astype(np.uint8)).save(str(occ_dir / ("frame_%04d.png" % (t + 1))))

    small, large = loaders.LoadImSequence(str(frames_dir), 1, nf, 4, "bicubic")
    v = flowio.load_flow_sequence(str(flow_dir), nf - 1)
    occ = flowio.load_occlusion_sequence(str(occ_dir), nf - 1)
    assert small.shape == (Ny // 4, Nx // 4, 3, nf) and v.shape == (Ny, Nx, nf - 1, 2) and np.array_equal(occ, occ_true)

    m = MultiframeMotionCoupling(small, flowField=v, occlusion=occ, verbose=0)
    m.kappa = 0.5                                                  # :79
    m.k = KBLUR
    m.init()
    m._opts.update(max_iters=80, tol=0.0)                          # equal iteration count for the comparison below
    m.run()
    y_small = np.stack([rgb2ycbcr(small[:, :, :, i])[:, :, 0] for i in range(nf)], axis=2)
    P = orc.build_problem(y_small, v, kappa=0.5, k=KBLUR, occ=occ)
    x, _, _ = orc.pdhg(P, 80)
    u_ref, w_ref = orc.split_primal(P, x)
    assert abs(m.h - P.h) < 1e-5 * P.h
    assert rel_l2(m._u, u_ref) < 1e-4 and rel_l2(m._w, w_ref) < 1e-3
    assert m.result1.shape == (Ny, Nx, 3, nf)
    m._opts.update(max_iters=1500, tol=1e-5)                       # the solver keeps its state: the solve goes on
    m.run()
    psnr_sr, ssim_sr = metrics.error_margin(m.result1, large)
    up = np.stack([np.clip(loaders.imresize(small[:, :, :, i], 4.0), 0, 1) for i in range(nf)], axis=3)
    psnr_up, ssim_up = metrics.error_margin(up, large)
    print("PSNR %.2f dB (bicubic %.2f), SSIM %.3f (bicubic %.3f)" % (psnr_sr, psnr_up, ssim_sr, ssim_up))
    assert psnr_sr > psnr_up + 0.3 and ssim_sr > ssim_up + 0.02
    m.close()
