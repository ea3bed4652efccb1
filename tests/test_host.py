"""CPU tests of the host-side mirror of the reference interface: option
validation and error behaviour of MultiframeMotionCoupling (same messages as the
MATLAB class), colour conversion, the bicubic start point, frame partitioning
and the synthetic workload."""
import numpy as np
import pytest

import mmc_oracle as orc
from superresolution_amd import MMCError, sharding, synthetic
from superresolution_amd.imresize import imresize_bicubic, upsample_sequence
from superresolution_amd.mmc import MultiframeMotionCoupling, rgb2ycbcr, ycbcr2rgb


def _obj(**kw):
    lr = np.random.default_rng(0).random((8, 8, 3))
    v = np.zeros((32, 32, 2, 2))
    m = MultiframeMotionCoupling(lr, flowField=v, verbose=0)
    for k, val in kw.items():
        setattr(m, k, val)
    return m


def test_constructor_defaults_match_reference():
    m = _obj()
    # MultiframeMotionCoupling.m:199-216
    assert (m.factor, m.alpha, m.beta, m.kappa) == (4, 0.01, 0.2, 0.25)
    assert m.flowDirection == "backward" and m.interpMethod == "average" and m.interpAA is False
    assert m.comp_mode == "accurate" and m.k is None and m.result1 is None


def test_constructor_errors():
    with pytest.raises(ValueError, match="Load an actual video sequence"):        # :122
        MultiframeMotionCoupling(np.zeros((4, 4)))
    with pytest.raises(ValueError, match="Give both fix frames u0 and w0"):        # :190-192
        MultiframeMotionCoupling(np.zeros((4, 4, 2)), u0_frame=np.zeros((16, 16)))
    with pytest.raises(MMCError, match="warpOp"):
        MultiframeMotionCoupling(np.zeros((4, 4, 2)), warpOp=np.eye(3))


@pytest.mark.parametrize("field,value,msg", [
    ("alpha", 0, "Input must be > 0"), ("factor", -2, "Input must be > 0"),
    ("kappa", np.ones(2), "Input is not scalar"), ("beta", "x", "Input is not numeric"),
])
def test_init_checks_positive_scalars(field, value, msg):                          # :228-231, :1051-1062
    with pytest.raises(ValueError, match=msg):
        _obj(**{field: value}).init()


def test_init_validates_strings_and_frameworks():
    with pytest.raises(ValueError, match="Expected input to match"):               # :242
        _obj(flowDirection="sideways").init()
    with pytest.raises(ValueError, match="Expected input to match"):
        _obj(comp_mode="turbo").init()
    with pytest.raises(ValueError, match='"Fastest" mode can currently only be used'):   # :248-250
        _obj(comp_mode="fastest", framework="flexBox").init()
    with pytest.raises(RuntimeError, match="prost framework not found"):           # :253-258
        _obj(framework="prost").init()
    with pytest.raises(ValueError, match="Invalid framework string"):               # :236 vs :303-306
        _obj(framework="flexBox-vector").init()
    # the flexBox presets run on the HIP library; what init_u_flexBox rejects is rejected first
    with pytest.raises(ValueError, match="Additive warp and full separation not implemented yet for flexBox"):   # :699-702
        _obj(framework="flexBox", kappa=float("nan")).init()
    with pytest.raises(ValueError, match="todo"):                                  # :776-778
        _obj(framework="flexBox", flowDirection="forward-backward").init()
    with pytest.raises(RuntimeError, match='run the "init" method first'):         # :317-319
        _obj().run()


def test_init_requires_precomputed_flow():
    m = MultiframeMotionCoupling(np.zeros((8, 8, 3)), verbose=0)
    with pytest.raises(MMCError, match="flowField"):
        m.init()


def test_fastest_mode_overrides_kappa_then_needs_the_library():
    import torch
    m = _obj(comp_mode="fastest")
    with pytest.warns(UserWarning, match="fastest"):                              # :283-290
        try:
            m.init()
        except MMCError:
            assert not torch.cuda.is_available()        # no GPU here: the HIP path fails loudly
    assert np.isnan(m.kappa)


def test_colour_conversion_round_trip():
    rgb = np.random.default_rng(1).random((5, 6, 3))
    ycc = rgb2ycbcr(rgb)
    assert ycc[..., 0].min() >= 16 / 255 - 1e-12 and ycc[..., 0].max() <= 235 / 255 + 1e-12
    assert np.allclose(ycbcr2rgb(ycc), rgb, atol=1e-12)
    assert np.allclose(rgb2ycbcr(np.ones((1, 1, 3)))[0, 0], [235 / 255, 128 / 255, 128 / 255])


def test_bicubic_start_point_matches_oracle_imresize():
    f = np.random.default_rng(2).random((9, 7, 3))
    ref = np.stack([orc.imresize(f[:, :, i], 4) for i in range(3)], axis=2)
    assert np.abs(upsample_sequence(f, 4) - ref).max() < 1e-13
    assert np.abs(imresize_bicubic(f[:, :, 0], 0.5) - orc.imresize(f[:, :, 0], 0.5)).max() < 1e-13


def test_partition_frames():
    assert sharding.partition_frames(240, 8) == [(30 * r, 30 * r + 30) for r in range(8)]
    assert sharding.partition_frames(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    with pytest.raises(ValueError):
        sharding.partition_frames(3, 4)
    assert sharding.pairs_needed(0, 5, 10) == (0, 5) and sharding.pairs_needed(5, 10, 10) == (4, 9)
    assert sharding.state_frames_needed(5, 8, 10) == (5, 9) and sharding.state_frames_needed(8, 10, 10) == (8, 10)


def test_synthetic_blocks_are_slices_of_the_whole_sequence():
    whole = synthetic.make_block(12, 16, 6, 0, 6, with_truth=True)
    part = synthetic.make_block(12, 16, 6, 2, 4, with_truth=True)
    assert np.array_equal(part["f"], whole["f"][:, :, 2:4])
    assert np.array_equal(part["f_state"], whole["f"][:, :, 2:5])
    assert np.array_equal(part["v"], whole["v"][:, :, 1:4])           # pairs 1,2,3
    assert np.array_equal(part["truth"], whole["truth"][:, :, 2:4])
    assert whole["v"].shape == (48, 64, 5, 2) and whole["f"].min() >= 0 and whole["f"].max() <= 1
    # the flow is the scene motion in the backward convention: warping frame t+1 by v_t gives frame t
    W = orc.warpingOperator([48, 64], whole["v"][:, :, 0, :].astype(np.float64))
    g0, g1 = whole["truth"][:, :, 0].astype(np.float64), whole["truth"][:, :, 1].astype(np.float64)
    warped = (W @ g1.ravel(order="F")).reshape(48, 64, order="F")
    valid = np.asarray(W.sum(axis=1)).ravel().reshape(48, 64, order="F") > 0
    assert np.median(np.abs(warped - g0)[valid]) < 5e-3


def test_video_batch_ranges_follow_the_reference_script():
    from superresolution_amd.video import batch_ranges
    # variants/VideoProcessor_supplementary_MMC.m:102-107, :127, :136 (1-based there)
    assert batch_ranges(10, 4) == [(0, 4), (3, 7), (6, 10), (8, 10)]      # batchEnd = 4*3-10 = 2: the last 2 frames again
    assert batch_ranges(7, 4) == [(0, 4), (3, 7), (5, 7)]
    assert batch_ranges(9, 4) == [(0, 4), (3, 7), (5, 9)]
    assert batch_ranges(12, 5) == [(0, 5), (4, 9), (7, 12)]
    with pytest.raises(ValueError):
        batch_ranges(5, 1)
