"""The image-sequence loaders (superresolution_amd/loaders.py) against the oracle's restatement of
imresize, helpers/samplingOperator.m and helpers/RepConvMtx.m, on frames written to a temporary
folder (helpers/LoadImSequence.m, LoadImSequenceForward.m, LoadImSequenceReal.m)."""
import os

import numpy as np
import pytest

import mmc_oracle as orc
from superresolution_amd import loaders
from superresolution_amd.mmc import rgb2ycbcr, ycbcr2rgb

PIL = pytest.importorskip("PIL.Image")


def write_frames(folder, n, rows=48, cols=64, ext="png", grey=False, seed=0, prefix="Frame"):
    rng = np.random.default_rng(seed)
    frames = []
    for i in range(n):
        yy, xx = np.mgrid[0:rows, 0:cols]
        base = 0.5 + 0.4 * np.sin(0.21 * (xx + 2 * i)) * np.cos(0.17 * (yy - i))
        img = np.stack([base, np.roll(base, 3, axis=0), 1.0 - base], axis=2) + 0.05 * rng.standard_normal((rows, cols, 3))
        u8 = np.clip(np.round(img * 255), 0, 255).astype(np.uint8)
        if grey:
            u8 = u8[:, :, 0]
        extra = {"quality": 100, "subsampling": 0} if ext == "jpg" else {}
        PIL.fromarray(u8).save(os.path.join(folder, "%s%03d.%s" % (prefix, i + 1, ext)), **extra)
        frames.append(u8.astype(np.float64) / 255.0)
    return frames


@pytest.mark.parametrize("method", ["bicubic", "bilinear", "nearest", "lanczos3"])
def test_load_im_sequence(tmp_path, method):
    frames = write_frames(str(tmp_path), 5)
    small, large = loaders.LoadImSequence(str(tmp_path), 2, 3, 4, method)
    assert small.shape == (12, 16, 3, 3) and large.shape == (48, 64, 3, 3)
    for j in range(3):
        assert np.array_equal(large[:, :, :, j], frames[1 + j])                      # frames 2..4, 1-based start
        for c in range(3):
            ref = np.clip(orc.imresize(frames[1 + j][:, :, c], 0.25, method, True), 0, 1)   # :45-46, :61-62
            assert np.abs(small[:, :, c, j] - ref).max() < 1e-13
    assert small.min() >= 0.0 and small.max() <= 1.0


def test_defaults_listing_order_and_errors(tmp_path):
    write_frames(str(tmp_path), 2, ext="png", seed=1)
    jpg = write_frames(str(tmp_path), 1, ext="jpg", seed=2, prefix="Zlast")         # dir('*.jpg') comes first (:18)
    small, large = loaders.LoadImSequence(str(tmp_path), 1, 1)
    assert np.abs(large[:, :, :, 0] - jpg[0]).max() < 0.03                           # lossy file, but that frame
    ref = np.clip(orc.imresize(large[:, :, 0, 0], 0.25, "bicubic", True), 0, 1)
    assert np.abs(small[:, :, 0, 0] - ref).max() < 1e-13                             # factor 4, bicubic by default
    real = loaders.LoadImSequenceReal(str(tmp_path))                                 # png before jpg there (:5)
    assert real.shape == (48, 64, 3, 3)
    assert np.abs(real[:, :, :, 2] - jpg[0]).max() < 0.03
    with pytest.raises(ValueError, match="dividable"):
        loaders.LoadImSequence(str(tmp_path), 1, 1, 5)                               # 48/5: :26-28
    empty = tmp_path / "none"
    empty.mkdir()
    with pytest.raises(ValueError, match="data not found"):
        loaders.LoadImSequenceReal(str(empty))


def test_jpeg_round_trip_stays_close(tmp_path):
    write_frames(str(tmp_path), 1)
    a, _ = loaders.LoadImSequence(str(tmp_path), 1, 1, 4, "bicubic", 0)
    b, _ = loaders.LoadImSequence(str(tmp_path), 1, 1, 4, "bicubic", 1)
    assert 0 < np.abs(a - b).max() < 0.05                                            # 8-bit quantisation + quality 100


def test_grey_frames(tmp_path):
    frames = write_frames(str(tmp_path), 3, grey=True)
    small, large = loaders.LoadImSequence(str(tmp_path), 1, 3)
    assert large.shape == (48, 64, 3, 3) and small.shape == (12, 16, 3, 3)
    for c in range(3):
        assert np.array_equal(large[:, :, c, 1], frames[1])
    real = loaders.LoadImSequenceReal(str(tmp_path), 2, 2)
    assert real.shape == (48, 64, 2) and np.array_equal(real[:, :, 0], frames[1])


@pytest.mark.parametrize("method", ["average", "stride", "bicubic"])
def test_load_im_sequence_forward(tmp_path, method):
    frames = write_frames(str(tmp_path), 3)
    k = orc.fspecial_gaussian(7, np.sqrt(0.6))
    k = k + 0.01 * np.arange(49).reshape(7, 7) / 49.0                               # not symmetric: the flip matters
    small, large = loaders.LoadImSequenceForward(str(tmp_path), 1, 2, 4, method, k, False)
    assert small.shape == (12, 16, 3, 2)
    D = orc.samplingOperator((48, 64), (12, 16), method)                             # :34
    A = D @ orc.RepConvMtx(k, (48, 64))                                              # :35
    for j in range(2):
        ycc = rgb2ycbcr(frames[j])
        y_s = (A @ ycc[:, :, 0].ravel(order="F")).reshape((12, 16), order="F")       # :43
        cb = orc.imresize(ycc[:, :, 1], 0.25)
        cr = orc.imresize(ycc[:, :, 2], 0.25)
        ref = ycbcr2rgb(np.stack([y_s, cb, cr], axis=2))
        assert np.abs(small[:, :, :, j] - ref).max() < 1e-12
    q, _ = loaders.LoadImSequenceForward(str(tmp_path), 1, 2, 4, method, k, True)
    assert np.abs(q * 255 - np.round(q * 255)).max() < 1e-9 and np.abs(q - small).max() <= 0.5 / 255 + 1e-12
    with pytest.raises(ValueError, match="odd"):
        loaders.LoadImSequenceForward(str(tmp_path), 1, 1, 4, method, np.ones((4, 4)) / 16)


def test_bundled_scene_if_present():
    """The reference's own scene (testfile.m:22-31) when its checkout is next to this repository."""
    folder = "/root/reference/data/video_scenes/tube3"
    if not os.path.isdir(folder):
        pytest.skip("reference checkout not present")
    small, large = loaders.LoadImSequence(folder, 1, 3, 4, "bicubic")
    assert large.shape == (600, 600, 3, 3) and small.shape == (150, 150, 3, 3)
    assert 0.0 <= small.min() and small.max() <= 1.0
    ref = np.clip(orc.imresize(large[:, :, 1, 2], 0.25, "bicubic", True), 0, 1)
    assert np.abs(small[:, :, 1, 2] - ref).max() < 1e-13
